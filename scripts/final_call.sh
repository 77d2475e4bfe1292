bash scripts/gpu_call.sh h1 tests "bench" "stats:--steps+3+--warmup+1+--no-cpu-baseline+--no-other-stages+--no-profile" traffic
cp gpurun_out/h1_kernel_stats.txt gpurun_out/h1_kernel_stats_mb1024.txt
bash scripts/gpu_call.sh h1b "stats:--microbatch+256+--steps+2+--warmup+1+--no-cpu-baseline+--no-other-stages+--no-profile" dp2gloo
cp gpurun_out/h1b_kernel_stats.txt gpurun_out/h1_kernel_stats_mb256.txt
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_h1_res512" -- python "$GRAFT_REPO_ROOT/scripts/run_stage.py" res_512_pretrain 1 > "$GRAFT_REPO_ROOT/gpurun_out/h1_res512.log" 2>&1 )
python scripts/rocpd_stats.py gpurun_out/prof_h1_res512 "python scripts/run_stage.py res_512_pretrain 1" > gpurun_out/h1_kernel_stats_res_512_pretrain.txt 2>&1
tail -n 2 gpurun_out/h1_res512.log
find gpurun_out/prof_h1_res512 -type f ! -name "*stats*" -size +1M -delete 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
