# The round's final measurements in ONE gpurun call (profiles/CALLS.md): full GPU test suite, smoke(), bench (driver flags), kernel
# statistics at microbatch 1024 / 256 and of the res-512 stage, PMC traffic, the 2-rank gloo run of the N > 1 bench path, the library yardstick.
#   gpurun --timeout 2400 -- 'bash scripts/final_call.sh <tag>'
cd "$GRAFT_REPO_ROOT" || exit 1
t=${1:-m1}
bash scripts/gpu_call.sh $t traffic
cp gpurun_out/${t}_gemm_traffic.json profiles/r6_gemm_traffic.json     # on the box: the bench below reads roofline.traffic from it (keyed to the GEMM sources)
bash scripts/gpu_call.sh $t tests "bench" "stats:--steps+3+--warmup+1+--no-cpu-baseline+--no-other-stages+--no-profile"
cp gpurun_out/${t}_kernel_stats.txt gpurun_out/${t}_kernel_stats_mb1024.txt
bash scripts/gpu_call.sh ${t}b "stats:--microbatch+256+--steps+2+--warmup+1+--no-cpu-baseline+--no-other-stages+--no-profile" dp2gloo
cp gpurun_out/${t}b_kernel_stats.txt gpurun_out/${t}_kernel_stats_mb256.txt
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_${t}_res512" -- python "$GRAFT_REPO_ROOT/scripts/run_stage.py" res_512_pretrain 1 > "$GRAFT_REPO_ROOT/gpurun_out/${t}_res512.log" 2>&1 )
python scripts/rocpd_stats.py gpurun_out/prof_${t}_res512 "python scripts/run_stage.py res_512_pretrain 1" > gpurun_out/${t}_kernel_stats_res_512_pretrain.txt 2>&1
tail -n 2 gpurun_out/${t}_res512.log
find gpurun_out/prof_${t}_res512 -type f ! -name "*stats*" -size +1M -delete 2>/dev/null
python scripts/bench_gemm_vs_library.py > gpurun_out/${t}_gemm_vs_hipblaslt.txt 2>&1; tail -n 30 gpurun_out/${t}_gemm_vs_hipblaslt.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
