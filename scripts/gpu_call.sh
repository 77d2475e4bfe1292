#!/bin/bash
# ONE parametrised runner for gpurun calls (replaces the 33 one-shot scripts of round 2; what each call of a round ran and
# where its output went is indexed in profiles/CALLS.md).  Usage, from the build container:
#
#   gpurun --timeout 900 -- 'bash scripts/gpu_call.sh <tag> <step> [<step> ...]'
#
# Steps (each writes gpurun_out/<tag>_<step>.log; steps run in the order given):
#   tests[:<pytest -k expr>]   python -m pytest tests -m gpu -q [-k expr]
#   testfile:<path>[:<k>]      one test file
#   bench[:<extra flags>]      python bench.py <flags>          (last line -> gpurun_out/<tag>_bench.json)
#   stats[:<bench flags>]      rocprofv3 --kernel-trace --stats around bench.py (summary -> gpurun_out/<tag>_kernel_stats.txt)
#   ab[:<args>]                python scripts/ab_step.py <args> (same-process A/B of engine options)
#   traffic                    the two PMC passes of scripts/pmc_workload.py + scripts/pmc_traffic.py -> gpurun_out/<tag>_gemm_traffic.json
#   py:<script and args>       python <script and args>
#   exe:<file.hip and args>    hipcc the standalone HIP program on the box and run it
#   pmcsq:<M N K akc bkc [mode]>  rocprofv3 SQ counter pass over scripts/pmc_gemm.py for one shape on pp256 (MFMA pipe busy)
#   dp2gloo[:<bench flags>]    bench.py --gpus 2, both ranks on this GPU over gloo (functional run of the N > 1 path)
# Colons separate the step name from its argument; spaces inside an argument must be written as '+'.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
tag=$1; shift
for step in "$@"; do
  name=${step%%:*}; arg=""; [[ "$step" == *:* ]] && arg=${step#*:}; arg=${arg//+/ }
  log=gpurun_out/${tag}_${name}.log
  echo "=== $step"
  case $name in
    tests)
      if [ -n "$arg" ]; then timeout -k 10 1500 python -m pytest tests -m gpu -q -x -k "$arg" > "$log" 2>&1
      else timeout -k 10 1500 python -m pytest tests -m gpu -q > "$log" 2>&1; fi
      tail -n 15 "$log" ;;
    testfile)
      f=${arg%%:*}; k=""; [[ "$arg" == *:* ]] && k=${arg#*:}
      log=gpurun_out/${tag}_$(basename "$f" .py).log
      if [ -n "$k" ]; then timeout -k 10 1500 python -m pytest "$f" -m gpu -q -x -k "$k" > "$log" 2>&1
      else timeout -k 10 1500 python -m pytest "$f" -m gpu -q -x > "$log" 2>&1; fi
      tail -n 12 "$log" ;;
    bench)
      # stdout alone (as the driver reads it): it must hold the JSON line and nothing else
      timeout -k 10 900 python bench.py $arg > gpurun_out/${tag}_bench.json 2> "$log"
      echo "stdout lines: $(wc -l < gpurun_out/${tag}_bench.json)"; cut -c1-400 gpurun_out/${tag}_bench.json ;;
    stats)
      flags=${arg:---steps 3 --warmup 1 --no-cpu-baseline --no-other-stages --no-profile}
      ( cd /tmp && timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$tag" -- \
          python "$GRAFT_REPO_ROOT/bench.py" $flags > "$GRAFT_REPO_ROOT/$log" 2>&1 )
      python scripts/rocpd_stats.py "gpurun_out/prof_$tag" "$flags" > gpurun_out/${tag}_kernel_stats.txt 2>&1
      head -n 40 gpurun_out/${tag}_kernel_stats.txt
      find "gpurun_out/prof_$tag" -type f ! -name "*stats*" -size +1M -delete 2>/dev/null ;;
    ab)
      timeout -k 10 900 python scripts/ab_step.py $arg > "$log" 2>&1; cat "$log" | grep -v "^$" | tail -n 40 ;;
    traffic)
      ( cd /tmp && timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_fetch_$tag" -- \
          python "$GRAFT_REPO_ROOT/scripts/pmc_workload.py" > "$GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_fetch.log" 2>&1 )
      ( cd /tmp && timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_write_$tag" -- \
          python "$GRAFT_REPO_ROOT/scripts/pmc_workload.py" > "$GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_write.log" 2>&1 )
      python scripts/pmc_traffic.py "gpurun_out/pmc_fetch_$tag" "gpurun_out/pmc_write_$tag" "gpurun_out/${tag}_gemm_traffic.json" 2>&1 | tail -n 8
      find "gpurun_out/pmc_fetch_$tag" "gpurun_out/pmc_write_$tag" -type f -size +1M -delete 2>/dev/null ;;
    py)
      set -- $arg; log=gpurun_out/${tag}_$(basename "$1" .py).log      # one log per script: two py steps of a call must not share one
      timeout -k 10 900 python $arg > "$log" 2>&1; tail -n 60 "$log" ;;
    exe)
      # a standalone HIP program: arg = "<source.hip> [args]" -- compiled on the box (hipcc, seconds) and run
      set -- $arg; src=$1; shift; log=gpurun_out/${tag}_$(basename "$src" .hip).log
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 "$src" -o /tmp/exe_$$ > "$log" 2>&1 && timeout -k 10 600 /tmp/exe_$$ "$@" >> "$log" 2>&1
      tail -n 80 "$log" ;;
    pmcsq)
      # SQ counter pass (MFMA pipe busy, wait cycles) over scripts/pmc_gemm.py for ONE shape: arg = "M N K akc bkc [mode]"
      set -- $arg; shp="$1x$2x$3_$4$5${6:-b}"
      ( cd /tmp && timeout -k 10 180 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS \
          SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_sq_${tag}_$shp" -- \
          python "$GRAFT_REPO_ROOT/scripts/pmc_gemm.py" $1 $2 $3 $4 $5 pp256 ${6:-b} 5 > /dev/null 2>&1 )
      python - "gpurun_out/pmc_sq_${tag}_$shp" <<'PY' | tee -a gpurun_out/${tag}_pmc_sq_summary.txt
import collections, csv, glob, sys
d = sys.argv[1]
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'gemm_bf16_pp' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print('##', d)
    for k, v in sorted(acc.items()):
        print(f'  {k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})')
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in acc and 'GRBM_GUI_ACTIVE' in acc:
        m = sum(acc['SQ_VALU_MFMA_BUSY_CYCLES']) / len(acc['SQ_VALU_MFMA_BUSY_CYCLES']) / 1024
        g = sum(acc['GRBM_GUI_ACTIVE']) / len(acc['GRBM_GUI_ACTIVE']) / 8
        print(f'  -> MFMA pipe busy {100 * m / g:.1f} % of the kernel; waves waiting {100 * sum(acc["SQ_WAIT_ANY"]) / sum(acc["SQ_WAVE_CYCLES"]):.1f} %')
PY
      find "gpurun_out/pmc_sq_${tag}_$shp" -type f -size +1M -delete 2>/dev/null ;;
    dp2gloo)
      # bench.py --gpus 2 with BOTH ranks on this one GPU over gloo (host-bounced collectives): a functional run of the N > 1
      # bench path (dp block of the JSON line, sharded optimiser step), not a measurement
      MD_DIST_BACKEND=gloo MD_DP_MODE=${MD_DP_MODE:-sharded} MD_DP_EXCHANGE=bf16 timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --microbatch 256 --no-cpu-baseline --no-other-stages $arg > "$log" 2>&1
      grep "^{" "$log" | tail -n 1 > gpurun_out/${tag}_bench_dp2_gloo.json; cut -c1-600 gpurun_out/${tag}_bench_dp2_gloo.json; tail -n 5 "$log" | cut -c1-300 ;;
    *) echo "unknown step $name" ;;
  esac
done
