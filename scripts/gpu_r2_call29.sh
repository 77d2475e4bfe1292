#!/bin/bash
# round-2 final: full GPU suite, the default bench line, kernel-trace stats, PMC traffic + SQ passes
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/c29_pytest.log; tail -3 gpurun_out/c29_pytest.log
timeout -k 10 500 python bench.py > gpurun_out/c29_bench.log 2>&1; tail -1 gpurun_out/c29_bench.log > gpurun_out/r2_final_bench.json; cut -c1-300 gpurun_out/r2_final_bench.json
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r2_final -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-stages --no-profile > $GRAFT_REPO_ROOT/gpurun_out/c29_rocprof.log 2>&1 )
tail -1 gpurun_out/c29_rocprof.log | cut -c1-200
( cd /tmp && timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_fetch_f -- python $GRAFT_REPO_ROOT/scripts/pmc_workload.py > $GRAFT_REPO_ROOT/gpurun_out/c29_pmc_fetch.log 2>&1 )
( cd /tmp && timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_write_f -- python $GRAFT_REPO_ROOT/scripts/pmc_workload.py > $GRAFT_REPO_ROOT/gpurun_out/c29_pmc_write.log 2>&1 )
python scripts/pmc_traffic.py gpurun_out/pmc_fetch_f gpurun_out/pmc_write_f gpurun_out/r2_gemm_traffic_final.json 2>&1 | tail -5
for shp in "65536 1024 1024" "65536 3072 1024"; do
( cd /tmp && timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq_final_$(echo $shp | tr ' ' x) -- python $GRAFT_REPO_ROOT/scripts/pmc_gemm.py $shp 1 1 pp256 b 5 > /dev/null 2>&1 )
done
python - <<'PY' > gpurun_out/c29_pmc_sq_summary.txt 2>&1
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_sq_final_*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'gemm_bf16_pp' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        print('##', d)
        for k, v in sorted(acc.items()):
            print(f'  {k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})')
PY
cat gpurun_out/c29_pmc_sq_summary.txt
find gpurun_out/prof_r2_final gpurun_out/pmc_fetch_f gpurun_out/pmc_write_f gpurun_out/pmc_sq_final_* -type f ! -name "*stats*" -size +1M -delete 2>/dev/null
