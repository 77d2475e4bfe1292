#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_dp_gpu.py tests/test_gemm_variants_gpu.py -q 2>&1 | tail -15 ) > gpurun_out/c5_pytest.log 2>&1
( timeout 600 python scripts/bench_gemm_variants.py --mb 1024 --variants pp256,pp256_b1 2>&1 | tail -40 ) > gpurun_out/c5_bench1024.log 2>&1
for V in pp256 pp256_b1; do
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq_$V -- python $GRAFT_REPO_ROOT/scripts/pmc_gemm.py 8192 8192 8192 1 1 $V b 5 2>&1 | tail -2 ) > gpurun_out/c5_pmc_sq_$V.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_lds_$V -- python $GRAFT_REPO_ROOT/scripts/pmc_gemm.py 8192 8192 8192 1 1 $V b 5 2>&1 | tail -2 ) > gpurun_out/c5_pmc_lds_$V.log 2>&1
done
python - <<'PY' > gpurun_out/c5_pmc_summary.txt 2>&1
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_*_pp256*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'gemm_bf16_pp' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        print('##', d)
        for k, v in sorted(acc.items()):
            print(f'  {k:28s} {sum(v)/len(v):16.0f}  (n={len(v)})')
PY
find gpurun_out/pmc_sq_* gpurun_out/pmc_lds_* -type f -size +1M -delete 2>/dev/null
tail -6 gpurun_out/c5_pytest.log; cat gpurun_out/c5_bench1024.log; cat gpurun_out/c5_pmc_summary.txt
