#!/bin/bash
# A/B of the restructured MoE combine / scatter-sum kernels against the in-step averages of profiles/r2_kernel_stats_bench_final.txt
OUT=gpurun_out/r2_moe
mkdir -p $OUT
timeout 40 python -m pytest tests/test_kernels_gpu.py -q -k "moe" > $OUT/t_moe.log 2>&1; echo "tests rc=$?" | tee $OUT/steps.log; tail -2 $OUT/t_moe.log
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 60 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-stages --no-profile > $OUT/bench.log 2>&1
echo "bench rc=$?" | tee -a $OUT/steps.log
DB=$(find $OUT/prof -name "*.db" 2>/dev/null | head -1)
if [ -n "$DB" ]; then
    python scripts/rocpd_stats.py $DB "rocprofv3 --kernel-trace: bench.py --steps 2 --warmup 1 (MoE combine / scatter-sum restructured)" 40 > $OUT/kernel_stats.txt 2>&1
    find $OUT/prof -name "*.db" -delete
fi
grep -E "moe_|gather_rows" $OUT/kernel_stats.txt; grep '^{"metric"' $OUT/bench.log | cut -c1-200
