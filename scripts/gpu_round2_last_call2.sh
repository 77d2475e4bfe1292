#!/bin/bash
# Second (final) GPU call of round 2, ~5 GPU-minutes: the measured dispatch rule of the attention backward (single-phase for the
# 64 x 64 buckets, two-phase elsewhere) and the forward without the register cap; the round's final bench line and kernel stats.
DEADLINE=${1:-285}
T0=$(date +%s)
OUT=gpurun_out/r2_last2
mkdir -p $OUT
left() { echo $(( DEADLINE - ($(date +%s) - T0) )); }
step() {   # step <name> <min seconds needed> <timeout> <command...>
    local name=$1 need=$2 tmo=$3; shift 3
    local l=$(left)
    if [ $l -lt $need ]; then echo "SKIP $name (only $l s left)" | tee -a $OUT/steps.log; return; fi
    [ $tmo -gt $l ] && tmo=$l
    local s=$(date +%s)
    timeout $tmo "$@" > $OUT/$name.log 2>&1
    local rc=$?
    echo "$name rc=$rc $(( $(date +%s) - s )) s (t+$(( $(date +%s) - T0 )) s)" | tee -a $OUT/steps.log
    tail -3 $OUT/$name.log
}
export HSA_ENABLE_IPC_MODE_LEGACY=0
step t_attn 20 120 python -m pytest tests/test_kernels_gpu.py -q -k "attention"
step bench_attn 20 60 python scripts/bench_attn.py 10 1024
step bench_final 100 230 python bench.py
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
step rocprof 40 120 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-stages --no-profile
DB=$(find $OUT/prof -name "*.db" 2>/dev/null | head -1)
if [ -n "$DB" ]; then
    python scripts/rocpd_stats.py $DB "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-stages --no-profile (round 2 final: measured attention-backward rule)" 60 > $OUT/kernel_stats.txt 2>&1
    find $OUT/prof -name "*.db" -delete
fi
step t_engine_small 60 200 python -m pytest tests/test_engine_gpu.py -q -x -k "not xl2"
echo "total $(( $(date +%s) - T0 )) s" | tee -a $OUT/steps.log
