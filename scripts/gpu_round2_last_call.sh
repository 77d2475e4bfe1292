#!/bin/bash
# The last GPU call of round 2 (about 10 GPU-minutes were left): validate and measure the occupancy work on the attention
# kernels (two-phase fused backward, capped forward), the LayerNorm backward and the RCCL one-rank exchange test.
# Every step writes under gpurun_out/ as it goes (a cut-off call keeps what finished) and is skipped when the deadline is near.
#   bash scripts/gpu_round2_last_call.sh [deadline_seconds]
DEADLINE=${1:-520}
T0=$(date +%s)
OUT=gpurun_out/r2_last
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
step t_attn_ln 40 200 python -m pytest tests/test_kernels_gpu.py -q -k "attention or ln_fwd_bwd or qkln"
step bench_attn 30 120 python scripts/bench_attn.py 10 1024
step bench_auto 90 240 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-stages
step t_engine 90 300 python -m pytest tests/test_engine_gpu.py -q -x
step t_dp 60 240 python -m pytest tests/test_dp_gpu.py -q
step bench_fused1 90 240 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-stages --no-profile --attn-bwd fused1
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
step rocprof 100 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-stages --no-profile
DB=$(find $OUT/prof -name "*.db" 2>/dev/null | head -1)
if [ -n "$DB" ]; then
    python scripts/rocpd_stats.py $DB "rocprofv3 --kernel-trace: bench.py --steps 2 --warmup 1 (round 2 last call)" 60 > $OUT/kernel_stats.txt 2>&1
    find $OUT/prof -name "*.db" -delete     # the database is large; the summary is what travels back
fi
find $OUT/prof -type f 2>/dev/null | head
step t_rest 120 400 python -m pytest tests/test_trainer_gpu.py tests/test_kernels_gpu.py -q -x -k "not attention and not ln_fwd_bwd and not loss_curve_1k"
echo "total $(( $(date +%s) - T0 )) s" | tee -a $OUT/steps.log
