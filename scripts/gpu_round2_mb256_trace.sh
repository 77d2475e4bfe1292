#!/bin/bash
# Kernel time against wall time at the per-rank shape of an 8-GPU run (microbatch 256): is the step launch-bound there?
OUT=gpurun_out/r2_mb256
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 100 rocprofv3 --kernel-trace --stats -d $OUT/prof -o bench -- python bench.py --microbatch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-other-stages --no-profile > $OUT/bench.log 2>&1
echo "rc=$?" >> $OUT/bench.log
DB=$(find $OUT/prof -name "*.db" 2>/dev/null | head -1)
if [ -n "$DB" ]; then
    python scripts/rocpd_stats.py $DB "rocprofv3 --kernel-trace --stats -- python bench.py --microbatch 256 --steps 2 --warmup 1 --no-cpu-baseline --no-other-stages --no-profile" 40 > $OUT/kernel_stats.txt 2>&1
    python - "$DB" > $OUT/gaps.txt 2>&1 <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select start, end from kernels order by start"))
# the last two thirds of the dispatches = the two timed steps (1 warm-up + 2 timed, equal work)
n = len(rows); rows = rows[n // 3:]
busy = sum(e - s for s, e in rows); span = rows[-1][1] - rows[0][0]
gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
pos = [g for g in gaps if g > 0]
print(f"dispatches {len(rows)}  span {span/1e6:.1f} ms  kernel time {busy/1e6:.1f} ms  idle between kernels {sum(pos)/1e6:.1f} ms ({100*sum(pos)/span:.1f} %)")
pos.sort()
print(f"gaps > 0: {len(pos)}  median {pos[len(pos)//2]/1e3:.2f} us  p90 {pos[int(len(pos)*0.9)]/1e3:.2f} us  p99 {pos[int(len(pos)*0.99)]/1e3:.2f} us  max {pos[-1]/1e3:.1f} us")
PY
    find $OUT/prof -name "*.db" -delete
fi
grep '^{"metric"' $OUT/bench.log | head -1; cat $OUT/gaps.txt
