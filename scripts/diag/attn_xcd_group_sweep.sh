#!/bin/bash
# sweep of the group size G of the (batch, head) -> XCD dealing (scratch_libs/lib_xcdg.so reads MD_ATTN_XCD_G; 1 = plain grid order, 0 = contiguous eighths)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for round in 1 2; do
  for g in 1 2 4 8 16 32 64 128 256 512 1024 0; do
    echo "## G=$g (round $round)"
    MD_ATTN_XCD_G=$g MICRODIT_LIB=scratch_libs/lib_xcdg.so timeout 200 python scripts/bench_attn.py 20 1024 2>&1 | grep "fwd\|bwd auto" | grep -v "S=1024\|1024x77"
  done
done > gpurun_out/x7_attn_xcd_group_sweep.txt 2>&1
python - <<'PY'
import re, collections
t = collections.defaultdict(lambda: collections.defaultdict(list)); g = None
for line in open("gpurun_out/x7_attn_xcd_group_sweep.txt"):
    m = re.match(r"## G=(\d+)", line)
    if m: g = int(m.group(1)); continue
    m = re.match(r"(.*?)\s+(fwd|bwd auto)\s*:\s*([\d.]+) us", line)
    if m: t[m.group(1).strip() + " " + m.group(2)][g].append(float(m.group(3)))
gs = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 0]
print("%-36s" % "us (mean of 2 rounds); G =", " ".join("%7s" % (x if x else "n/8") for x in gs))
for k, d in t.items():
    print("%-36s" % k, " ".join("%7.1f" % (sum(d[x]) / len(d[x])) if d[x] else "      -" for x in gs))
PY
