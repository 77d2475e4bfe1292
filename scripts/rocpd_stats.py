"""Per-kernel stats table (the --stats view) of a rocprofv3 --kernel-trace run.
    python scripts/rocpd_stats.py <rocpd .db | output directory with *kernel_trace.csv> [title] [rows]"""
import csv
import glob
import os
import re
import sqlite3
import sys
from collections import defaultdict

src = sys.argv[1]
title = sys.argv[2] if len(sys.argv) > 2 else ""
top = int(sys.argv[3]) if len(sys.argv) > 3 else 48
agg = defaultdict(list)          # name -> [durations ns]
spans = []                       # (start, end)
if os.path.isdir(src):
    files = glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
    dbs = glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
    if files:
        for f in files:
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                    agg[r["Kernel_Name"]].append(e - s)
                    spans.append((s, e))
    elif dbs:
        src = dbs[0]
if not agg:
    db = sqlite3.connect(src)
    for n, s, e in db.execute("select name, start, end from kernels"):
        agg[n].append(e - s)
        spans.append((s, e))


def clean(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\((md_|__hip|float|int|long|void|unsigned|at::|c10::).*$", "", n)
    return n[:78]


rows = sorted(((n, len(v), sum(v), min(v), max(v)) for n, v in agg.items()), key=lambda r: -r[2])
tot = sum(r[2] for r in rows)
print(f"# {title}\n# total kernel time {tot / 1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches")
print(f"{'kernel':78s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'pct':>6s}")
for n, c, s, mi, ma in rows[:top]:
    print(f"{clean(n):78s} {c:7d} {s / 1e6:10.2f} {s / c / 1e3:9.1f} {mi / 1e3:8.1f} {ma / 1e3:9.1f} {100 * s / tot:6.2f}")
# classes
cls = defaultdict(float)
for n, c, s, mi, ma in rows:
    k = clean(n)
    key = ("gemm" if k.startswith("gemm_bf16") else "splitk_reduce" if "splitk" in k else "attention" if k.startswith("attn_") else
           "qk_layernorm" if k.startswith("qkln") else "layernorm" if k.startswith("ln_") else "moe routing/combine" if k.startswith("moe_") or "gather" in k or "scatter" in k else
           "swiglu" if "swiglu" in k else "gate_bwd" if "gate_bwd" in k else "adamw/norm" if "adamw" in k or "sumsq" in k else "other")
    cls[key] += s
print("# classes: " + "  ".join(f"{k} {100 * v / tot:.1f}%" for k, v in sorted(cls.items(), key=lambda kv: -kv[1])))
if spans:
    spans.sort()
    span = spans[-1][1] - spans[0][0]
    idle, cur = 0, spans[0][1]
    for s, e in spans[1:]:
        if s > cur:
            idle += s - cur
        cur = max(cur, e)
    print(f"# first to last kernel of the process {span / 1e6:.1f} ms, of which no kernel running {idle / 1e6:.1f} ms (includes model build and "
          f"host-side set-up between the legs: not a measure of launch gaps inside a step)")
