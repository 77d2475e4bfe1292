"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel stats table (the --stats view)."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
title = sys.argv[2] if len(sys.argv) > 2 else ""
rows = list(db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
def clean(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\((md_|__hip|float|int|long|void|unsigned|at::|c10::).*$", "", n)
    return n[:78]
print(f"# {title}\n# total kernel time {tot/1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches")
print(f"{'kernel':78s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>9s} {'pct':>6s}")
for n, c, s, a, mi, ma in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print(f"{clean(n):78s} {c:7d} {s/1e6:10.2f} {a/1e3:9.1f} {mi/1e3:8.1f} {ma/1e3:9.1f} {100*s/tot:6.2f}")
