#!/bin/bash
# Same-box A/B of w4 k-loop schedules: scripts/ab_w4.sh <variants-arg of bench_w4.py> lib1.so lib2.so ...  (two rounds, A B A B)
v=$1; shift
for round in 1 2; do for lib in "$@"; do MICRODIT_LIB=$lib timeout 120 python scripts/bench_w4.py "$v" ${W4_BKC:-1} 2>&1 | grep "|"; done; done
