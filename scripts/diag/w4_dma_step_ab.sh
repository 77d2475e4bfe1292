cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
mb=${MB:-1024}
for round in 1 2; do for L in "$@"; do echo "== $L"; MICRODIT_LIB=$L timeout 600 python scripts/ab_step.py --microbatch $mb --steps 3 --rounds 1 v: 2>&1 | grep "round 0"; done; done | tee gpurun_out/w4_dma_step_ab_$mb.txt
