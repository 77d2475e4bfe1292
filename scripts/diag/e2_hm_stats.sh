cd "$GRAFT_REPO_ROOT"
F="--steps+2+--warmup+1+--no-cpu-baseline+--no-other-stages+--no-profile"
MD_QK_HEAD_MAJOR=0 bash scripts/gpu_call.sh e2p "stats:$F"
MD_QK_HEAD_MAJOR=1 bash scripts/gpu_call.sh e2h "stats:$F"
