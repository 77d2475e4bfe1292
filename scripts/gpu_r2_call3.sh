#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > gpurun_out/c3_pytest.log 2>&1
tail -15 gpurun_out/c3_pytest.log
