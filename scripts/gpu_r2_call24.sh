#!/bin/bash
# full GPU suite on the current library (pp256 PLAIN epilogue, DPP reductions + prefetch in the norm kernels)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > gpurun_out/c24_pytest.log; tail -5 gpurun_out/c24_pytest.log
