#!/bin/bash
# GPU box: the whole -m gpu suite, full report into gpurun_out/gputest_full.log
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q --tb=short -rA "$@" > gpurun_out/gputest_full.log 2>&1
grep -E "passed|failed" gpurun_out/gputest_full.log | tail -3
grep -E "^(FAILED|ERROR)" gpurun_out/gputest_full.log | head -20
