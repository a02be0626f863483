#!/bin/bash
# rocprofv3 kernel stats of the cached-question workload (run on the GPU box from the repo root)
set -e
OUT=gpurun_out/prof_cached_q
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 tools/cached_q_profile.py 4 ${1:-16} > "$OUT/stats.log" 2>&1
find "$OUT/stats" -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} "$OUT/kernel_stats.csv"
tail -2 "$OUT/stats.log"
