#!/bin/bash
# kernel trace of the file-fed eval loop alone (bench.py --eval-runner-only without the reuse part) + its idle gaps
set -e
OUT=gpurun_out/prof_evalrun
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export V3D_BENCH_SKIP_REUSE=1
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 bench.py --eval-runner-only --steps 16 --no-cpu-baseline > "$OUT/run.log" 2>&1
F=$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)
python3 tools/trace_gaps.py "$F" 3000 > "$OUT/gaps.txt"
cat "$OUT/gaps.txt"
grep -o '"eval_runner": {[^}]*}' "$OUT/run.log" | cut -c1-900
