#!/bin/bash
# kernel trace of the headline loop for tools/native_in_timed.py (run on the GPU box from the repo root)
OUT=gpurun_out/prof_native
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/bench.log" 2>&1
F=$(find "$OUT/trace" -name '*kernel_trace.csv' | head -1)
python3 tools/native_in_timed.py "$F" > "$OUT/native_in_timed.txt" 2>&1
cat "$OUT/native_in_timed.txt"
rm -rf "$OUT/trace"
