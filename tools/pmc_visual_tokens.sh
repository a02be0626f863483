#!/bin/bash
# PMC passes for the north-star kernel (visual_tokens_kernel) and its neighbours: separate --pmc runs for FETCH_SIZE and
# WRITE_SIZE (MI355X_MICROARCH.md, HBM / rocprofv3 section), kernel-trace only.  Run on the GPU box from the repo root:
#   bash tools/pmc_visual_tokens.sh gpurun_out/pmc_r02
set -e
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --kernel-include-regex 'visual_tokens|coord_pool|unproject_sampled' --output-format csv -d "$OUT/$C" -- \
    python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > "$OUT/$C.log" 2>&1
  find "$OUT/$C" -name '*counter_collection.csv' | head -1 | xargs -I{} cp {} "$OUT/${C}_counters.csv"
done
ls -la "$OUT"
