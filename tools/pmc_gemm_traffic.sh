#!/bin/bash
# HBM traffic of the gate/up GEMM and the causal prefill attention (bench.py's `roofline.traffic`): separate rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE; --kernel-trace only) over tools/one_gemm_attn.py, run on the GPU box from the repo root.
set -e
OUT=gpurun_out/prof_r03_gemm
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --kernel-include-regex 'gemm256pp|attn_prefill' --output-format csv -d "$OUT/$C" -- python3 tools/one_gemm_attn.py > "$OUT/$C.log" 2>&1
done
find "$OUT" -name '*counter_collection.csv' | xargs ls -la
