#!/bin/bash
# Kernel trace of the hash-grid training step at a given batch (run on the GPU box): bash tools/trace_ngp.sh <rays> <out_dir>
set -e
RAYS=${1:-65536}
OUT=${2:-gpurun_out/ngp_trace}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --model ngp --rays "$RAYS" --steps 10 --warmup 3 --no-cpu > "$OUT/trace.log" 2>&1
python3 tools/summarize_trace.py "$(ls "$OUT"/trace/*/*_kernel_trace.csv | head -1)" > "$OUT/kernels_by_grid.txt"
tail -1 "$OUT/trace.log" | cut -c1-300
cat "$OUT/kernels_by_grid.txt"
