#!/bin/bash
# Hardware-counter passes over bench.py (one rocprofv3 run per counter group; run on the GPU box).
# Usage: bash tools/collect_pmc.sh <out_dir>
set -e
OUT=${1:-gpurun_out/pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() {  # name, counters...
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 3 --warmup 1 --no-cpu > "$OUT/$name.log" 2>&1
}
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
ls -R "$OUT" | head -30
