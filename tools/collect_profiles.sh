#!/bin/bash
# All profile passes behind profiles/ (run on the GPU box): a kernel trace of bench.py, then one
# rocprofv3 run per hardware-counter group over bench.py and over bench_kernels.py.
# Usage: bash tools/collect_profiles.sh <out_dir>
set -e
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 bench.py --steps 10 --warmup 3 --no-cpu --no-probe > "$OUT/trace.log" 2>&1
python3 tools/summarize_trace.py "$(ls "$OUT"/trace/*/*_kernel_trace.csv | head -1)" > "$OUT/bench_kernels_by_grid.txt"
cp "$(ls "$OUT"/trace/*/*_kernel_stats.csv | head -1)" "$OUT/bench_kernel_stats.csv"
run() {  # name, program, counters...
  name=$1; prog=$2; shift; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $prog > "$OUT/$name.log" 2>&1
}
run mfma "bench.py --steps 3 --warmup 1 --no-cpu --no-probe" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
run fetch "bench.py --steps 3 --warmup 1 --no-cpu --no-probe" FETCH_SIZE
run write "bench.py --steps 3 --warmup 1 --no-cpu --no-probe" WRITE_SIZE
run kfetch "bench_kernels.py" FETCH_SIZE
run kwrite "bench_kernels.py" WRITE_SIZE
python3 tools/summarize_pmc.py "$OUT"/mfma/*/*_counter_collection.csv "$OUT"/fetch/*/*_counter_collection.csv "$OUT"/write/*/*_counter_collection.csv > "$OUT/bench_pmc_summary.txt"
python3 tools/summarize_pmc.py "$OUT"/kfetch/*/*_counter_collection.csv "$OUT"/kwrite/*/*_counter_collection.csv > "$OUT/bench_kernels_pmc_summary.txt"
ls "$OUT"
python3 tools/make_traffic_json.py "$OUT" > "$OUT/roofline_traffic.json"
cp "$OUT/roofline_traffic.json" profiles/roofline_traffic.json   # what the bench run below reports as roofline.traffic
# the bandwidth-bound stages through the C ABI (the table of achieved GB/s per stage)
timeout -s KILL 240 python3 bench_kernels.py > "$OUT/bench_kernels.jsonl" 2> "$OUT/bench_kernels.err"
timeout -s KILL 240 python3 bench.py --steps 20 --warmup 3 > "$OUT/bench.log" 2>&1
grep "^{" "$OUT/bench.log" | tail -1 > "$OUT/bench.json"
# the hash-grid step by itself at the data-parallel and at the single-GPU batch, and the launch-by-launch timeline of the latter
bash tools/trace_ngp.sh 65536 "$OUT/ngp_65536" > /dev/null
bash tools/trace_ngp.sh 4096 "$OUT/ngp_4096" > /dev/null
cp "$OUT/ngp_65536/kernels_by_grid.txt" "$OUT/ngp_65536rays_kernels_by_grid.txt"
python3 tools/step_timeline.py "$(ls "$OUT"/ngp_4096/trace/*/*_kernel_trace.csv | head -1)" > "$OUT/ngp_4096rays_step_timeline.txt"
