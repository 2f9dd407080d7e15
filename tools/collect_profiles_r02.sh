#!/bin/bash
# Round-2 profile evidence, collected on the GPU box through gpurun from the repo root:
#   1. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE; never combined with other trace domains) over tools/probe_trunk.py 7056 2
#      -> per-launch HBM bytes of the conv kernels, the fused dense-block kernels among them -> profiles/pmc_traffic.json
#   2. rocprofv3 --kernel-trace --stats over the default bench.py command (which reads that file for roofline.traffic)
#   3. the N=1 bench at 2 / 4 / 8 / 16 tiles (what a rank sees at N = 8 / 4 / 2 / 1 under the strong-scaling partition of SURVEY.md 8e)
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof2
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
TAG=${1:-r02}
NDISP=${2:-31}
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 tools/probe_trunk.py 7056 2 > "$OUT/pmc_fetch.log" 2>&1 || { echo "FETCH_SIZE pass failed"; tail -5 "$OUT/pmc_fetch.log"; exit 1; }
echo "FETCH_SIZE pass done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 tools/probe_trunk.py 7056 2 > "$OUT/pmc_write.log" 2>&1 || { echo "WRITE_SIZE pass failed"; tail -5 "$OUT/pmc_write.log"; exit 1; }
echo "WRITE_SIZE pass done"
python3 tools/make_pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" $NDISP 7056 > "$OUT/pmc_traffic.log" 2>&1 && cp profiles/pmc_traffic.json "$OUT/pmc_traffic.json"
python3 tools/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" > "$OUT/${TAG}_pmc_hbm_traffic_b7056.txt" 2>&1 || true
# the profiled command carries only full-size launches of the hot kernels (no parity patches, no cfg3 / cfg4 rows), so that the per-kernel averages of
# the stats file are directly the ones bench.py's HIP events report; the complete default line (parity, cpu_baseline, rows) is taken un-profiled below
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o bench -- python3 bench.py --steps 2 --warmup 1 --no-rows --no-parity --no-zero-step > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err" || { echo "bench profile failed"; tail -5 "$OUT/bench.err"; exit 1; }
tail -1 "$OUT/${TAG}_bench.json" > "$OUT/${TAG}_bench.line" && mv "$OUT/${TAG}_bench.line" "$OUT/${TAG}_bench.json"
cp "$(find "$OUT/bench" -name "*kernel_stats.csv" | head -1)" "$OUT/${TAG}_bench_kernel_stats.csv"
echo "bench + kernel stats done"
python3 bench.py > "$OUT/${TAG}_bench_default.tmp" 2> "$OUT/bench_default.err" && tail -1 "$OUT/${TAG}_bench_default.tmp" > "$OUT/${TAG}_bench_default.json"; rm -f "$OUT/${TAG}_bench_default.tmp"
echo "default bench line done"
for t in 2 4 8 16; do
  python3 bench.py --tiles $t --tiles-per-call $t --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-profile --no-rows 2>/dev/null | tail -1 >> "$OUT/${TAG}_tiles_sweep.jsonl" || echo "tiles $t failed"
done
echo "tiles sweep done"
# DVFS diagnostic (MI355X_MICROARCH.md "DVFS give-back" item 1): the same launches on all-zero weights and tiles -- what the kernels' times become when no operand bit toggles
python3 bench.py --zero-data --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_zero_data.json" || echo "zero-data run failed"
echo "zero-data run done"
rm -rf "$OUT/bench" ; find "$OUT/pmc_fetch" "$OUT/pmc_write" -name "*.csv" -size +8M -delete
cat "$OUT/pmc_traffic.log"; head -6 "$OUT/${TAG}_bench_kernel_stats.csv"; cut -c1-300 "$OUT/${TAG}_bench.json"; python3 -c "
import json
for l in open('$OUT/${TAG}_tiles_sweep.jsonl'): d=json.loads(l); print(d['config']['tiles_this_rank'], round(d['value'],2), round(d['ms_per_step'],1))"
