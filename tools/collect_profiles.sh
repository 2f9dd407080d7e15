#!/bin/bash
# Collects the round's profile evidence on the GPU box (run through gpurun from the repo root):
#   1. two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) over tools/probe_trunk.py 7056 2 -> per-launch HBM bytes -> profiles/pmc_traffic.json
#   2. rocprofv3 --kernel-trace --stats over the default bench.py command (which reads that file for roofline.traffic)
# Each rocprofv3 run is wrapped in `timeout -k 10`; counters other than these two have hung the profiler on this pool.
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
TAG=${1:-r01}
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 tools/probe_trunk.py 7056 2 > "$OUT/pmc_fetch.log" 2>&1 || { echo "FETCH_SIZE pass failed"; tail -5 "$OUT/pmc_fetch.log"; exit 1; }
echo "FETCH_SIZE pass done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 tools/probe_trunk.py 7056 2 > "$OUT/pmc_write.log" 2>&1 || { echo "WRITE_SIZE pass failed"; tail -5 "$OUT/pmc_write.log"; exit 1; }
echo "WRITE_SIZE pass done"
python3 tools/make_pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" 40 7056 > "$OUT/pmc_traffic.log" 2>&1 && cp profiles/pmc_traffic.json "$OUT/pmc_traffic.json"
python3 tools/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" > "$OUT/${TAG}_pmc_hbm_traffic_b7056.txt" 2>&1 || true
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o bench -- python3 bench.py --steps 2 --warmup 1 > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err" || { echo "bench profile failed"; tail -5 "$OUT/bench.err"; exit 1; }
tail -1 "$OUT/${TAG}_bench.json" > "$OUT/${TAG}_bench.line" && mv "$OUT/${TAG}_bench.line" "$OUT/${TAG}_bench.json"
find "$OUT/bench" -type f | head -20; cp "$(find "$OUT/bench" -name "*kernel_stats.csv" | head -1)" "$OUT/${TAG}_bench_kernel_stats.csv"
echo "bench + kernel stats done"
# keep the merge-back small: the raw traces are large
rm -rf "$OUT/bench" ; find "$OUT/pmc_fetch" "$OUT/pmc_write" -name "*.csv" -size +8M -delete
cat "$OUT/pmc_traffic.log"; head -4 "$OUT/${TAG}_bench_kernel_stats.csv"; cat "$OUT/${TAG}_bench.json" | cut -c1-400
