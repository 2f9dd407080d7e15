#!/bin/bash
# cfg3 (the full ESRGAN _train_step at x4 / NB 23 / G 32 / 16 x 24^2) evidence, collected on the GPU box through gpurun from the repo root:
#   1. the un-profiled step time (tools/bench_train.py)            -> <tag>_cfg3_train_step.json
#   2. rocprofv3 --kernel-trace --stats over the same command      -> <tag>_cfg3_train_kernel_stats.csv
# (SKIP_CFG3=1 skips 1 and 2)
# plus, because the headline's stats file must cover real-data launches only, the bench.py stats collection of collect_profiles_r02.sh on its own.
set -o pipefail
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof3
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
TAG=${1:-r04}
if [ -z "$SKIP_CFG3" ]; then
python3 tools/bench_train.py 5 16 2>/dev/null | tail -1 > "$OUT/${TAG}_cfg3_train_step.json" || { echo "cfg3 step failed"; exit 1; }
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train" -o train -- python3 tools/bench_train.py 3 16 > "$OUT/train.log" 2>&1 || { echo "cfg3 profile failed"; tail -5 "$OUT/train.log"; exit 1; }
cp "$(find "$OUT/train" -name "*kernel_stats.csv" | head -1)" "$OUT/${TAG}_cfg3_train_kernel_stats.csv"
rm -rf "$OUT/train"
echo "cfg3 done"
fi
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o bench -- python3 bench.py --steps 2 --warmup 1 --no-rows --no-parity --no-zero-step > "$OUT/${TAG}_bench.tmp" 2> "$OUT/bench.err" || { echo "bench profile failed"; tail -5 "$OUT/bench.err"; exit 1; }
tail -1 "$OUT/${TAG}_bench.tmp" > "$OUT/${TAG}_bench.json"; rm -f "$OUT/${TAG}_bench.tmp"
cp "$(find "$OUT/bench" -name "*kernel_stats.csv" | head -1)" "$OUT/${TAG}_bench_kernel_stats.csv"
rm -rf "$OUT/bench"
echo "bench + kernel stats done"
[ -z "$SKIP_CFG3" ] && { cut -c1-400 "$OUT/${TAG}_cfg3_train_step.json"; head -8 "$OUT/${TAG}_cfg3_train_kernel_stats.csv" | cut -c1-200; }; cut -c1-600 "$OUT/${TAG}_bench.json"; head -5 "$OUT/${TAG}_bench_kernel_stats.csv" | cut -c1-200
