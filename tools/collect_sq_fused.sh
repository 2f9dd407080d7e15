#!/bin/bash
# SQ counters of the fused dense-block kernels, two separate --pmc passes over tools/probe_trunk.py 7056 1
# (rocprofv3 --kernel-trace --pmc only; never combined with other trace domains).  Prints the LAST dispatch of each fused kernel.
set -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/sq_fused
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/p1" -- python3 tools/probe_trunk.py 7056 1 > "$OUT/p1.log" 2>&1 || { echo "pass 1 failed"; tail -5 "$OUT/p1.log"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/p2" -- python3 tools/probe_trunk.py 7056 1 > "$OUT/p2.log" 2>&1 || { echo "pass 2 failed"; tail -5 "$OUT/p2.log"; exit 1; }
{
  echo "# rocprofv3 --kernel-trace --pmc, two passes (SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY | SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES"
  echo "# SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE) over tools/probe_trunk.py 7056 1 (ESRGAN x4 trunk NB=2, bf16, 7056 patches 48x48); last dispatch of each kernel"
  for k in "chain2_kernel<5, 2, 4, 1, false" "chain2_kernel<5, 2, 4, 1, true" "chain2_kernel<3, 2, 2, 0" "conv1_stream_kernel"; do
    echo "## $k"
    python3 tools/pmc_kernel.py "$k" "$OUT/p1" "$OUT/p2"
  done
} > "$OUT/${TAG}_sq_fused_kernels.txt"
find "$OUT" -name "*.csv" -size +4M -delete
cat "$OUT/${TAG}_sq_fused_kernels.txt"
