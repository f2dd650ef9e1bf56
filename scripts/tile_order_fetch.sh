#!/bin/bash
# FETCH_SIZE of the bulk update and of lauum_kernel in the default row order and in 8 x 8 supertile order (one rocprofv3
# --pmc pass each, program directly after `--`; the orders come from the environment defaults DGP_SYRK_ORDER /
# DGP_LAUUM_ORDER).  usage (through gpurun): bash scripts/tile_order_fetch.sh OUTDIR
set -o pipefail
out=${1:-gpurun_out/fetch_order}
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
export DGP_SPLIT_CHAIN=0
for v in rows super8; do
  if [ "$v" = super8 ]; then export DGP_SYRK_ORDER=8 DGP_LAUUM_ORDER=8; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/$v" -- python3 bench.py --roofline-only --steps 3 > "$out/$v.log" 2>&1 || { echo "FAILED $v"; exit 1; }
  echo "== $v (FETCH_SIZE in KiB as counted: x 2 x 1024 = bytes on gfx950; 3 steps)" >> "$out/summary.txt"
  python3 scripts/pmc_one.py "$out/$v" FETCH_SIZE | head -6 >> "$out/summary.txt"
  rm -rf "$out/$v"
done
cat "$out/summary.txt"
