#!/bin/bash
# Copy what scripts/collect_profiles.sh left in OUTDIR into profiles/ under the round's names:
#   scripts/install_profiles.sh gpurun_out/r5/prof r05
set -e
src=${1:?collection directory}; tag=${2:?round tag, e.g. r05}
cp "$src/bench_line.json" "profiles/${tag}_bench_n8192_f64.json"
cp "$src/kernel_stats.csv" "profiles/${tag}_bench_n8192_f64_kernel_stats.csv"
cp "$src/last_step_summary.txt" "profiles/${tag}_bench_n8192_f64_last_step_summary.txt"
cp "$src/chain_timeline.txt" "profiles/${tag}_potrf_chain_timeline_batched.txt"
cp "$src/pmc_hbm.json" "profiles/${tag}_pmc_hbm_n8192_f64.json"
cp "$src/mfma_busy.json" "profiles/${tag}_pmc_mfma_busy_n8192_f64.json"
cp "$src/fetch_counter_collection.csv" "profiles/${tag}_pmc_fetch_size_counter_collection.csv"
cp "$src/write_counter_collection.csv" "profiles/${tag}_pmc_write_size_counter_collection.csv"
cp "$src/mfma_counter_collection.csv" "profiles/${tag}_pmc_mfma_counter_collection.csv"
cp "$src/lauum_three_ways.txt" "profiles/${tag}_lauum_three_ways.txt"
cp "$src/lauum_three_ways.json" "profiles/${tag}_lauum_three_ways.json"
cp "$src/stats_stdout.json" "profiles/${tag}_stats_pass_roofline_only.json"
cp "$src/mfma_stdout.json" "profiles/${tag}_counter_pass_roofline_only.json"
