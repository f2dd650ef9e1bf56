#!/bin/bash
# Collect the round's profiles on the GPU box (run from the repo root through gpurun):
#   scripts/collect_profiles.sh OUTDIR COMMIT
# One rocprofv3 pass per counter set, the program directly after `--`.  Kernel DURATIONS are taken from the `stats`
# pass only (--kernel-trace --stats, no counters); the --pmc passes also carry --kernel-trace (needed to attribute the
# counters to kernels; never any other trace domain) and serialise the streams, so their durations are PERTURBED and
# are not reported as timings anywhere.
set -o pipefail
out=${1:-gpurun_out/prof}; commit=${2:-unknown}
: "${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
# every pass keeps its program's stdout (the --roofline-only JSON line: the process's OWN HIP-event times of the dominant kernel
# and its in-kernel shader clock) next to the tracer's files: scripts/roofline_check.py puts them side by side
run() { name=$1; shift; echo "== $name" >> "$out/log.txt"; timeout -k 10 400 "$@" > "$out/${name}_stdout.txt" 2>> "$out/log.txt" || { echo "FAILED: $name" | tee -a "$out/log.txt"; return 1; }; grep '^{' "$out/${name}_stdout.txt" | tail -1 > "$out/${name}_stdout.json"; cat "$out/${name}_stdout.txt" >> "$out/log.txt"; }
run stats  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --roofline-only --steps 4 || exit 1
# Counter passes serialise dispatches in queue-ready order: a kernel that polls a word another kernel publishes (the
# split panel chain of SINGLE-SITE plans, dgp_chol.hip::chain_wait) could be granted before its producer and would then
# sit out its 2 s time-out.  The profiled command runs the batched plan only, which has no such wait -- the switch is
# exported anyway so that no counter pass ever depends on that.
export DGP_SPLIT_CHAIN=0
run mfma   rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/mfma" -- python3 bench.py --roofline-only --steps 3 --no-clock-probe || exit 1
run fetch  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 bench.py --roofline-only --steps 3 --no-clock-probe || exit 1
run write  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 bench.py --roofline-only --steps 3 --no-clock-probe || exit 1
# the un-profiled line of the SAME lease, last (stats pass, MFMA / GRBM pass and the committed line in one
# lease, in that order); BENCH_ARGS e.g. "--no-cpu-baseline --no-configs" for a quick collection
unset DGP_SPLIT_CHAIN
echo "== bench" >> "$out/log.txt"
timeout -k 10 900 python3 bench.py ${BENCH_ARGS:-} > "$out/bench_line.json" 2>> "$out/log.txt" || { echo "FAILED: bench" | tee -a "$out/log.txt"; exit 1; }
ok=1
python3 scripts/pmc_summary.py "$out/fetch" "$out/write" "$out/pmc_hbm.json" 3 32 "$commit" > "$out/pmc_hbm.txt" 2>&1 || ok=0
python3 scripts/mfma_util.py "$out/mfma" "$out/mfma_busy.json" > "$out/mfma_busy.txt" 2>&1 || ok=0
python3 scripts/trace_summary.py "$out/stats" > "$out/last_step_summary.txt" 2>&1 || ok=0
python3 scripts/chain_timeline.py "$(ls $out/stats/*/*_kernel_trace.csv | head -1)" > "$out/chain_timeline.txt" 2>&1 || ok=0
cp "$(ls $out/stats/*/*_kernel_stats.csv | head -1)" "$out/kernel_stats.csv" || ok=0
cp "$(ls $out/fetch/*/*_counter_collection.csv | head -1)" "$out/fetch_counter_collection.csv" || ok=0
cp "$(ls $out/write/*/*_counter_collection.csv | head -1)" "$out/write_counter_collection.csv" || ok=0
cp "$(ls $out/mfma/*/*_counter_collection.csv | head -1)" "$out/mfma_counter_collection.csv" || ok=0
python3 scripts/roofline_check.py "$out" "$(ls $out/stats/*/*_kernel_trace.csv | head -1)" > "$out/lauum_three_ways.txt" 2>&1 || ok=0
if [ "$ok" != 1 ]; then echo "a summary step failed: raw rocprofv3 output kept under $out" | tee -a "$out/log.txt"; exit 1; fi
rm -rf "$out/stats" "$out/fetch" "$out/write" "$out/mfma"
tail -3 "$out/pmc_hbm.txt"; head -6 "$out/mfma_busy.txt"; cat "$out/lauum_three_ways.txt"
