#!/bin/bash
# usage: scripts/clock_watch.sh OUT.log -- command ...   polls rocm-smi (sclk, power) every 0.2 s while the command runs
out=$1; shift; shift
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket" | tr '\n' ' ' ; echo; sleep 0.2; done ) > "$out" &
w=$!
"$@"
rc=$?
kill $w
exit $rc
