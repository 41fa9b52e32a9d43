#!/bin/bash
# Samples GPU power / clocks / temperature (rocm-smi) while a command runs.  usage: power_trace.sh OUT CMD...
out=$1; shift
( while true; do rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done ) > "$out" &
smi=$!
"$@"
rc=$?
kill $smi
exit $rc
