#!/bin/bash
# rocm-smi power / clock samples while the same launches run on random data and on all-zero operands
# (tools/zero_data_clock_test.py): usage power_compare.sh OUTDIR
out=$1; mkdir -p $out
for mode in data zeros; do
  ( while true; do rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done ) > $out/smi_$mode.jsonl &
  smi=$!
  python3 tools/zero_data_clock_test.py --soak $mode > $out/rate_$mode.txt 2>/dev/null
  kill $smi
  sleep 2
done
