#!/bin/bash
# same-device comparison of several trees: tools/ab3.sh "<dir> <dir> ..." [rounds]   ('.' = the working tree)
n=${2:-2}
for i in $(seq $n); do
  line="round $i"
  for d in $1; do
    t=$(python3 $d/bench.py --steps 30 --warmup 5 --cpu-steps 0 --no-extras --no-roofline 2>&1 | grep -o "timed region: [0-9.]*")
    line="$line   $d: ${t#timed region: }"
  done
  echo "$line"
done
