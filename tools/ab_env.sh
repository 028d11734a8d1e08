#!/bin/bash
# same-device comparison of environment settings on the working tree: tools/ab_env.sh "VAR=a VAR=b X=1:Y=2 ..." [rounds]
# (several variables of one setting are joined by ':')
n=${2:-3}
for i in $(seq $n); do
  line="round $i"
  for e in $1; do
    t=$(env ${e//:/ } python3 bench.py --steps 30 --warmup 5 --cpu-steps 0 --no-extras --no-roofline 2>&1 | grep -o "timed region: [0-9.]*")
    line="$line   $e: ${t#timed region: }"
  done
  echo "$line"
done
