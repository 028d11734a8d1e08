#!/bin/bash
# A/B of environment switches on ONE device: tools/ab_env.sh "<VAR=val ...>" [rounds]; prints ms/step without / with them
n=${2:-3}
for i in $(seq $n); do
  a=$(python3 bench.py --steps 30 --warmup 5 --cpu-steps 0 --no-extras --no-roofline 2>&1 | grep -o "timed region: [0-9.]* ms")
  b=$(env $1 python3 bench.py --steps 30 --warmup 5 --cpu-steps 0 --no-extras --no-roofline 2>&1 | grep -o "timed region: [0-9.]* ms")
  echo "round $i   default: $a    with $1: $b"
done
