#!/bin/bash
# A/B on ONE device: bench the working tree against another revision's Python sources (same built library) in
# alternation.  Devices of the pool differ by several per cent, so numbers from two gpurun calls do not compare.
# usage (here): tools/ab_bench.sh prepare <rev>     -> ab_prev/ holds that revision's mesm_amd + bench.py
#       (GPU):  tools/ab_bench.sh run [rounds]
if [ "$1" = "prepare" ]; then
  rm -rf ab_prev && mkdir -p ab_prev
  git archive "$2" mesm_amd bench.py | tar -x -C ab_prev
  cp mesm_amd/libmesm_gfx950.so ab_prev/mesm_amd/
  echo "ab_prev = $2 (with the CURRENT library)"
  exit 0
fi
n=${2:-3}
for i in $(seq $n); do
  a=$(python3 ab_prev/bench.py --steps 30 --warmup 5 --cpu-steps 0 --no-extras --no-roofline 2>&1 | grep -o "timed region: [0-9.]* ms")
  b=$(python3 bench.py --steps 30 --warmup 5 --cpu-steps 0 --no-extras --no-roofline 2>&1 | grep -o "timed region: [0-9.]* ms")
  echo "round $i   prev: $a    current: $b"
done
