#!/bin/bash
# Short GPU pass for an iteration: named GPU tests, same-device A/B against another tree, one kernel trace of the step.
# Usage: tools/quick_trace.sh <tag> "<pytest args>" ["<trees for ab3>"]
tag=$1; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
python3 -m pytest $2 -x -q -m gpu 2>&1 | tail -6 > $out/tests.txt
[ -n "$3" ] && bash tools/ab3.sh "$3" 3 > $out/ab.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o t -- python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-roofline --no-extras > $out/trace.log 2>&1
python3 tools/step_timeline.py $out/trace/t_kernel_trace.csv > $out/step_timeline.txt 2>&1
rm -rf $out/trace
tail -3 $out/tests.txt; tail -3 $out/ab.txt; tail -1 $out/step_timeline.txt
