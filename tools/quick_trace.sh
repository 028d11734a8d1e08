#!/bin/bash
# Kernel trace of a short bench run, summarised (no PMC): tools/quick_trace.sh <tag> [grep pattern]
tag=${1:-q}; pat=${2:-.}
out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-roofline --no-extras > $out/trace.log 2>&1
python3 tools/trace_summary.py $out/trace/t_kernel_trace.csv 100 > $out/trace_summary.txt 2>&1
python3 tools/step_timeline.py $out/trace/t_kernel_trace.csv > $out/step_timeline.txt 2>&1
rm -f $out/trace/t_kernel_trace.csv
head -1 $out/trace_summary.txt; grep -E "$pat" $out/trace_summary.txt | cut -c1-150
