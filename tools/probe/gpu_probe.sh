#!/bin/bash
# quick look at where one step's launches come from: tools/gpu_probe.sh <tag>
tag=${1:-probe}
out=gpurun_out/$tag
mkdir -p $out
python3 tools/count_kernels.py > $out/count_kernels.txt 2>&1
python3 tools/aten_origin.py > $out/aten_origin.txt 2>&1
python3 tools/attn_bench.py > $out/attn_bench.txt 2>&1
python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-extras > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json | cut -c1-400
