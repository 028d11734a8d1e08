#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench line (with cpu_baseline), rocprofv3 kernel trace and
# the two HBM-traffic PMC passes.  Usage: tools/gpu_round.sh <tag>   (outputs in gpurun_out/<tag>/)
tag=${1:-r1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/summary.txt
python3 __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/summary.txt
python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/summary.txt
tail -1 $out/bench.json
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-roofline > $out/trace.log 2>&1; echo "trace rc=$?" | tee -a $out/summary.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/pmc_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-roofline > $out/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?" | tee -a $out/summary.txt
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/pmc_write -o w -- python3 bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-roofline > $out/pmc_write.log 2>&1; echo "pmc write rc=$?" | tee -a $out/summary.txt
ls -la $out $out/trace $out/pmc_fetch $out/pmc_write 2>/dev/null | head -40
du -sh $out
