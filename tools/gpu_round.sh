#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench line (with cpu_baseline), rocprofv3 kernel trace and
# the two HBM-traffic PMC passes.  Usage: tools/gpu_round.sh <tag> [quick]  (outputs in gpurun_out/<tag>/)
tag=${1:-r1}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
if [ "$2" != "quick" ]; then
python3 -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/summary.txt
python3 __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/summary.txt
fi
python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/summary.txt
tail -1 $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-roofline --no-extras > $out/trace.log 2>&1; echo "trace rc=$?" | tee -a $out/summary.txt
python3 tools/trace_summary.py $out/trace/t_kernel_trace.csv 100 > $out/trace_summary.txt 2>&1
python3 tools/step_timeline.py $out/trace/t_kernel_trace.csv > $out/step_timeline.txt 2>&1
rm -f $out/trace/t_kernel_trace.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o p -- python3 bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-roofline --no-extras > $out/pmc_$c.log 2>&1; echo "pmc $c rc=$?" | tee -a $out/summary.txt
  python3 tools/pmc_summary.py $out/pmc_$c/p_counter_collection.csv 60 > $out/pmc_${c}_summary.txt 2>&1
done
# MFMA utilisation (its own pass: counters never share a run with the trace domains)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $out/pmc_mfma -o p -- python3 bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-roofline --no-extras > $out/pmc_mfma.log 2>&1; echo "pmc mfma rc=$?" | tee -a $out/summary.txt
python3 tools/pmc_mfma.py $out/pmc_mfma/p_counter_collection.csv 30 > $out/pmc_mfma_summary.txt 2>&1
rm -rf $out/pmc_mfma
python3 tools/pmc_traffic.py $out/pmc_FETCH_SIZE/p_counter_collection.csv $out/pmc_WRITE_SIZE/p_counter_collection.csv $out/gemm_traffic.json 7 $out/bench.json $tag
for c in FETCH_SIZE WRITE_SIZE; do rm -f $out/pmc_$c/p_counter_collection.csv $out/pmc_$c/p_kernel_trace.csv; done
python3 tools/ablate.py > $out/ablation.txt 2>&1
python3 tools/attn_bench.py > $out/attn_bench.txt 2>&1
python3 tools/gemm_census.py > $out/gemm_census.txt 2>&1
python3 tools/load_batch_probe.py > $out/load_batch.txt 2>&1
for w in C2 C3b C5; do python3 bench.py --workload $w --steps 10 --warmup 3 --cpu-steps 0 --no-roofline --no-extras 2>/dev/null | python3 -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"$w %.3f ms/step %.0f pairs/s\" % (l[\"ms_per_step\"], l[\"value\"]))" >> $out/other_workloads.txt; done
python3 tools/ln_bench.py > $out/ln_bench.txt 2>&1
python3 tools/branch_cost.py > $out/branch_cost.txt 2>&1
python3 tools/bf16x_check.py > $out/bf16x_check.txt 2>&1
python3 tools/gemm_plan.py 2> $out/gemm_plan.txt > /dev/null
python3 tools/host_cost.py > $out/host_cost.txt 2>&1
python3 tools/tape_profile.py > $out/tape_profile.txt 2>&1
python3 tools/aten_origin.py > $out/aten_origin.txt 2>&1
python3 tools/eager_profile.py > $out/eager_profile.txt 2>&1
[ -d ab_r4 ] && tools/ab3.sh "ab_r4 ." 3 > $out/ab_vs_r4.txt 2>&1
find $out -type f | xargs ls -la | head -40
du -sh $out
