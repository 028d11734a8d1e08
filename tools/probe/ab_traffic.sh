#!/bin/bash
# A/B of two library builds: step time and the FETCH_SIZE / WRITE_SIZE passes per kernel.
# usage: tools/ab_traffic.sh <tag> <libA or ""> <libB>   (outputs in gpurun_out/<tag>/)
tag=$1; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
i=0
for v in "$2" "$3"; do
  i=$((i+1)); echo "== build $i: ${v:-in-tree}" | tee -a $out/ab.txt
  export MESM_LIB_PATH=$v
  python3 bench.py --steps 200 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config'].get('settled_not_in_metric',{}).get('hip_event_median_ms_per_step'), d['roofline']['achieved'])" | tee -a $out/ab.txt
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out/pmc_${c}_$i -o p -- python3 bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-roofline --no-extras > $out/pmc_${c}_$i.log 2>&1
    python3 tools/pmc_summary.py $out/pmc_${c}_$i/p_counter_collection.csv 24 > $out/pmc_${c}_${i}_summary.txt 2>&1
    rm -rf $out/pmc_${c}_$i
    head -14 $out/pmc_${c}_${i}_summary.txt | cut -c1-150 | tee -a $out/ab.txt
  done
done
