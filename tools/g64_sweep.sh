#!/bin/bash
# sweep of the 64 x 64 grouped launch's admission rule in split-bf16 mode (MESM_G64_* knobs, gemm.hip: joins64)
out=gpurun_out/${1:-g64}; mkdir -p $out
run() { echo "== $*" | tee -a $out/g64_sweep.txt; env "$@" python3 bench.py --steps 200 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d['roofline']['achieved'])" | tee -a $out/g64_sweep.txt; }
B="MESM_GEMM_BF16X=6 MESM_GEMM_GROUP64=1 MESM_G64_TALL=1 MESM_G64_MINDIM=1024"
run MESM_GEMM_BF16X=0
run $B
run $B MESM_G64_JOIN_MF=0
run $B MESM_G64_JOIN_MF=50
run $B MESM_G64_JOIN_MF=150
run $B MESM_G64_JOIN_MF=400
run $B MESM_G64_MINDIM=512
run $B
