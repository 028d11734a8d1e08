#!/bin/bash
# sweep of the 64 x 64 grouped launch's admission rule in split-bf16 mode (MESM_G64_* knobs, gemm.hip: joins64)
out=gpurun_out/${1:-g64}; mkdir -p $out
run() { echo "== $*" | tee -a $out/g64_sweep.txt; env "$@" python3 bench.py --steps 200 --warmup 30 --cpu-steps 0 --no-extras 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['gemm_ms_per_step'], d['roofline']['achieved'])" | tee -a $out/g64_sweep.txt; }
run MESM_GEMM_BF16X=6
run MESM_G64_MINDIM=512
run MESM_G64_MINDIM=2048
run MESM_G64_MINB64=64
run MESM_G64_MINB64=256
run MESM_G64_MINDIM=256 MESM_G64_MINB64=32
run MESM_LIB_PATH=mesm_amd/variants/libmesm_gmax12.so
run MESM_LIB_PATH=mesm_amd/variants/libmesm_gmax16.so
run MESM_GEMM_SPLIT_ROWS=0
run MESM_GEMM_BF16X=6
