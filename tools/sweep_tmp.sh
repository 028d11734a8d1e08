for v in "" variants/libmesm_st2.so variants/libmesm_st2w5.so variants/libmesm_st4.so; do
echo "== $v"
if [ -n "$v" ]; then export MESM_LIB_PATH=mesm_amd/$v; else unset MESM_LIB_PATH; fi
python3 tools/gemm_sweep.py 4800:1024:256:0:1:1:3:0 4800:1024:256:0:0:1:3:0 4800:256:1024:0:1:1:3:0 1024:5003:256:0:1:1:3:0 2400:2818:256:0:0:1:3:0 2>&1 | grep -v amdgpu
done
