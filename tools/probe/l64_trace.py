"""In-kernel phase timing of the lds64 GEMM kernel (trace build: tools/build_variant.sh trace -DMESM_L64_TRACE,
run with MESM_LIB_PATH=mesm_amd/variants/libmesm_trace.so; MESM_GEMM_TILE=3 (default) stamps the lds64 kernel,
MESM_GEMM_TILE=2 the wstage kernel, whose stamp slots have the same meaning).  usage: l64_trace.py M N K ta tb [split]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MESM_GEMM_TILE", "3")
import numpy as np
import torch
from mesm_amd import kernels as kn
from mesm_amd._lib import lib
M, N, K, ta, tb = [int(x) for x in sys.argv[1:6]]
split = int(sys.argv[6]) if len(sys.argv) > 6 else 1
dev = torch.device("cuda:0")
sets = [(torch.randn((K, M) if ta else (M, K), device=dev), torch.randn((N, K) if tb else (K, N), device=dev),
         torch.zeros(M, N, device=dev)) for _ in range(4)]
for i in range(6):   # the stamps of the LAST launch stay in the buffer
    A, B, C = sets[i % 4]
    kn.gemm(A, B, C, trans_a=bool(ta), trans_b=bool(tb), split_k=split)
torch.cuda.synchronize()
buf = np.zeros(1024 * 32, dtype=np.uint64)
rc = lib().mesm_l64_trace_read(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
t = buf.reshape(1024, 32)
nt = min(1024, ((M + 63) // 64) * ((N + 63) // 64))
t = t[:nt]
nst = min(8, (K // max(split, 1) + 31) // 32)
rel = t[:, :31].astype(np.int64) - t[:, 0:1].astype(np.int64)  # per workgroup, since its own entry (XCD clocks differ)
names = ["entry", "setup done", "first loads issued"]
for st in range(nst):
    names += ["k%d landed" % st, "k%d barrier" % st, "k%d mfma issued" % st]
cols = list(range(3 + 3 * nst)) + [28, 29, 30]
names += ["mfma drained", "stores issued", "stores acked"]
print("tiles %d, k-tiles stamped %d; cycles since the workgroup's own entry (median / p10 / p90 / max over workgroups)" % (nt, nst))
for c, nm in zip(cols, names):
    v = rel[:, c]
    print("%-20s med %8d  p10 %8d  p90 %8d  max %8d" % (nm, np.median(v), np.percentile(v, 10), np.percentile(v, 90), v.max()))
life = rel[:, 30] - rel[:, 0]
print("workgroup lifetime: med %d  p10 %d  p90 %d cycles" % (np.median(life), np.percentile(life, 10), np.percentile(life, 90)))
hw = t[:, 31]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xF
cu = ((hw & np.uint64(0xFFFFFFFF)).astype(np.int64) >> 8) & 0xF
se = ((hw & np.uint64(0xFFFFFFFF)).astype(np.int64) >> 13) & 0x7
key = xcc * 1000 + se * 100 + cu
uniq, cnt = np.unique(key, return_counts=True)
print("distinct (xcc,se,cu) %d; workgroups per CU: min %d max %d" % (len(uniq), cnt.min(), cnt.max()))
