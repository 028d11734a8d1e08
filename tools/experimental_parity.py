"""The GPU parity suite under the EXPERIMENTAL split-bf16 GEMM modes (MESM_GEMM_BF16X = 6 | 3), tolerances unchanged:
writes {mode: {passed, failed, failed_tests}} to the given JSON file (committed under profiles/, quoted by bench.py's
roofline.experimental).  usage: python tools/experimental_parity.py out.json"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for mode in ("6", "3"):
    env = dict(os.environ, MESM_GEMM_BF16X=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "-q", "--tb=no", "-rf", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True)
    txt = r.stdout + r.stderr
    m = re.search(r"(\d+) passed", txt)
    f = re.search(r"(\d+) failed", txt)
    failed = sorted(set(re.findall(r"^FAILED (\S+)", txt, re.M)))
    out["bf16x" + mode] = {"passed": int(m.group(1)) if m else 0, "failed": int(f.group(1)) if f else 0,
                           "failed_tests": failed, "tolerances": "unchanged (1e-4 logits / losses, bit-exact matcher, 5e-4 "
                           "kink-free gradients); only GEMMs with >= 2400 output rows or reduce indices take the split path"}
    print("bf16x%s: %s passed, %s failed" % (mode, out["bf16x" + mode]["passed"], out["bf16x" + mode]["failed"]), flush=True)
with open(sys.argv[1], "w") as fh:
    json.dump(out, fh, indent=1)
