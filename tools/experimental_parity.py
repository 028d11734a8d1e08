"""The GPU parity suite under the OTHER GEMM arithmetics -- MESM_GEMM_BF16X = 0 (every product on the f32 MFMA instruction)
and 3 (EXPERIMENTAL two-term split) -- tolerances unchanged; the default (6: three-term split, six products) is what the
plain `pytest -m gpu` run covers.  Writes {mode: {passed, failed, failed_tests}} to the given JSON file (committed under
profiles/, quoted by bench.py's roofline.exact_f32 / roofline.experimental).  usage: python tools/experimental_parity.py out.json"""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for mode in ("0", "3"):
    env = dict(os.environ, MESM_GEMM_BF16X=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "-q", "--tb=no", "-rf", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, capture_output=True, text=True)
    txt = r.stdout + r.stderr
    m = re.search(r"(\d+) passed", txt)
    f = re.search(r"(\d+) failed", txt)
    failed = sorted(set(re.findall(r"^FAILED (\S+)", txt, re.M)))
    out[{"0": "exact_f32", "3": "bf16x3"}[mode]] = {"passed": int(m.group(1)) if m else 0, "failed": int(f.group(1)) if f else 0,
                           "failed_tests": failed, "tolerances": "unchanged (1e-4 logits / losses, bit-exact matcher, 5e-4 "
                           "kink-free gradients)"}
    k = {"0": "exact_f32", "3": "bf16x3"}[mode]
    print("%s: %s passed, %s failed" % (k, out[k]["passed"], out[k]["failed"]), flush=True)
with open(sys.argv[1], "w") as fh:
    json.dump(out, fh, indent=1)
