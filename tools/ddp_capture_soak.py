"""Start the own-communicator capture worker N times (tests/ddp_capture_worker.py own-overlapped / own-inline) and
count the starts that end with a correct result: the robustness figure of the captured data-parallel modes
(the torch process group's watchdog aborted ~3 % of starts in round 2).  usage: ddp_capture_soak.py [N] > profiles/...txt"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for mode in ("own-overlapped", "own-inline"):
    ok = bad = 0
    for i in range(n // 2):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ddp_capture_worker.py"), mode],
                           env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=600)
        lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        good = False
        if lines:
            res = json.loads(lines[-1][7:])
            good = res["launch_log_tail"] == [5, 4, 3, 2, 1, 0] and res["loss_err"] < 1e-5 and res["grad_err"] < 1e-4
        ok += good
        bad += not good
        if not good:
            print("start %d of %s failed: rc %s\n%s" % (i, mode, r.returncode, r.stderr[-1500:]), flush=True)
    print("%s: %d of %d starts captured, replayed and matched the eager gradients" % (mode, ok, ok + bad), flush=True)
