"""Shader clock and socket power while ONE GEMM shape runs back to back (is the split kernel's loop power-bound?):
rocm-smi is sampled from a thread during ~2 s of launches per mode.  usage: python tools/clock_probe.py"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mesm_amd import kernels as kn
dev = torch.device("cuda:0")
M, N, K = 8192, 256, 1024
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev); C = torch.zeros(M, N, device=dev)


def sample(stop, out):
    while not stop.is_set():
        try:
            t = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", t)
            p = re.search(r"Power \(W\): ([\d.]+)", t) or re.search(r"Socket Graphics Package Power \(W\): ([\d.]+)", t)
            out.append((int(m.group(1)) if m else None, float(p.group(1)) if p else None))
        except Exception as e:
            out.append((None, None))
        time.sleep(0.05)


for name, tile, bf in (("idle", None, None), ("f32 MFMA k-split 64x64", 4, 0), ("split-bf16 k-split 64x64", 4, 6)):
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    t0 = time.perf_counter(); n = 0
    if tile is None:
        time.sleep(2.0)
    else:
        kn.gemm_switches(tile=tile, bf16x=bf)
        while time.perf_counter() - t0 < 2.5:
            for _ in range(200):
                kn.gemm(A, B, C, trans_b=True)
            torch.cuda.synchronize(); n += 200
    dt = time.perf_counter() - t0
    stop.set(); th.join()
    clk = [c for c, _ in out if c]; pw = [p for _, p in out if p]
    print("%-26s %s sclk MHz min/med/max %s  power W med %s  samples %d" % (
        name, ("%.1f us/launch" % (dt / n * 1e6)) if n else "", (min(clk), sorted(clk)[len(clk) // 2], max(clk)) if clk else None,
        sorted(pw)[len(pw) // 2] if pw else None, len(out)), flush=True)
kn.gemm_switches(tile=0, bf16x=6)
