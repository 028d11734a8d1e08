"""bench.py — clip-query pairs/s (fwd + criterion + bwd) of the MESM hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = model(**batch) -> criterion(outputs, batch) -> zero_grad -> total.backward()
[-> gradient all-reduce for N > 1], in TRAIN mode (all dropouts active), on the QVHighlights
C+SF workload "C3a" of SURVEY.md §8d (32 pairs per GPU, Lv=75, Lw=32, Dv=2818, Dt=512,
C=5003, fp32).  Inputs are resident in HBM before the timed region; the host-side draws of
the reference (negative query index, MLM word choice) are re-drawn every step.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  roofline     — the dominant kernel family (mesm_gemm_f32: exact-f32 MFMA GEMMs, 61 % of the
                 step by in-situ ablation, tools/ablate.py): algorithmic FLOPs per launch / mean
                 launch duration, HIP events on the launch stream around back-to-back replays of
                 the GEMM launches of one captured step, right after the timed region; `traffic`
                 = HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/);
  cpu_baseline — the CPU oracle (a port of the reference step, oracle/mesm_oracle.py) timed on
                 this box's host cores on the same workload (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C3a")
    ap.add_argument("--cpu-steps", type=int, default=3, help="oracle steps for cpu_baseline (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python (no HIP graph)")
    return ap.parse_args()


def log(msg):
    print("[bench %8.2fs] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.perf_counter()


def host_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    # respect a cgroup CPU quota if there is one
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return n


def main():
    opt = parse()
    from mesm_amd import build_criterion, build_model, synthetic
    from mesm_amd import kernels as kn
    from mesm_amd.ddp import GradReducer, init_process_group_from_env

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if opt.gpus != world:
        if world == 1 and opt.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    import torch.distributed as dist
    if world > 1:
        init_process_group_from_env(dev)

    wl = synthetic.WORKLOADS[opt.workload]
    args = synthetic.make_args(opt.workload, device=str(dev))
    torch.manual_seed(1234)  # identical weights on every rank
    model = build_model(args)
    crit = build_criterion(args)
    model.train()
    from mesm_amd.graphed import GraphedStep

    batch_cpu = synthetic.workload_batch(opt.workload, seed=rank)
    batch = synthetic.to_device(batch_cpu, dev)
    n_pairs = batch_cpu["video_feat"].shape[0]
    torch.manual_seed(99 + rank)

    def eager_step():
        out = model(**batch, dataset_name=args.dataset_name, is_training=True)
        losses, total = crit(out, batch, True)
        model.zero_grad(set_to_none=True)
        total.backward()  # the reducer's finish() runs as an engine callback for world > 1
        return total

    if opt.eager:
        reducer = GradReducer(model.gradbuf()) if world > 1 else None
        step = eager_step
    else:
        # one HIP graph per step: forward + criterion + backward; fresh host draws + dropout masks
        # every replay; for N > 1 the flat gradient buffer is all-reduced right after the replay
        gstep = GraphedStep(model, crit, batch, args.dataset_name)
        reducer = GradReducer(model.gradbuf(), hook=False) if world > 1 else None
        log("step captured in a HIP graph")

        def step():
            total = gstep.run(redraw=True)
            if reducer is not None:
                reducer.finish()
            return total

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log("model built (%d params), starting warm-up" % sum(p.numel() for p in model.parameters()))
    for i in range(opt.warmup):
        step()
        if i == 0:
            torch.cuda.synchronize()
            log("first step done")
    fence()
    log("warm-up done")
    t0 = time.perf_counter()
    for _ in range(opt.steps):
        last = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert torch.isfinite(last), "non-finite loss in the timed region"
    log("timed region: %.3f ms/step" % (dt / opt.steps * 1e3))

    roofline = None
    if not opt.no_roofline:
        # the GEMM launches of one captured step, replayed back to back from C++ with a HIP-event
        # pair around every launch (same arguments and buffers as the graph; see mesm_gemm_tape)
        model.gradbuf().on_ready = None
        istep = GraphedStep(model, crit, batch, args.dataset_name, warmup=1, instrument=True)
        istep.run()
        torch.cuda.synchronize()
        kn.gemm_tape_replay(1)  # warm
        prof = kn.gemm_tape_replay(opt.steps)
        if prof["launches"] > 0:
            avg_ms = prof["ms"] / prof["launches"]
            flops_per_launch = prof["flops"] / prof["launches"]
            achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "gemm_traffic.json")
            if os.path.exists(tpath):  # PMC FETCH_SIZE / WRITE_SIZE passes of this same command
                with open(tpath) as f:
                    per_step = json.load(f).get("hbm_bytes_per_step")
                if per_step:  # same "launch" as achieved: one mesm_gemm_f32 / mesm_gemm_group call
                    traffic = per_step / (prof["launches"] / opt.steps)
            roofline = {"kernel": "mesm_gemm_f32 (gemm_wstage / gemm_lds64 / gemm_frag / gemm_f32 kernels, "
                                  "v_mfma_f32_32x32x2_f32)", "bound": "mfma",
                        "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                        "launches_per_step": prof["launches"] / opt.steps,
                        "measured": "HIP event pair around the back-to-back GEMM launches of one captured "
                                    "step, replayed %d x" % opt.steps,
                        "avg_launch_us": avg_ms * 1e3, "flops_per_launch": flops_per_launch,
                        "gemm_ms_per_step": prof["ms"] / opt.steps,
                        "algorithmic_bytes_per_launch": prof.get("bytes", 0) / max(prof["launches"], 1)}

    # informational (NOT part of the metric, which is fwd + bwd): the optimizer tail of train.py:70-72 on
    # the flat buffers, global-norm clip + AdamW in two launches
    opt_tail_ms = None
    if rank == 0:
        from mesm_amd.optim import FlatAdamW
        fo = FlatAdamW(model, lr=1e-4, weight_decay=1e-4)
        for _ in range(3):
            fo.step(grad_clip=0.1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            fo.step(grad_clip=0.1)
        torch.cuda.synchronize()
        opt_tail_ms = (time.perf_counter() - t1) / 20 * 1e3

    # informational: the same steps with every batch arriving from HOST memory (PCIe-inclusive rate;
    # never `value`): host batch copied into the graph static inputs between replays (mesm_amd/feeder.py)
    pcie = None
    if rank == 0 and not opt.eager:
        from mesm_amd.feeder import BatchFeeder
        feeder = BatchFeeder(gstep, keys=("video_feat", "words_id", "video_mask", "saliency_label", "clip_mask",
                                          "unknown_mask", "words_label"))
        for _ in range(3):
            feeder.feed(batch_cpu); gstep.run(redraw=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(opt.steps):
            feeder.feed(batch_cpu); gstep.run(redraw=True)
        torch.cuda.synchronize()
        pdt = (time.perf_counter() - t2) / opt.steps
        pcie = {"pairs_per_s": n_pairs / pdt, "ms_per_step": pdt * 1e3, "host_bytes_per_step": feeder.bytes}

    cpu_baseline = None
    if rank == 0 and world == 1 and opt.cpu_steps > 0:
        from oracle import mesm_oracle as O
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        cfg = dict(vars(args))
        neg, masked = synthetic.host_draws(batch_cpu, seed=0)
        ncores = min(host_cores(), 64)
        torch.set_num_threads(ncores)
        log("cpu baseline on %d threads" % ncores)
        O.train_step(sd, cfg, batch_cpu, neg, masked)  # warm-up
        log("cpu warm-up step done")
        c0 = time.perf_counter()
        for _ in range(opt.cpu_steps):
            O.train_step(sd, cfg, batch_cpu, neg, masked)
        cdt = time.perf_counter() - c0
        cpu_baseline = {"value": n_pairs * opt.cpu_steps / cdt, "unit": "pairs/s", "cores": ncores,
                        "kind": "port",
                        "sample": "%d fwd+bwd steps of %s (%d pairs each), dropout off, torch-CPU fp32"
                                  % (opt.cpu_steps, opt.workload, n_pairs)}

    if rank == 0:
        line = {
            "metric": "clip-query pairs/sec (fwd+bwd)", "value": n_pairs * world * opt.steps / dt,
            "unit": "pairs/s", "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup,
            "ms_per_step": dt / opt.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: QVHighlights C+SF, %d pairs/GPU (%d groups), Lv=%d, Lw=%d, "
                                   "Dv=%d, Dt=%d, C=%d, 10 moment queries, train mode (dropout on)"
                                   % (opt.workload, n_pairs, len(wl["groups"]), wl["Lv"], wl["Lw"],
                                      wl["v_feat_dim"], wl["t_feat_dim"], wl["vocab_size"] + 1),
                       "global_pairs": n_pairs * world, "parallelism": "dp%d" % world,
                       "launch": "eager" if opt.eager else "hip-graph",
                       "optimizer_tail_ms_not_in_metric": opt_tail_ms,
                       "pcie_inclusive_not_in_metric": pcie},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
