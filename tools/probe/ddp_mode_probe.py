"""Probe on ONE GPU (RCCL world size 1): what does each data-parallel mode cost the captured step, without any
wire time?  (a) no reducer; (b) bucket all-reduces captured inside the step graph on the collective stream;
(c) one all-reduce of the flat gradient buffer after the replay."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist


def bench(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    from mesm_amd import build_criterion, build_model, synthetic
    from mesm_amd.ddp import GradReducer
    from mesm_amd.graphed import GraphedStep
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from mesm_amd.ddp import init_process_group_from_env
    os.environ.update(RANK="0", WORLD_SIZE="1")
    init_process_group_from_env(dev)
    args = synthetic.make_args("C3a", device="cuda:0")
    torch.manual_seed(7)
    model = build_model(args); crit = build_criterion(args); model.train()
    batch = synthetic.to_device(synthetic.workload_batch("C3a", seed=1), dev)
    mode = sys.argv[1] if len(sys.argv) > 1 else "a"   # one mode per process: a | c | ci | b1 | b6 | i1
    if mode in ("a", "c", "ci"):
        g0 = GraphedStep(model, crit, batch, args.dataset_name)
        if mode == "a":
            print("(a) no reducer:                         %.3f ms/step" % bench(lambda: g0.run()), flush=True)
        else:
            # ci: ONE blocking all-reduce on the compute stream itself (no second queue); c: six asynchronous
            # ones on the process group's stream
            red = GradReducer(model.gradbuf(), hook=False, force=True, inline=mode == "ci",
                              n_buckets=1 if mode == "ci" else 6)

            def after():
                g0.run()
                red.finish()
            print("(%s) all-reduce after the replay (%s): %.3f ms/step"
                  % (mode, "blocking, compute stream" if mode == "ci" else "asynchronous, collective stream", bench(after)),
                  flush=True)
    else:
        nb = int(mode[1:])   # b<n>: overlapped buckets on the collective stream; i<n>: blocking, on the capture stream
        red2 = GradReducer(model.gradbuf(), n_buckets=nb, hook=True, force=True, inline=mode[0] == "i")
        g1 = GraphedStep(model, crit, batch, args.dataset_name, warmup=2, reducer=red2)
        print("(b) %d bucket(s) captured in the graph:   %.3f ms/step" % (nb, bench(lambda: g1.run())), flush=True)


if __name__ == "__main__":
    main()
    sys.stdout.flush()
    os._exit(0)
