"""Child process of tests/test_graph_ddp_gpu.py::test_allreduce_captured_inside_the_step_graph."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist


def main():
    from mesm_amd import build_criterion, build_model, synthetic
    from mesm_amd.ddp import GradReducer
    from mesm_amd.graphed import GraphedStep
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from mesm_amd.ddp import RcclComm, init_process_group_from_env
    mode = sys.argv[1] if len(sys.argv) > 1 else "overlapped"
    own = mode.startswith("own-")  # this library's own RCCL communicator (csrc/ddp.hip): no torch process group at all
    comm = None
    if own:
        comm = RcclComm(dev)
    else:
        os.environ.update(RANK="0", WORLD_SIZE="1")
        init_process_group_from_env(dev)  # (turns the flight recorder off: see ddp.py)
    args = synthetic.make_args("C3a", device="cuda:0")
    torch.manual_seed(7)
    model = build_model(args)
    crit = build_criterion(args)
    for m in model.modules():
        if hasattr(m, "p"):
            m.p = 0.0
    batch = synthetic.to_device(synthetic.workload_batch("C3a", seed=1), dev)
    inline = mode.endswith("inline")  # collectives on the capture stream itself
    red = GradReducer(model.gradbuf(), n_buckets=6, force=True, inline=inline, comm=comm, fold_scale=own)
    g = GraphedStep(model, crit, batch, args.dataset_name, warmup=2, reducer=red)
    total_g = float(g.run(redraw=False))
    torch.cuda.synchronize()
    flat_g = model.gradbuf().flat.clone()
    log = list(red.launch_log[-6:])
    # a SECOND graph (another batch shape met later by this rank only): its warm-up must not issue collectives,
    # only the capture records its six buckets
    n_before = len(red.launch_log)
    b2 = synthetic.to_device(synthetic.workload_batch("C3b", seed=2), dev)
    g2 = GraphedStep(model, crit, b2, args.dataset_name, warmup=1, reducer=red)
    g2.run(redraw=False)
    torch.cuda.synchronize()
    second_graph_launches = len(red.launch_log) - n_before
    model.gradbuf().on_ready = None
    out = model(**batch, dataset_name=args.dataset_name, is_training=True, plan=g.plan)
    _, total = crit(out, batch, True)
    model.zero_grad(set_to_none=True)
    total.backward()
    torch.cuda.synchronize()
    flat_e = model.gradbuf().flat
    res = {"launch_log_tail": log, "second_graph_launches": second_graph_launches,
           "loss_err": abs(float(total) - total_g) / max(1.0, abs(float(total))),
           "grad_err": float((flat_e - flat_g).norm()) / max(float(flat_e.norm()), 1e-6)}
    print("RESULT " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
    sys.stdout.flush()
    os._exit(0)  # no communicator / graph teardown
