"""In-situ cost of a kernel family: time the captured training step with that family's launches
turned into no-ops (results are garbage; only the step time matters).  rocprofv3 inflates short
kernels by ~2 us each, so this is the attribution that adds up to the real step time.
Usage: python tools/ablate.py [families...]   families: gemm attn ln loss elt glue  (default: each in turn)
Caveat: with the attention core ablated its outputs are uninitialised memory, and the data-dependent
assignment loop of the criterion runs longer or shorter on garbage -- price attention from the kernel trace
(tools/trace_summary.py) or tools/attn_bench.py instead."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mesm_amd import _lib, build_criterion, build_model, synthetic
from mesm_amd.graphed import GraphedStep

FAMILIES = {
    "gemm": ["mesm_gemm_f32", "mesm_gemm_group"],
    "attn": ["mesm_attn_fwd", "mesm_attn_bwd", "mesm_attn_fwd_group", "mesm_attn_bwd_group"],
    "ln": ["mesm_layernorm_fwd", "mesm_layernorm_bwd", "mesm_layernorm_bwd2", "mesm_layernorm_fwd2",
           "mesm_layernorm_bwd3", "mesm_layernorm_fwd_group", "mesm_layernorm_bwd_group"],
    "glue": ["mesm_stack_rows", "mesm_unstack_rows", "mesm_prepend_fwd", "mesm_prepend_bwd", "mesm_split_token_fwd",
             "mesm_split_token_bwd", "mesm_token_mix_fwd", "mesm_token_mix_bwd", "mesm_gather_rows_fwd",
             "mesm_gather_rows_bwd", "mesm_add_wrap", "mesm_glue_group", "mesm_add_n"],
    "loss": ["mesm_criterion_fwd", "mesm_criterion_bwd", "mesm_set_loss_fwd", "mesm_set_loss_fwd_layers", "mesm_set_loss_bwd", "mesm_rec_ss_fwd", "mesm_rec_ss_bwd", "mesm_rec_fw_reduce",
             "mesm_rec_fw_rowgrad", "mesm_nll_smooth_fwd", "mesm_nll_smooth_bwd", "mesm_saliency_loss_fwd",
             "mesm_saliency_loss_bwd", "mesm_weighted_sum", "mesm_scale_vec", "mesm_rowdot_fwd", "mesm_rowdot_bwd"],
    "elt": ["mesm_dropout", "mesm_act_bias_bwd", "mesm_sine_pos_fwd", "mesm_query_sine_fwd", "mesm_query_sine_bwd",
            "mesm_text_prep", "mesm_ref_update_fwd", "mesm_ref_update_bwd", "mesm_qsine_scale_fwd",
            "mesm_qsine_scale_bwd", "mesm_act_dropout"],
}
L = _lib.lib()
REAL = {n: getattr(L, n) for fam in FAMILIES.values() for n in fam}


def step_ms(skip, workload="C3a", reps=20):
    for n, f in REAL.items():
        setattr(L, n, f)
    for fam in skip:
        for n in FAMILIES[fam]:
            setattr(L, n, lambda *a, **k: 0)
    try:
        dev = torch.device("cuda", torch.cuda.current_device())
        args = synthetic.make_args(workload, device=str(dev))
        torch.manual_seed(1234)
        model = build_model(args); crit = build_criterion(args); model.train()
        batch = synthetic.to_device(synthetic.workload_batch(workload, seed=0), dev)
        g = GraphedStep(model, crit, batch, args.dataset_name, warmup=1)
        for _ in range(3):
            g.run(redraw=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.run(redraw=False)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    finally:
        for n, f in REAL.items():
            setattr(L, n, f)


def family_costs(workload="C3a", families=None, reps=20):
    """{family: ms it costs inside the captured step, ..., 'full_step': ms, 'floor': ms with all of them off}"""
    base = step_ms([], workload, reps)
    res = {"full_step": base}
    for f in families or list(FAMILIES):
        res[f] = base - step_ms([f], workload, reps)
    res["floor"] = step_ms(list(FAMILIES), workload, reps)
    return res


if __name__ == "__main__":
    fams = sys.argv[1:] or list(FAMILIES)
    r = family_costs(families=fams)
    print("full step            %.3f ms" % r["full_step"])
    for f in fams:
        print("without %-6s       %.3f ms   -> %-6s costs %.3f ms" % (f, r["full_step"] - r[f], f, r[f]))
    print("without all of them  %.3f ms   (= ATen glue + launch floor)" % r["floor"])
