"""The frozen text encoders on the HIP path (mesm_amd/text_encoder.py, csrc/clip_text.hip) -- SURVEY.md 8a A12.

  * CLIPTextEncoder.forward + MESM.CLIP_encode_text against the outputs of the REAL reference
    (tests/golden/clip_text_tiny.npz), weights loaded through the reference's state-dict names;
  * the same at the released checkpoint's size (width 512, 8 heads, 12 layers, context 77, 32 pairs)
    against the CPU oracle (random weights: the released ones are not in the repository);
  * a whole training step of a model built by build_model with tokenizer_type="CLIP" (the shipped
    config/QVHighlights/C+SF_C.json form) against the reference's step (tests/golden/qvh_clip_tiny.npz);
  * GloveTextEncoder + GloVe_encode_text against the oracle;
  * build_model with the field values of C+SF_C.json returns a model (the round-1 boundary hole).

Tolerance of the fp16 tower.  Activations are IEEE fp16: one ulp at the top binade of a tensor is up to
9.8e-4 of its largest magnitude, and the reference's own fp16 attention (SDPA) differs from an fp32
evaluation of the same formula by one ulp on a third of the elements (measured).  So: relative L2 error
< 1e-3, and no element further than 3 fp16 ulps of the tensor maximum (3e-3) from the reference.
"""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from golden_io import GOLDEN, Fixture

pytestmark = pytest.mark.gpu
L2_TOL, MAX_TOL = 1e-3, 3e-3


def dev():
    return torch.device("cuda:0")


def errs(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-9)), float((a - b).abs().max() / b.abs().max().clamp_min(1e-9))


def check16(a, b, what, l2_tol=L2_TOL):
    l2, mx = errs(a, b)
    assert l2 < l2_tol and mx < MAX_TOL, (what, l2, mx)


def load_tiny():
    z = np.load(os.path.join(GOLDEN, "clip_text_tiny.npz"))
    sd = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("sd.")}
    return sd, z, (lambda k: torch.from_numpy(z[k].copy()))


def test_clip_text_encoder_matches_reference_golden():
    from mesm_amd.text_encoder import clip_text_encoder_from_state_dict
    sd, z, t = load_tiny()
    enc = clip_text_encoder_from_state_dict(sd).to(dev())
    assert enc.transformer.resblocks[0].attn.in_proj_weight.dtype == torch.float16
    assert enc.token_embedding.weight.dtype == torch.float32
    out = enc(t("ids").to(dev()))
    hid = out["last_hidden_state"]
    assert hid.dtype == torch.float16
    check16(hid, t("hidden"), "last_hidden_state")
    # the end-of-text features through text_projection (text_encoder.py:352-354), pinned by the real reference too
    zp = np.load(os.path.join(GOLDEN, "clip_pooler_tiny.npz"))
    assert out["pooler_output"].dtype == torch.float16
    # one more fp16 product on top of the tower's output (which itself sits at ~1e-3 of the reference): the bound of
    # the 12-layer case
    check16(out["pooler_output"], torch.from_numpy(zp["pooler_output"].copy()), "pooler_output", l2_tol=2e-3)


def test_clip_encode_text_wrapper_matches_reference_golden():
    fx = Fixture("qvh_clip_tiny")
    sd, z, t = load_tiny()
    args = argparse.Namespace(**fx.cfg)
    args.device = "cuda:0"
    model = _build_clip_model(args, fx.sd)
    wf, sf, wid, wm = model.CLIP_encode_text(t("ids").to(dev()), t("mask").to(dev()))
    assert torch.equal(wid.cpu(), t("words_id_cut")) and torch.equal(wm.cpu(), t("words_mask_cut"))
    check16(wf, t("words_feat"), "words_feat")
    check16(sf, t("sentence_feat"), "sentence_feat")
    assert float(wf[~wm].abs().max()) == 0.0


def _build_clip_model(args, sd):
    """build_model reads the text tower's sizes from the checkpoint file (runner.py:166-187): write the
    fixture's `text_encoder.*` tensors to a temporary .pth like the released CLIP weights."""
    import tempfile
    from mesm_amd import build_model
    te = {k[len("text_encoder."):]: v for k, v in sd.items() if k.startswith("text_encoder.")}
    path = os.path.join(tempfile.mkdtemp(), "clip.pth")
    torch.save(te, path)
    args.text_model_path = path
    model = build_model(args)
    model.load_state_dict(sd)
    return model


def test_full_size_clip_tower_against_cpu_oracle():
    from mesm_amd.text_encoder import CLIPTextEncoder
    from oracle import clip_text_oracle as C
    torch.manual_seed(3)
    enc = CLIPTextEncoder(512, 77, 49408, 512, 8, 12)
    with torch.no_grad():
        for n, p in enc.named_parameters():
            if "ln_" in n:
                p.add_(torch.randn_like(p) * 0.1)
            elif n.endswith("bias"):
                p.copy_(torch.randn(p.shape) * 0.05)
            elif "embedding" in n:
                p.mul_(10.0)
    sd = {k: v.clone() for k, v in enc.state_dict().items()}
    N = 32
    tl = torch.tensor([5 + (7 * i) % 70 for i in range(N)])
    mask = torch.arange(77)[None, :] < tl[:, None]
    ids = torch.randint(1, 49406, (N, 77)) * mask
    ids[torch.arange(N), tl - 1] = 49407
    want = C.clip_text_forward(sd, ids)
    got = enc.to(dev())(ids.to(dev()))["last_hidden_state"]
    # rounding noise of the residual stream grows like sqrt(depth): the 3-block fixture meets 1e-3, 12 blocks 2e-3
    check16(got, want, "last_hidden_state 12 layers", l2_tol=2e-3)
    wf_o, sf_o, _, _ = C.clip_encode_text(sd, ids, mask, 32)
    from mesm_amd import kernels as kn
    wf, sf = kn.text_pool(got, mask.to(dev()), 32, True)
    check16(wf, wf_o, "words_feat", l2_tol=2e-3)
    check16(sf, sf_o, "sentence_feat", l2_tol=2e-3)


def test_training_step_with_clip_tokenizer_matches_reference_golden():
    """The reference built by runner.build_model(tokenizer_type='CLIP') vs this build, same weights and token
    ids: the fp16 text features differ by fp16 rounding (above), everything behind them is fp32; outputs
    and losses within 2e-3, matched indices identical, MLM head has vocab_size + 3 classes."""
    from mesm_amd import build_criterion, synthetic
    fx = Fixture("qvh_clip_tiny")
    args = argparse.Namespace(**fx.cfg)
    args.device = "cuda:0"
    model = _build_clip_model(args, fx.sd)
    crit = build_criterion(args)
    model.eval()
    batch = synthetic.to_device(fx.batch, dev())
    out = model(**batch, dataset_name="qvhighlights", is_training=True, neg_index=fx.neg_index,
                masked_words=fx.masked_words)
    losses, total = crit(out, batch, True)
    model.zero_grad()
    total.backward()
    assert out["recfw_words_logit"].shape[-1] == fx.cfg["vocab_size"] + 3
    for k in ("pred_logits", "pred_spans", "saliency_scores", "neg_saliency_scores", "recfw_words_logit",
              "recon_feat", "projed_words_feat", "expanded_words_feat", "enhanced_video_feat"):
        l2, mx = errs(out[k], fx.out[k])
        assert l2 < 2e-3 and mx < 5e-3, (k, l2, mx)
    assert torch.equal(out["words_mask"].cpu(), fx.out["words_mask"].bool())
    for k, v in fx.losses.items():
        got = float(total) if k == "total" else float(losses[k])
        assert abs(got - v) < 2e-3 * max(1.0, abs(v)), (k, got, v)
    mq = crit.last_match[0].cpu().tolist()
    sizes = fx.match["main.sizes"].tolist()
    got, k = set(), 0
    for b, s in enumerate(sizes):
        for t_ in range(s):
            got.add((b, mq[k], t_))
            k += 1
    assert got == fx.matched_pairs("main")
    grads = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}
    assert set(grads) == set(fx.grads)
    assert not any(n.startswith("text_encoder") for n in grads)  # frozen
    # the text features carry fp16 rounding noise (~1e-3): gradients follow it; relative L2 with the floor the
    # full-width tests use for gradients that are analytically ~0 (softmax-invariant key projections)
    def l2(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return float((a - b).norm()) / max(float(b.norm()), 1e-3 * b.numel() ** 0.5)
    # (d = 32: a PReLU / ReLU unit whose pre-activation sits within the 1e-3 text noise of zero flips and moves
    # one rank-1 term of the weight gradients upstream, hence 5e-2 and not the 1e-2 of the fp32-only fixtures)
    worst = max((l2(grads[n], g), n) for n, g in fx.grads.items())
    assert worst[0] < 5e-2, worst


def test_glove_text_encoder_against_oracle():
    from mesm_amd import kernels as kn
    from mesm_amd.text_encoder import GloveTextEncoder
    from oracle import clip_text_oracle as C
    torch.manual_seed(1)
    enc = GloveTextEncoder(1113, dim=300)
    enc.emb.weight.data.normal_()
    enc.emb.weight.data[0].zero_()  # <PAD>
    N, Lw = 6, 16
    wl = torch.tensor([3, 16, 7, 1, 9, 12])
    mask = torch.arange(Lw)[None, :] < wl[:, None]
    ids = torch.randint(1, 1113, (N, Lw)) * mask
    wf_o, sf_o = C.glove_encode_text(enc.emb.weight.data, ids, mask)
    enc = enc.to(dev())
    emb = enc(ids.to(dev()))
    assert torch.equal(emb.cpu(), enc.emb.weight.data.cpu()[ids])
    wf, sf = kn.text_pool(emb, mask.to(dev()), Lw, True)
    assert errs(wf, wf_o)[1] < 1e-5 and errs(sf, sf_o)[1] < 1e-5


def test_build_model_from_the_shipped_clip_config_values():
    """config/QVHighlights/C+SF_C.json (tokenizer_type 'CLIP', load_vocab_pkl false): build_model must return
    a model (round 1 raised NotImplementedError).  The CLIP weights themselves are not in the repository
    (.gitignore: pretrained_models), so a random tower of the released size stands in for the file."""
    import tempfile
    from mesm_amd import build_model
    from mesm_amd.text_encoder import CLIPTextEncoder
    shipped = dict(  # the model / text fields of config/QVHighlights/C+SF_C.json
        tokenizer_type="CLIP", load_vocab_pkl=False, max_words_l=32, max_video_l=75, v_feat_dim=2818,
        t_feat_dim=512, vocab_size=5000, hidden_dim=256, nheads=8, dim_feedforward=1024, num_recfw_layers=2,
        t2v_layers=2, enc_layers=2, dec_layers=2, num_recss_layers=4, num_queries=10, dropout=0.1,
        input_dropout=0.5, n_input_proj=2, rec_fw=True, rec_ss=True, aux_loss=True, span_loss_type="l1",
        use_txt_pos=False, pre_norm=False, position_embedding="sine", normalize_txt=True, share_MLP=True)
    enc = CLIPTextEncoder(512, 77, 49408, 512, 8, 12)
    path = os.path.join(tempfile.mkdtemp(), "clip_text_encoder.pth")
    torch.save(enc.state_dict(), path)
    args = argparse.Namespace(device="cuda:0", text_model_path=path, **shipped)
    model = build_model(args)
    assert isinstance(model.text_encoder, CLIPTextEncoder)
    assert model.output_txt_proj[1].weight.shape == (5003, 256)
    assert not any(p.requires_grad for p in model.text_encoder.parameters())
    keys = set(model.state_dict())
    assert "text_encoder.transformer.resblocks.11.mlp.c_proj.weight" in keys
    from mesm_amd import synthetic
    feat = synthetic.workload_batch("C3a", seed=0)
    batch = synthetic.to_device(synthetic.with_clip_tokens(feat, vocab=49408), dev())
    model.train()
    out = model(**batch, dataset_name="qvhighlights", is_training=True)
    assert out["recfw_words_logit"].shape == (32, 32, 5003) and bool(torch.isfinite(out["pred_spans"]).all())
