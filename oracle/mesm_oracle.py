"""CPU oracle for the MESM hot path — TEST INFRASTRUCTURE, NOT A PRODUCT PATH.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker (or as the timed CPU baseline, kind="port").
``mesm_amd`` never imports it and has no CPU fallback.

What it is: a from-scratch, functional (state-dict in, tensors out) restatement in plain
PyTorch-CPU ops of the reference's training step

    MESM.forward            /root/reference/model/model.py:154-359
    Criterion.forward       /root/reference/model/criterion.py:319-367
    HungarianMatcher        /root/reference/model/matcher.py:39-117

with the reference's parameter names so the same state dict drives
the reference, this oracle and the HIP build.  Dropout is not modelled (parity is defined
with dropout off); the two host-RNG draws of the reference (negative-query index,
model.py:260, and MLM word choice, model.py:361-384) are *inputs* here so that they can
be replayed.

Parity pinning: `tools/gen_golden.py` imports the real reference in the build container,
runs it on seeded inputs and stores inputs + outputs under tests/golden/;
tests/test_oracle_golden.py checks this oracle against every one of those vectors and
against the reference's own span/gIoU doctest values (utils/span_utils.py:12-19, 31-38,
54-60, 105-109).  Third-party arithmetic on the path: torch (reference pins 1.11.0,
fixtures generated with 2.10.0), scipy.optimize.linear_sum_assignment (pinned 1.9.1,
here 1.15.3).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment

# ----------------------------------------------------------------------------- helpers


def _lin(x, sd, prefix):
    w = sd[prefix + ".weight"]
    # (x follows the parameters' dtype: the fp64 adjudication run of train_step64 feeds fp32 sine tables)
    return x.to(w.dtype) @ w.t() + sd[prefix + ".bias"]


def _ln(x, sd, prefix, eps=1e-5):
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def _prelu(x, slope):
    return torch.where(x > 0, x, slope * x)


def _l2norm(x, eps):
    return x / x.norm(dim=-1, keepdim=True).clamp_min(eps)


# Test switch mirrored by mesm_amd.testing.no_relu(): the kink control run of tests/test_model_gpu.py.
NO_RELU = False


# Train-mode dropout (bench.py's cpu_baseline times the step the way SURVEY.md 8d prescribes: dropout ON, like the
# GPU leg).  None = off (every parity check: a CPU mask stream cannot match the GPU's counter hash anyway);
# (p, p_input) = nn.Dropout(p) at the reference's transformer sites, nn.Dropout(p_input) inside LinearLayer.
DROPOUT = None


def _drop(x, which=0):
    if DROPOUT is None or DROPOUT[which] <= 0.0:
        return x
    return torch.nn.functional.dropout(x, DROPOUT[which], training=True)


def _relu(x):
    return x if NO_RELU else torch.relu(x)


def linear_layer_stack(x, sd, prefix, n, relu_flags):
    """nn.Sequential of LinearLayer (model.py:412-434): LN -> [dropout] -> Linear -> [ReLU]."""
    for i in range(n):
        x = _drop(_ln(x, sd, "%s.%d.LayerNorm" % (prefix, i)), 1)  # model.py:421-431
        x = _lin(x, sd, "%s.%d.net.1" % (prefix, i))
        if relu_flags[i]:
            x = _relu(x)
    return x


def mlp(x, sd, prefix, n):
    """MLP (model.py:397-409 / transformer.py:21-33): ReLU between layers."""
    for i in range(n):
        x = _lin(x, sd, "%s.layers.%d" % (prefix, i))
        if i < n - 1:
            x = _relu(x)
    return x


def sine_position(mask, d):
    """PositionEmbeddingSine.forward (position_encoding.py:51-72), normalize=True."""
    x = mask.cumsum(1, dtype=torch.float32)
    x = x / (x[:, -1:] + 1e-6) * (2 * math.pi)
    i = torch.arange(d, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(i, 2, rounding_mode="trunc") / d)
    ang = x[:, :, None] / dim_t
    out = torch.empty_like(ang)
    out[..., 0::2] = ang[..., 0::2].sin()
    out[..., 1::2] = ang[..., 1::2].cos()
    return out


def query_sine(ref, d):
    """gen_sineembed_for_position (transformer.py:43-59); ref (..., 2) -> (..., d)."""
    half = d // 2
    i = torch.arange(half, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(i, 2, rounding_mode="trunc") / half)
    parts = []
    for c in range(2):
        ang = (ref[..., c] * (2 * math.pi))[..., None] / dim_t
        o = torch.empty_like(ang)
        o[..., 0::2] = ang[..., 0::2].sin()
        o[..., 1::2] = ang[..., 1::2].cos()
        parts.append(o)
    return torch.cat(parts, -1)


def inverse_sigmoid(x, eps=1e-3):
    """transformer.py:36-40 / data_utils.py:139-143."""
    x = x.clamp(0, 1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def span_cxw_to_xx(s):
    return torch.stack([s[..., 0] - 0.5 * s[..., 1], s[..., 0] + 0.5 * s[..., 1]], -1)


def span_xx_to_cxw(s):
    return torch.stack([s.sum(-1) * 0.5, s[..., 1] - s[..., 0]], -1)


def temporal_iou(a, b):
    """span_utils.py:45-72."""
    inter = (torch.min(a[:, None, 1], b[:, 1]) - torch.max(a[:, None, 0], b[:, 0])).clamp(min=0)
    union = (a[:, 1] - a[:, 0])[:, None] + (b[:, 1] - b[:, 0]) - inter
    return inter / union, union


def generalized_temporal_iou(a, b):
    """span_utils.py:92-121."""
    a = a.float()
    b = b.float()
    assert (a[:, 1] >= a[:, 0]).all() and (b[:, 1] >= b[:, 0]).all()
    iou, union = temporal_iou(a, b)
    enc = (torch.max(a[:, None, 1], b[:, 1]) - torch.min(a[:, None, 0], b[:, 0])).clamp(min=0)
    return iou - (enc - union) / enc


# ----------------------------------------------------------------------------- attention


def _heads(x, h):
    n, l, e = x.shape
    return x.view(n, l, h, e // h).transpose(1, 2)  # (N, h, L, dh)


def attention_core(q, k, v, h, masked, scale):
    """softmax(scale * q k^T with -inf where `masked` (N,h,Lq,Lk) or broadcastable) v; batch-first."""
    s = torch.einsum("nhqd,nhkd->nhqk", _heads(q, h) * scale, _heads(k, h))
    if masked is not None:
        s = s.masked_fill(masked, float("-inf"))
    p = _drop(torch.softmax(s, -1))  # attention-probability dropout (nn.MultiheadAttention / attention.py:382)
    o = torch.einsum("nhqk,nhkd->nhqd", p, _heads(v, h))
    return o.transpose(1, 2).reshape(q.shape[0], q.shape[1], -1)


def t2v_mask(qpad, kpad, h):
    """Effective mask of transformer.py:528-533: key_padding_mask OR the (N*h, Lq, Lk) attn_mask
    built with .repeat(nhead,1,1) but consumed as index b*h+head."""
    n = kpad.shape[0]
    outer = qpad[:, :, None] & kpad[:, None, :]  # (N, Lq, Lk), indexed by b'
    bprime = (torch.arange(n)[:, None] * h + torch.arange(h)[None, :]) % n  # (N, h)
    return kpad[:, None, None, :] | outer[bprime]


def mha_packed(q_in, k_in, v_in, sd, prefix, h, masked):
    """nn.MultiheadAttention with packed in_proj (rows q,k,v) and out_proj; batch-first tensors."""
    e = q_in.shape[-1]
    w, b = sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"]
    q = q_in @ w[:e].t() + b[:e]
    k = k_in @ w[e:2 * e].t() + b[e:2 * e]
    v = v_in @ w[2 * e:].t() + b[2 * e:]
    o = attention_core(q, k, v, h, masked, (e // h) ** -0.5)
    return _lin(o, sd, prefix + ".out_proj")


def t2v_layer(txt, vid, pos_txt, pos_vid, txt_pad, vid_pad, sd, prefix, h, mlm=False):
    """T2V_TransformerEncoderLayer[_TwoMLP].forward_post (transformer.py:508-540, 573-612):
    query = vid(+pos), key = txt(+pos), value = txt."""
    q = vid if pos_vid is None else vid + pos_vid
    k = txt if pos_txt is None else txt + pos_txt
    masked = t2v_mask(vid_pad, txt_pad, h)
    x = vid + _drop(mha_packed(q, k, txt, sd, prefix + ".self_attn", h, masked))
    sfx = "_1" if mlm else ""
    y = _ln(x, sd, prefix + ".norm1" + sfx)
    y = _lin(_drop(_prelu(_lin(y, sd, prefix + ".linear1" + sfx), sd[prefix + ".activation.weight"])),
             sd, prefix + ".linear2" + sfx)
    return _ln(x + _drop(y), sd, prefix + ".norm2" + sfx)


def t2v_stack(txt, vid, pos_txt, pos_vid, txt_pad, vid_pad, sd, prefix, nlayers, h, mlm=False):
    for i in range(nlayers):
        vid = t2v_layer(txt, vid, pos_txt, pos_vid, txt_pad, vid_pad, sd,
                        "%s.layers.%d" % (prefix, i), h, mlm)
    return vid


def encoder_layer(x, pos, pad, sd, prefix, h):
    """TransformerEncoderLayer.forward_post (transformer.py:637-650)."""
    qk = x + pos
    x = _ln(x + _drop(mha_packed(qk, qk, x, sd, prefix + ".self_attn", h, pad[:, None, None, :])),
            sd, prefix + ".norm1")
    y = _lin(_drop(_prelu(_lin(x, sd, prefix + ".linear1"), sd[prefix + ".activation.weight"])),
             sd, prefix + ".linear2")
    return _ln(x + _drop(y), sd, prefix + ".norm2")


def decoder_layer(tgt, memory, mem_pad, pos, query_pos, qsine, sd, prefix, h, first):
    """TransformerDecoderLayer.forward (transformer.py:723-797) with the custom MHA
    (attention.py:185-394: no in-proj, q scaled by head_dim^-0.5, dk may differ from dv)."""
    n, nq, d = tgt.shape
    q = _lin(tgt, sd, prefix + ".sa_qcontent_proj") + _lin(query_pos, sd, prefix + ".sa_qpos_proj")
    k = _lin(tgt, sd, prefix + ".sa_kcontent_proj") + _lin(query_pos, sd, prefix + ".sa_kpos_proj")
    v = _lin(tgt, sd, prefix + ".sa_v_proj")
    sa = _lin(attention_core(q, k, v, h, None, (d // h) ** -0.5), sd, prefix + ".self_attn.out_proj")
    tgt = _ln(tgt + _drop(sa), sd, prefix + ".norm1")

    qc = _lin(tgt, sd, prefix + ".ca_qcontent_proj")
    kc = _lin(memory, sd, prefix + ".ca_kcontent_proj")
    v = _lin(memory, sd, prefix + ".ca_v_proj")
    kp = _lin(pos, sd, prefix + ".ca_kpos_proj")
    if first:
        qc = qc + _lin(query_pos, sd, prefix + ".ca_qpos_proj")
        kc = kc + kp
    qs = _lin(qsine, sd, prefix + ".ca_qpos_sine_proj")
    dh = d // h
    lm = memory.shape[1]
    # per head: [content || positional] -> head dim 2*dh (transformer.py:778-784)
    q2 = torch.cat([qc.view(n, nq, h, dh), qs.view(n, nq, h, dh)], -1).reshape(n, nq, 2 * d)
    k2 = torch.cat([kc.view(n, lm, h, dh), kp.view(n, lm, h, dh)], -1).reshape(n, lm, 2 * d)
    ca = attention_core(q2, k2, v, h, mem_pad[:, None, None, :], (2 * d // h) ** -0.5)
    tgt = _ln(tgt + _drop(_lin(ca, sd, prefix + ".cross_attn.out_proj")), sd, prefix + ".norm2")
    y = _lin(_drop(_prelu(_lin(tgt, sd, prefix + ".linear1"), sd[prefix + ".activation.weight"])),
             sd, prefix + ".linear2")
    return _ln(tgt + _drop(y), sd, prefix + ".norm3")


def detr_transformer(src, vid_mask, pos, sd, cfg, run_decoder=True):
    """Transformer.forward (transformer.py:174-205) + TransformerDecoder.forward (:333-420).
    src (N, L, d), vid_mask True = valid.  Returns hs (layers, N, nq, d), refs (layers, N, nq, 2),
    memory_local (N, L, d), memory_global (N, d)."""
    n, l, d = src.shape
    h = cfg["nheads"]
    g_tok = sd["global_rep_token"].view(1, 1, d).expand(n, 1, d)
    g_pos = sd["global_rep_pos"].view(1, 1, d).expand(n, 1, d)
    x = torch.cat([g_tok, src], 1)
    p = torch.cat([g_pos, pos], 1)
    pad = torch.cat([torch.ones(n, 1, dtype=torch.bool), ~vid_mask], 1)  # global token is a masked key
    for i in range(cfg["enc_layers"]):
        x = encoder_layer(x, p, pad, sd, "transformer.encoder.layers.%d" % i, h)
    mem_g, mem = x[:, 0], x[:, 1:]
    if not run_decoder:
        return None, None, mem, mem_g

    nq = cfg["num_queries"]
    pre = "transformer.decoder"
    ref = torch.sigmoid(sd["query_embed.weight"])[None].expand(n, nq, 2)
    refs = [ref]
    out = torch.zeros(n, nq, d, dtype=src.dtype)
    inter = []
    nl = cfg["dec_layers"]
    for li in range(nl):
        qsine = query_sine(ref, d)
        query_pos = mlp(qsine, sd, pre + ".ref_point_head", 2)
        if li > 0:
            qsine = qsine * mlp(out, sd, pre + ".query_scale", 2)
        cond = torch.sigmoid(mlp(out, sd, pre + ".ref_anchor_head", 2))
        qsine = qsine * (cond[..., 0] / ref[..., 1])[..., None]
        out = decoder_layer(out, mem, pad[:, 1:], pos, query_pos, qsine, sd,
                            "%s.layers.%d" % (pre, li), h, li == 0)
        new_ref = torch.sigmoid(mlp(out, sd, pre + ".bbox_embed", 3) + inverse_sigmoid(ref))
        if li != nl - 1:
            refs.append(new_ref)
        ref = new_ref.detach()
        inter.append(_ln(out, sd, pre + ".norm"))
    return torch.stack(inter), torch.stack(refs), mem, mem_g


# ----------------------------------------------------------------------------- model


def _pad_groups(chunks):
    """pad_sequences_1d on a list of (len_i, ...) tensors -> padded (n, Lmax, ...), mask."""
    lmax = max(c.shape[0] for c in chunks)
    out = chunks[0].new_zeros((len(chunks), lmax) + tuple(chunks[0].shape[1:]))
    mask = torch.zeros(len(chunks), lmax, dtype=torch.bool)
    for i, c in enumerate(chunks):
        out[i, :c.shape[0]] = c
        mask[i, :c.shape[0]] = True
    return out, mask


def _mix_token(x, where, token):
    """x with rows selected by `where` replaced by `token` (model.py:380-382, 391-393, 499-501)."""
    return torch.where(where[..., None], token.expand_as(x), x)


def mesm_forward(sd, cfg, batch, neg_index, masked_words=None, is_training=True):
    """MESM.forward (model.py:154-359).  Text: words_id holds (N,Lw,Dt) features (text_encoder=None,
    post_process_text) unless sd carries a frozen encoder: `text_encoder.*` CLIP tensors -> CLIP_encode_text,
    `text_encoder.emb.weight` -> GloVe_encode_text (oracle/clip_text_oracle.py).

    batch: video_feat (N,Lv,Dv), video_mask (N,Lv) bool, words_id (N,Lw,Dt), num_clips (G,),
           unknown_mask (N,Lw), clip_mask (N,Lv) for the MLM branch.
    neg_index (N,) int64: replay of sample_outclass_neg (model.py:260).
    masked_words (N,Lw) bool: replay of _mask_words' choice (model.py:366-378).
    """
    h = cfg["nheads"]
    d = cfg["hidden_dim"]
    nproj = cfg["n_input_proj"]
    relu_flags = [True] * 3
    relu_flags[nproj - 1] = False
    video_feat, video_mask = batch["video_feat"], batch["video_mask"]
    num_clips = batch["num_clips"]
    n = video_feat.shape[0]

    if "text_encoder.ln_final.weight" in sd:  # CLIP_encode_text, model.py:103-134
        from .clip_text_oracle import clip_encode_text
        te = {k[len("text_encoder."):]: v.detach() for k, v in sd.items() if k.startswith("text_encoder.")}
        with torch.no_grad():
            words, sent, _, words_mask = clip_encode_text(te, batch["words_id"], batch["words_mask"],
                                                          cfg["max_words_l"], cfg.get("normalize_txt", True))
    elif "text_encoder.emb.weight" in sd:  # GloVe_encode_text, model.py:136-143
        from .clip_text_oracle import glove_encode_text
        words_mask = batch["words_mask"]
        with torch.no_grad():
            words, sent = glove_encode_text(sd["text_encoder.emb.weight"].detach(), batch["words_id"], words_mask,
                                            cfg.get("normalize_txt", True))
    else:
        # post_process_text, model.py:145-152
        words = batch["words_id"]
        if cfg.get("normalize_txt", True):
            words = _l2norm(words, 1e-5)
        words_mask = words.sum(-1) != 0
        sent = words.sum(1) / words_mask.sum(1)[:, None]
        if cfg.get("normalize_txt", True):
            sent = _l2norm(sent, 1e-5)

    pv = linear_layer_stack(video_feat, sd, "input_vid_proj", nproj, relu_flags)
    pw = linear_layer_stack(words, sd, "input_txt_proj", nproj, relu_flags)
    vpos = sine_position(video_mask, d)
    use_tpos = cfg.get("use_txt_pos", False)  # False in every shipped config

    def txt_position(x):
        """TrainablePositionalEncoding.forward (position_encoding.py:19-32; dropout off): LN(x + E[0:L])"""
        if not use_tpos:
            return torch.zeros_like(x)  # model.py:171-172, 227-228
        e = sd["txt_position_embed.position_embeddings.weight"][:x.shape[1]]
        return F.layer_norm(x + e[None], (d,), sd["txt_position_embed.LayerNorm.weight"],
                            sd["txt_position_embed.LayerNorm.bias"], 1e-5)

    tpos = txt_position(pw)

    def enhance(txt, vid, txt_pad, vid_pad, ptxt, pvid, mlm=False):
        return t2v_stack(txt, vid, ptxt, pvid, txt_pad, vid_pad, sd, "enhance_encoder.t2v_encoder",
                         cfg["num_recfw_layers"], h, mlm and not cfg.get("share_MLP", True))

    enhanced = enhance(pw, pv, ~words_mask, ~video_mask, tpos, vpos) if cfg["rec_fw"] else pv

    out = {}
    if cfg["rec_ss"]:
        groups = num_clips.tolist()
        if cfg["dataset_name"] == "qvhighlights":
            # all segments of a group concatenated, repeated once per query (model.py:190-195)
            vid_chunks, start = [], 0
            for g in groups:
                seg = [video_feat[i][video_mask[i]] for i in range(start, start + g)]
                cat = torch.cat(seg, 0)
                vid_chunks += [cat] * g
                start += g
            bvid, bvid_mask = _pad_groups(vid_chunks)
        else:
            bvid, bvid_mask = video_feat, video_mask
        sent_chunks, start = [], 0
        for g in groups:
            sent_chunks += [sent[start:start + g]] * g
            start += g
        bsent, bsent_mask = _pad_groups(sent_chunks)
        bvid = linear_layer_stack(bvid, sd, "input_vid_proj", nproj, relu_flags)
        bsent = linear_layer_stack(bsent, sd, "input_txt_proj", nproj, relu_flags)
        # SegSenRecon.forward, model.py:467-503: mask the i-th sentence of its group
        slot = torch.cat([torch.arange(g) for g in groups])
        loc = torch.zeros(n, bsent.shape[1], dtype=torch.bool)
        loc[torch.arange(n), slot] = True
        q_tok = _mix_token(bsent, loc, sd["ss_reconstructor.masked_sent_token"].view(1, 1, d))
        rec = t2v_stack(bvid, q_tok, None, None, ~bvid_mask, ~bsent_mask, sd,
                        "ss_reconstructor.recon_trans", cfg["num_recss_layers"], h)
        recon = F.normalize(rec[loc])
        projed_recon = linear_layer_stack(recon, sd, "ss_reconstructor.output_sent_proj", 2,
                                          [True, False])
        ewords = torch.cat([recon[:, None], pw], 1)
        emask = torch.cat([torch.ones(n, 1, dtype=torch.bool), words_mask], 1)
    else:
        ewords, emask = pw, words_mask
    epos = txt_position(ewords)  # model.py:225-228 (of the EXPANDED words: the sentence token takes index 0)

    def align(txt, vid, txt_pad, ptxt):
        return t2v_stack(txt, vid, ptxt, vpos, txt_pad, ~video_mask, sd, "t2v_encoder.t2v_encoder",
                         cfg["t2v_layers"], h)

    encoded = align(ewords, enhanced, ~emask, epos)
    hs, refs, memory, memory_g = detr_transformer(encoded, video_mask, vpos, sd, cfg)
    logits = _lin(hs, sd, "class_embed")
    spans = torch.sigmoid(mlp(hs, sd, "span_embed", 3) + inverse_sigmoid(refs))

    # negative pass (model.py:260-299); its decoder output is discarded (:295)
    n_ewords, n_emask, n_epos = ewords[neg_index], emask[neg_index], epos[neg_index]
    if cfg["rec_ss"]:
        # (the negative words keep the positions 1.. of the expanded sequence, model.py:263-267)
        n_words, n_wmask, n_tpos = n_ewords[:, 1:], n_emask[:, 1:], n_epos[:, 1:]
    else:
        n_words, n_wmask, n_tpos = n_ewords, n_emask, n_epos
    n_enh = enhance(n_words, pv, ~n_wmask, ~video_mask, n_tpos, vpos) if cfg["rec_fw"] else pv
    n_enc = align(n_ewords, n_enh, ~n_emask, n_epos)
    _, _, n_memory, n_memory_g = detr_transformer(n_enc, video_mask, vpos, sd, cfg, run_decoder=False)

    def saliency(mem, mem_g):
        return (_lin(mem, sd, "saliency_proj1") * _lin(mem_g, sd, "saliency_proj2")[:, None]).sum(-1) \
            / np.sqrt(d)

    out.update({
        "pred_logits": logits[-1], "pred_spans": spans[-1],
        "saliency_scores": saliency(memory, memory_g),
        "neg_saliency_scores": saliency(n_memory, n_memory_g),
    })
    if cfg["aux_loss"]:
        out["aux_outputs"] = [{"pred_logits": a, "pred_spans": b} for a, b in zip(logits[:-1], spans[:-1])]

    if cfg["rec_fw"] and is_training:
        # FW-MESM masked language modelling, model.py:307-332
        unk = linear_layer_stack(sd["unknown_token"].view(1, 1, -1), sd, "input_txt_proj", nproj, relu_flags)
        msk = linear_layer_stack(sd["masked_token"].view(1, 1, -1), sd, "input_txt_proj", nproj, relu_flags)
        w = _mix_token(pw, batch["unknown_mask"], unk)
        clip_mask = batch["clip_mask"]
        lens = clip_mask.sum(1).tolist()
        sel_feat = torch.split(pv[clip_mask], lens)
        sel_pos = torch.split(vpos[clip_mask], lens)
        cfeat, cmask = _pad_groups(list(sel_feat))
        cpos, _ = _pad_groups(list(sel_pos))
        w = _mix_token(w, masked_words.bool(), msk)
        rec = enhance(cfeat, w, ~cmask, ~words_mask, cpos, tpos, mlm=True)
        hid = linear_layer_stack(rec, sd, "output_txt_proj", 1, [True])
        out["recfw_words_logit"] = _lin(hid, sd, "output_txt_proj.1")
        out["words_mask"] = words_mask
    if cfg["rec_ss"]:
        out.update({"projed_video_feat": pv, "recon_feat": recon, "projed_recon_feat": projed_recon,
                    "expanded_words_feat": ewords, "expanded_words_mask": emask,
                    "enhanced_video_feat": enhanced, "projed_words_feat": pw})
    return out


# ----------------------------------------------------------------------------- criterion


def match_cost(logits, spans, tgt_cxw, tgt_xx, cfg):
    """Cost matrix of matcher.py:70-105: (N*Q, sumT)."""
    prob = logits.flatten(0, 1).softmax(-1)
    sp = spans.flatten(0, 1)
    l1 = (sp[:, None, :] - tgt_cxw[None]).abs().sum(-1)
    giou = generalized_temporal_iou(span_cxw_to_xx(sp), tgt_xx)
    return cfg["set_cost_span"] * l1 + cfg["set_cost_giou"] * (-giou) + cfg["set_cost_class"] * (-prob[:, :1])


def hungarian(logits, spans, targets, cfg):
    """HungarianMatcher.forward.  Returns per pair (query_idx, target_idx) int64 tensors."""
    n, q = spans.shape[:2]
    multi = cfg["dataset_name"] == "qvhighlights"
    if multi:
        tgt_xx = torch.cat([t["moments"] for t in targets["norm_moment"]])
        tgt_cxw = torch.cat([t["spans"] for t in targets["norm_span"]])
        sizes = [len(t["spans"]) for t in targets["norm_span"]]
    else:
        tgt_xx, tgt_cxw = targets["norm_moment"], targets["norm_span"]
        sizes = [1] * n
    c = match_cost(logits.detach(), spans.detach(), tgt_cxw, tgt_xx, cfg).view(n, q, -1)
    res, start = [], 0
    for i, s in enumerate(sizes):
        qi, ti = linear_sum_assignment(c[i, :, start:start + s].numpy())
        res.append((torch.as_tensor(qi, dtype=torch.int64), torch.as_tensor(ti, dtype=torch.int64)))
        start += s
    return res


def _matched(spans, targets, indices, cfg):
    bi = torch.cat([torch.full_like(q, i) for i, (q, _) in enumerate(indices)])
    qi = torch.cat([q for q, _ in indices])
    src = spans[bi, qi]
    if cfg["dataset_name"] == "qvhighlights":
        tgt = torch.cat([t["spans"][j] for t, (_, j) in zip(targets["norm_span"], indices)])
        tgt_xx = span_cxw_to_xx(tgt)
    else:
        tgt, tgt_xx = targets["norm_span"], targets["norm_moment"]
    return bi, qi, src, tgt, tgt_xx


def loss_spans(spans, targets, indices, cfg):
    """criterion.py:71-110."""
    _, _, src, tgt, tgt_xx = _matched(spans, targets, indices, cfg)
    l1 = (src - tgt).abs().mean()
    giou = 1 - torch.diag(generalized_temporal_iou(span_cxw_to_xx(src), tgt_xx))
    return {"loss_span": l1, "loss_giou": giou.mean()}


def loss_labels(logits, targets, indices, cfg):
    """criterion.py:112-137: weighted CE (fg weight 1, bg weight eos_coef), plain mean."""
    n, q = logits.shape[:2]
    bi, qi, _, _, _ = _matched(logits.new_zeros(n, q, 2), targets, indices, cfg)
    cls = torch.ones(n, q, dtype=torch.int64)
    cls[bi, qi] = 0
    logp = logits.log_softmax(-1)
    w = torch.tensor([1.0, cfg["eos_coef"]], dtype=logits.dtype)
    ce = -(logp.gather(-1, cls[..., None]).squeeze(-1)) * w[cls]
    picked = logits[bi, qi]
    acc = (picked.argmax(-1) == 0).float().sum() * (100.0 / picked.shape[0])
    return {"loss_label": ce.mean(), "class_error": 100 - acc}


def loss_saliency(out, targets, cfg):
    """criterion.py:139-221."""
    vm = targets["video_mask"].float()
    sn = out["neg_saliency_scores"]
    neg_pair = (-torch.log(1.0 - torch.sigmoid(sn)) * vm).sum(1).mean()
    label = targets["saliency_label"] if "saliency_label" in targets else targets["clip_mask"].float()
    sc = torch.cat([out["saliency_scores"], sn], 1)
    lab = torch.cat([label, torch.zeros_like(label)], 1)
    vm2 = vm.repeat(1, 2)
    sc = vm2 * sc + (1.0 - vm2) * -1e3
    x = sc / 0.5
    lg = x - x.max(1, keepdim=True)[0]
    logp = lg - torch.log(torch.exp(lg).sum(1, keepdim=True) + 1e-6)
    rank = 0.0
    for r in range(1, 12):
        pos = lab >= r
        if pos.sum() == 0:
            continue
        mean_lp = (pos * logp * vm2).sum(1) / (pos.sum(1) + 1e-6)
        rank = rank + (-mean_lp * (pos.sum(1) > 0)).mean()
    total = rank / cfg["rank_coef"] + neg_pair
    if cfg["use_triplet"]:
        s = out["saliency_scores"]
        bi = torch.arange(s.shape[0])[:, None]
        ps, ns = s[bi, targets["pos_idx"]], s[bi, targets["neg_idx"]]
        total = total + torch.clamp(cfg["saliency_margin"] + ns - ps, min=0).sum() / ps.numel() * 2
    return {"loss_saliency": total}


def loss_rec_ss(out, targets, cfg):
    """criterion.py:223-274 (ablation 3: clip feature vs expanded words feature)."""
    groups = targets["num_clips"].tolist()
    if cfg["dataset_name"] == "qvhighlights":
        mom = torch.stack([torch.stack([m["moments"].min(), m["moments"].max()]) for m in targets["norm_moment"]])
    else:
        mom = targets["norm_moment"]
    n = mom.shape[0]
    pos = torch.zeros(n, n, dtype=torch.bool)
    start = 0
    for g in groups:
        blk = mom[start:start + g]
        pos[start:start + g, start:start + g] = generalized_temporal_iou(blk, blk) >= cfg["iou_gamma"]
        start += g
    cm = targets["clip_mask"][..., None]
    clip = (out["projed_video_feat"] * cm).sum(1) / cm.sum(1)
    wm = out["expanded_words_mask"][..., None]
    wf = (out["expanded_words_feat"] * wm).sum(1) / wm.sum(1)
    sim = F.normalize(clip, dim=-1) @ F.normalize(wf, dim=-1).t() / cfg["recss_tau"]
    lg = sim - sim.max(1, keepdim=True)[0]
    logp = lg - torch.log(torch.exp(lg).sum(1, keepdim=True) + 1e-6)
    return {"loss_rec_ss": (-(pos * logp).sum(1) / (pos.sum(1) + 1e-6)).mean()}


def loss_rec_fw(out, targets, cfg):
    """criterion.py:276-306, label smoothing 0.1."""
    logit, idx, mask = out["recfw_words_logit"], targets["words_label"], out["words_mask"]
    acc = ((logit.argmax(-1) == idx).float() * mask).sum() / mask.sum()
    logp = logit.log_softmax(-1)
    nll = -logp.gather(-1, idx[..., None]).squeeze(-1)
    nll = 0.9 * nll + 0.1 / logit.shape[-1] * (-logp.sum(-1))
    nll = nll.masked_fill(~mask, 0).sum(-1) / mask.sum(-1)
    return {"loss_rec_fw": nll.mean(), "rec_fw_acc": acc}


def weight_dict(cfg):
    """runner.py:313-330."""
    w = {"loss_span": cfg["loss_span_coef"], "loss_giou": cfg["loss_giou_coef"],
         "loss_label": cfg["loss_label_coef"], "loss_saliency": cfg["loss_saliency_coef"]}
    if cfg["aux_loss"]:
        base = dict(w)
        for i in range(cfg["dec_layers"] - 1):
            w.update({k + "_%d" % i: v for k, v in base.items() if k != "loss_saliency"})
    if cfg["rec_fw"]:
        w["loss_rec_fw"] = cfg["loss_recfw_coef"]
    if cfg["rec_ss"]:
        w["loss_rec_ss"] = cfg["loss_recss_coef"]
    return w


def criterion_forward(out, targets, cfg, is_training=True):
    """Criterion.forward (criterion.py:319-367).  Returns (losses, total, indices_per_layer)."""
    losses = {}
    idx = hungarian(out["pred_logits"], out["pred_spans"], targets, cfg)
    all_idx = [idx]
    losses.update(loss_spans(out["pred_spans"], targets, idx, cfg))
    losses.update(loss_labels(out["pred_logits"], targets, idx, cfg))
    losses.update(loss_saliency(out, targets, cfg))
    if cfg["rec_fw"] and is_training:
        losses.update(loss_rec_fw(out, targets, cfg))
    if cfg["rec_ss"]:
        losses.update(loss_rec_ss(out, targets, cfg))
    for i, aux in enumerate(out.get("aux_outputs", [])):
        ai = hungarian(aux["pred_logits"], aux["pred_spans"], targets, cfg)
        all_idx.append(ai)
        for k, v in {**loss_spans(aux["pred_spans"], targets, ai, cfg),
                     **loss_labels(aux["pred_logits"], targets, ai, cfg)}.items():
            losses["%s_%d" % (k, i)] = v
    wd = weight_dict(cfg)
    total = sum(losses[k] * wd[k] for k in losses if k in wd)
    return losses, total, all_idx


def train_step64(sd, cfg, batch, neg_index, masked_words):
    """train_step with every floating tensor in fp64: the referee when the fp32 oracle and the device disagree on a
    gradient -- an activation whose pre-activation is within fp32 rounding of zero has no defined fp32 side (the sign of
    z, hence the PReLU / ReLU derivative, depends on the summation order of the GEMM that produced it)."""
    def up(x):
        if torch.is_tensor(x):
            return x.double() if x.is_floating_point() else x
        if isinstance(x, dict):
            return {k: up(v) for k, v in x.items()}
        if isinstance(x, (list, tuple)):
            return type(x)(up(v) for v in x)
        return x
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        return train_step(up(sd), cfg, up(batch), neg_index, masked_words)
    finally:
        torch.set_default_dtype(old)


def train_step(sd, cfg, batch, neg_index, masked_words):
    """One fwd + criterion + backward on leaf copies of `sd`; returns (out, losses, total, grads)."""
    params = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    out = mesm_forward(params, cfg, batch, neg_index, masked_words, True)
    losses, total, idx = criterion_forward(out, batch, cfg, True)
    total.backward()
    grads = {k: p.grad for k, p in params.items() if p.requires_grad and p.grad is not None}
    return out, losses, total, grads, idx
