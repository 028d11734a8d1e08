"""CPU restatement of the frozen text encoders of the MESM forward (TEST INFRASTRUCTURE ONLY: imported by
tests/ and by nothing in mesm_amd/).

  clip_text_forward   CLIPTextEncoder.forward, model/text_encoder.py:340-354, blocks :168-189 (pre-LN,
                      x += attn(ln_1 x); x += c_proj(QuickGELU(c_fc(ln_2 x)))), QuickGELU :163-165,
                      fp16-safe LayerNorm :154-160 (fp32 inside, cast back), causal additive mask :325-331,
                      nn.MultiheadAttention(need_weights=False) in fp16
  clip_encode_text    MESM.CLIP_encode_text, model/model.py:103-134
  glove_encode_text   MESM.GloVe_encode_text, model/model.py:136-143

Pinned by tests/golden/clip_text_tiny.npz (outputs of the real reference, tools/gen_golden_r2.py) in
tests/test_text_encoder_cpu.py.  `sd` holds the reference's parameter names with the dtypes
convert_weights leaves: fp16 Linear / attention / projection tensors, fp32 embeddings and LayerNorms.
"""
import torch
import torch.nn.functional as F


def _ln(x, w, b):
    return F.layer_norm(x.float(), (x.shape[-1],), w, b, 1e-5).to(x.dtype)


def clip_text_forward(sd, text, heads=None):
    """text (N, L) int64 -> last_hidden_state (N, L, W) fp16."""
    W = sd["ln_final.weight"].shape[0]
    heads = heads or W // 64
    layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")})
    L = sd["positional_embedding"].shape[0]
    x = sd["token_embedding.weight"][text].half() + sd["positional_embedding"].half()
    mask = torch.full((L, L), float("-inf")).triu_(1).half()
    x = x.permute(1, 0, 2)  # (L, N, W) like the reference
    for i in range(layers):
        p = "transformer.resblocks.%d." % i
        h = _ln(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
        a, _ = F.multi_head_attention_forward(
            h, h, h, W, heads, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"], None, None, False, 0.0,
            sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], training=False, need_weights=False,
            attn_mask=mask)
        x = x + a
        h = _ln(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
        f = F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])
        f = f * torch.sigmoid(1.702 * f)
        x = x + F.linear(f, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
    x = x.permute(1, 0, 2)
    return _ln(x, sd["ln_final.weight"], sd["ln_final.bias"])


def _pool(words_feat, words_mask, normalize):
    words_feat = words_feat.masked_fill(~words_mask.unsqueeze(-1), 0)
    sent = words_feat.sum(1) / words_mask.sum(1).unsqueeze(-1)
    if normalize:
        words_feat = F.normalize(words_feat, dim=-1, p=2, eps=1e-5)
        sent = F.normalize(sent, dim=-1, p=2, eps=1e-5)
    return words_feat, sent


def clip_encode_text(sd, words_id, words_mask, max_words_l, normalize=True):
    """-> words_feat (N, Lw, W) f32, sentence_feat (N, W), words_id[:, :Lw], words_mask[:, :Lw]."""
    hid = clip_text_forward(sd, words_id).float()[:, :max_words_l]
    wid, wm = words_id[:, :max_words_l], words_mask[:, :max_words_l]
    wf, sent = _pool(hid, wm, normalize)
    return wf, sent, wid, wm


def glove_encode_text(emb_weight, words_id, words_mask, normalize=True):
    return _pool(emb_weight[words_id], words_mask, normalize)
