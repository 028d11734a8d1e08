"""Frozen text encoders of the MESM forward on the gfx950 kernels (reference model/text_encoder.py).

CLIPTextEncoder    :240-354  12-layer (sizes come from the checkpoint, runner.py:167-174) causal pre-LN
                   transformer in fp16: token + positional embedding, per block
                   x += out_proj(attn(ln_1(x)));  x += c_proj(QuickGELU(c_fc(ln_2(x)))), then ln_final.
                   Linear / attention weights are fp16 (convert_weights :373-394), embeddings and LayerNorm
                   parameters fp32, exactly like the reference after build_CLIP_text_encoder; forward is
                   no_grad and returns {"last_hidden_state": (N, L, W) fp16}.  `pooler_output` (the EOS token
                   through text_projection) is dead on the MESM path (model.py:120-123 uses the masked mean,
                   "ablation 2") and is not computed; the parameter exists for checkpoint compatibility.
GloveTextEncoder   :432-454  frozen nn.Embedding lookup.

Same parameter names as the reference, so `text_encoder.*` entries of a reference state dict load
unchanged (eval.py:515-518).  No CPU path: the kernels raise on a host tensor.
"""
import torch
from torch import nn

from . import kernels as kn


class _LN(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))


class _Linear16(nn.Module):
    """nn.Linear parameters in fp16 (what convert_weights leaves behind)."""

    def __init__(self, i, o):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i, dtype=torch.float16))
        self.bias = nn.Parameter(torch.zeros(o, dtype=torch.float16))


class _MHA16(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d, dtype=torch.float16))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d, dtype=torch.float16))
        self.out_proj = _Linear16(d, d)


class _MLP(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.c_fc = _Linear16(d, 4 * d)
        self.c_proj = _Linear16(4 * d, d)


class ResidualAttentionBlock(nn.Module):
    """text_encoder.py:168-189."""

    def __init__(self, d, h):
        super().__init__()
        self.attn = _MHA16(d)
        self.ln_1 = _LN(d)
        self.mlp = _MLP(d)
        self.ln_2 = _LN(d)
        self.n_head = h

    def forward(self, x):
        """x (N, L, d) fp16 -> same."""
        N, L, d = x.shape
        x2 = x.view(N * L, d)
        a = self.attn
        h = kn.layernorm_f16(x2, self.ln_1.weight, self.ln_1.bias)
        qkv = kn.gemm_f16(h, a.in_proj_weight, a.in_proj_bias, out_f32=True).view(N, L, 3 * d)
        o, _ = kn.attn_fwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], self.n_head, causal=True,
                           want_lse=False)
        x2 = kn.gemm_f16(o.view(N * L, d), a.out_proj.weight, a.out_proj.bias, residual=x2)
        h = kn.layernorm_f16(x2, self.ln_2.weight, self.ln_2.bias)
        f = kn.gemm_f16(h, self.mlp.c_fc.weight, self.mlp.c_fc.bias, quick_gelu=True)
        x2 = kn.gemm_f16(f, self.mlp.c_proj.weight, self.mlp.c_proj.bias, residual=x2)
        return x2.view(N, L, d)


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads) for _ in range(layers)])


class CLIPTextEncoder(nn.Module):
    def __init__(self, embed_dim, context_length, vocab_size, transformer_width, transformer_heads,
                 transformer_layers):
        super().__init__()
        self.context_length = context_length
        self.vocab_size = vocab_size
        self.transformer = _Transformer(transformer_width, transformer_layers, transformer_heads)
        self.token_embedding = nn.Embedding(vocab_size, transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width))
        self.ln_final = _LN(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim, dtype=torch.float16))
        self.initialize_parameters()
        for p in self.parameters():
            p.requires_grad_(False)

    def initialize_parameters(self):
        """text_encoder.py:296-323 (distributions only: real weights come from the CLIP checkpoint)."""
        t = self.transformer
        nn.init.normal_(self.token_embedding.weight, std=0.02)
        nn.init.normal_(self.positional_embedding, std=0.01)
        proj_std = (t.width ** -0.5) * ((2 * t.layers) ** -0.5)
        attn_std = t.width ** -0.5
        fc_std = (2 * t.width) ** -0.5

        def normal16(p, std):
            p.data.copy_(torch.randn(p.shape) * std)

        for b in t.resblocks:
            normal16(b.attn.in_proj_weight, attn_std)
            normal16(b.attn.out_proj.weight, proj_std)
            normal16(b.mlp.c_fc.weight, fc_std)
            normal16(b.mlp.c_proj.weight, proj_std)
        normal16(self.text_projection, t.width ** -0.5)

    @property
    def dtype(self):
        return torch.float16

    @torch.no_grad()
    def forward(self, text):
        """text (N, context_length) int64 token ids -> {"last_hidden_state": (N, L, W) fp16}."""
        x = kn.clip_embed(text, self.token_embedding.weight, self.positional_embedding)
        for blk in self.transformer.resblocks:
            x = blk(x)
        N, L, d = x.shape
        x = kn.layernorm_f16(x.view(N * L, d), self.ln_final.weight, self.ln_final.bias).view(N, L, d)
        # text_encoder.py:352-354: the features of the end-of-text token (the largest id) through text_projection,
        # fp16 like the rest of the tower (MESM itself only reads last_hidden_state)
        eos = x[torch.arange(N, device=x.device), text.argmax(dim=-1)]
        # the transposed operand is rebuilt per call: a (W x E) one-off, and a cache keyed on the source tensor would have
        # to see load_state_dict / convert_weights / .half() / in-place updates of text_projection
        pooled = kn.gemm_f16(eos.contiguous(), self.text_projection.detach().to(torch.float16).t().contiguous())
        return dict(last_hidden_state=x, pooler_output=pooled)


def convert_weights(model):
    """text_encoder.py:373-394: the Linear / attention / projection tensors to fp16 (this build's modules are
    created in that state already; calling it again is harmless and keeps call sites of the reference valid)."""
    for m in model.modules():
        if isinstance(m, _Linear16):
            m.weight.data = m.weight.data.half()
            m.bias.data = m.bias.data.half()
        if isinstance(m, _MHA16):
            m.in_proj_weight.data = m.in_proj_weight.data.half()
            m.in_proj_bias.data = m.in_proj_bias.data.half()
        if isinstance(m, CLIPTextEncoder):
            m.text_projection.data = m.text_projection.data.half()


def clip_text_encoder_from_state_dict(state_dict):
    """runner.py:166-187 (build_CLIP_text_encoder): every size is read from the checkpoint."""
    embed_dim = state_dict["text_projection"].shape[1]
    context_length = state_dict["positional_embedding"].shape[0]
    vocab_size = state_dict["token_embedding.weight"].shape[0]
    width = state_dict["ln_final.weight"].shape[0]
    heads = width // 64
    layers = len(set(k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks")))
    model = CLIPTextEncoder(embed_dim, context_length, vocab_size, width, heads, layers)
    sd = {k: v for k, v in state_dict.items() if k not in ("input_resolution", "context_length", "vocab_size")}
    convert_weights(model)
    model.load_state_dict(sd)
    return model.eval()


def build_CLIP_text_encoder(path):
    return clip_text_encoder_from_state_dict(torch.load(path, map_location="cpu"))


class GloveTextEncoder(nn.Module):
    """text_encoder.py:432-454: a frozen (len(vocab), 300) embedding filled from GloVe vectors."""

    def __init__(self, vocab, glove=None, dim=300):
        super().__init__()
        n = len(vocab) if not isinstance(vocab, int) else vocab
        dim = glove.dim if glove is not None else dim
        self.emb = nn.Embedding(num_embeddings=n, embedding_dim=dim)
        for p in self.emb.parameters():
            p.requires_grad = False
        if glove is not None:
            for w, i in vocab.wtoi.items():
                self.emb.weight.data[i, :] = glove.get(w)

    def forward(self, word_ids):
        return kn.embed_rows(word_ids, self.emb.weight)


class GloVe:
    """text_encoder.py:397-429: {word: 300-d vector} read from a glove.*.txt file."""

    def __init__(self, glove_path, dim=300):
        import numpy as np
        self.dim = dim
        self.glove = {}
        with open(glove_path, "r") as f:
            for line in f:
                parts = line.split()
                word = " ".join(parts[:len(parts) - dim])  # some words include spaces
                self.glove[word] = torch.from_numpy(np.array(parts[-dim:], dtype=np.float32))
        self.glove["<PAD>"] = torch.zeros(dim)
        self.glove["<UNK>"] = torch.randn(dim)

    def get(self, word):
        return self.glove[word] if word in self.glove else self.glove["<UNK>"]

    def contains(self, word):
        return word in self.glove


def build_GloVe_text_encoder(glove_path, vocab):
    return GloveTextEncoder(vocab, GloVe(glove_path))
