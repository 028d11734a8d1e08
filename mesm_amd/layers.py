"""Parameter containers + fused forward blocks of the MESM transformer stacks.

Module / parameter names reproduce the reference's state_dict keys
so checkpoints interchange; the forward passes are written on mesm_amd.ops (HIP kernels),
batch-first, and never call torch.nn.functional compute ops.
"""
import copy
import math

import torch
from torch import nn

from . import kernels as kn
from . import ops
from .ops import drop_state


class ParamLinear(nn.Module):
    """weight (out, in) + bias (out) with nn.Linear's default initialisation; no forward —
    the owning block passes the tensors to a fused op."""

    def __init__(self, in_f, out_f):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_f, in_f))
        self.bias = nn.Parameter(torch.empty(out_f))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_f) if in_f > 0 else 0
        nn.init.uniform_(self.bias, -bound, bound)


class ParamLayerNorm(nn.Module):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))

    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias)


class PReLUParam(nn.Module):
    """nn.PReLU(): one learnable slope, init 0.25 (runner.py:199 'prelu', transformer.py:902-903)."""

    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.full((1,), 0.25))


class PackedMHAParams(nn.Module):
    """Parameters of nn.MultiheadAttention(d, h): in_proj_{weight,bias}, out_proj.{weight,bias}."""

    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = ParamLinear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class OutProjOnly(nn.Module):
    """The custom decoder MultiheadAttention (attention.py:61-182) only owns out_proj."""

    def __init__(self, vdim):
        super().__init__()
        self.out_proj = ParamLinear(vdim, vdim)
        nn.init.zeros_(self.out_proj.bias)


class MLPHead(nn.Module):
    """MLP (model.py:397-409 / transformer.py:21-33): ReLU between layers."""

    def __init__(self, in_d, hid, out_d, n):
        super().__init__()
        dims = [in_d] + [hid] * (n - 1) + [out_d]
        self.layers = nn.ModuleList(ParamLinear(a, b) for a, b in zip(dims[:-1], dims[1:]))

    def forward(self, x):
        n = len(self.layers)
        for i, l in enumerate(self.layers):
            x = ops.linear(x, l.weight, l.bias, relu=i < n - 1)
        return x


def mlp_heads_steps(pairs, extra=()):
    """Chain fragment: several independent MLPHeads (head, input) layer by layer, the i-th layers of all heads forked
    in one round (one grouped launch forward AND backward); `extra`: further independent calls for the first round.
    -> (outputs of the heads, results of the extra calls)"""
    xs = [x for _, x in pairs]
    depth = max(len(h.layers) for h, _ in pairs)
    ex = []
    for i in range(depth):
        calls = []
        for k, (h, _) in enumerate(pairs):
            if i < len(h.layers):
                l = h.layers[i]
                calls.append(ops.linear_call(xs[k], l.weight, l.bias, relu=i < len(h.layers) - 1))
            else:
                calls.append(None)
        if i == 0:
            calls += list(extra)
        res = yield calls
        for k in range(len(pairs)):
            if calls[k] is not None:
                xs[k] = res[k]
        if i == 0:
            ex = res[len(pairs):]
    return xs, ex


def mlp_heads_grouped(pairs):
    """Run several independent MLPHeads (head, input) layer by layer, the i-th layers of all heads
    in one grouped GEMM launch."""
    return ops.seq(mlp_heads_steps(pairs))[0]


class LinearLayer(nn.Module):
    """LN -> dropout -> Linear -> [ReLU] (model.py:412-434); the dropout is applied by the LayerNorm
    kernel's store (once per element; as a GEMM operand transform it was re-hashed by every
    column tile) and replayed on dy by the LayerNorm backward."""

    def __init__(self, in_f, out_f, dropout, relu):
        super().__init__()
        self.LayerNorm = ParamLayerNorm(in_f)
        self.net = nn.ModuleList([nn.Identity(), ParamLinear(in_f, out_f)])
        self.p = dropout
        self.relu = relu

    def steps(self, x):
        """the layer as a chain of block calls (ops.lockstep)"""
        ln = self.LayerNorm
        x = yield ops.layer_norm_call(x, ln.weight, ln.bias, drop=drop_state.next(self.p))
        lin = self.net[1]
        return (yield ops.linear_call(x, lin.weight, lin.bias, relu=self.relu))

    def forward(self, x):
        return ops.seq(self.steps(x))


class T2VLayer(nn.Module):
    """T2V_TransformerEncoderLayer[_TwoMLP].forward_post (transformer.py:508-540, 573-612)."""

    def __init__(self, d, h, ff, dropout, two_mlp=False):
        super().__init__()
        self.self_attn = PackedMHAParams(d)
        self.linear1 = ParamLinear(d, ff)
        self.linear2 = ParamLinear(ff, d)
        self.norm1 = ParamLayerNorm(d)
        self.norm2 = ParamLayerNorm(d)
        self.activation = PReLUParam()
        if two_mlp:
            self.linear1_1 = ParamLinear(d, ff)
            self.linear2_1 = ParamLinear(ff, d)
            self.norm1_1 = ParamLayerNorm(d)
            self.norm2_1 = ParamLayerNorm(d)
        self.two_mlp = two_mlp
        self.nhead = h
        self.p = dropout

    def steps(self, txt, vid, pos_txt, pos_vid, txt_pad, vid_pad, is_mlm=False, group=0, vid_p=None,
              out_pos=None, kv_share=None, join_qp=False, txt_p=None):
        """The layer as a chain of three block calls (ops.lockstep).
        txt_p: txt + pos_txt formed without autograd (ops.gather_add): the key projection reads it as a plain operand,
        its gradient joins d txt inside the block (like join_qp on the query side) and d txt can be shared across layers.
        join_qp: vid_p was formed without autograd (vid + a constant): its gradient joins d vid inside the block.
        kv_share: ops.GradShare of the stack's layers for d txt (no key position term).
        vid_p: vid + pos_vid when the producer of vid has already written it (None: formed here).
        out_pos: also return output + out_pos (the next block's query) -> (out, out_p)."""
        sa = self.self_attn
        if vid_p is None and pos_vid is not None:
            vid_p = vid + pos_vid
        keyless = pos_txt is None or txt_p is not None  # (no position term left on the key side of the backward)
        x = yield ops.mha_call(vid, vid_p, txt, pos_txt if txt_p is None else None, vid, sa.in_proj_weight, sa.in_proj_bias,
                               sa.out_proj.weight, sa.out_proj.bias, self.nhead, kpad=txt_pad, qpad=vid_pad,
                               attn_drop=drop_state.next(self.p), out_drop=drop_state.next(self.p), group=group,
                               kv_share=kv_share if keyless else None, join_qp=join_qp, xkp=txt_p)
        alt = self.two_mlp and is_mlm
        n1, n2 = (self.norm1_1, self.norm2_1) if alt else (self.norm1, self.norm2)
        l1, l2 = (self.linear1_1, self.linear2_1) if alt else (self.linear1, self.linear2)
        # x + FFN(LN1(x)) (pre-norm FFN, transformer.py:536-538): one block, the two routes of dx meet in-kernel
        y = yield ops.norm_ffn_call(x, n1.weight, n1.bias, l1.weight, l1.bias, self.activation.weight, l2.weight,
                                    l2.bias, mid_drop=drop_state.next(self.p), out_drop=drop_state.next(self.p))
        return (yield ops.layer_norm_call(y, n2.weight, n2.bias, add=out_pos))

    def forward(self, *a, **kw):
        return ops.seq(self.steps(*a, **kw))


def _clones(m, n):
    return nn.ModuleList([copy.deepcopy(m) for _ in range(n)])


class T2VStack(nn.Module):
    """T2V_TransformerEncoder (transformer.py:208-242)."""

    def __init__(self, layer, n):
        super().__init__()
        self.layers = _clones(layer, n)

    def steps(self, txt, vid, pos_txt, pos_vid, txt_pad, vid_pad, is_mlm=False, group=0, vid_p=None,
              out_pos=None, join_vid_p=False, txt_p=None):
        n = len(self.layers)
        # every layer reads the same txt: their d txt shares are summed by the dX GEMMs' epilogues, not by autograd
        share = ops.GradShare(n) if (n > 1 and (pos_txt is None or txt_p is not None) and torch.is_grad_enabled()
                                     and txt.requires_grad) else None
        for i, l in enumerate(self.layers):
            op = out_pos if i == n - 1 else pos_vid
            res = yield from l.steps(txt, vid, pos_txt, pos_vid, txt_pad, vid_pad, is_mlm, group, vid_p=vid_p,
                                     out_pos=op, kv_share=share, join_qp=join_vid_p and i == 0, txt_p=txt_p)
            vid, vid_p = res if op is not None else (res, None)
        return (vid, vid_p) if out_pos is not None else vid

    def forward(self, *a, **kw):
        return ops.seq(self.steps(*a, **kw))


def _xavier_(module):
    for p in module.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)


class T2VEncoder(nn.Module):
    """T2VEncoder / T2VEncoder_TwoMLP (transformer.py:62-116): xavier on every matrix."""

    def __init__(self, d, h, n_layers, ff, dropout, two_mlp=False):
        super().__init__()
        self.t2v_encoder = T2VStack(T2VLayer(d, h, ff, dropout, two_mlp), n_layers)
        _xavier_(self)
        self.d_model, self.nhead = d, h

    def forward(self, txt, vid, pos_txt, pos_vid, txt_pad, vid_pad, is_mlm=False, group=0, vid_p=None,
                out_pos=None):
        """group: rows per independent batch when the positive and the negative pass are stacked
        along the batch dimension (the Q1 mask rule wraps inside a group).  vid_p / out_pos: see T2VLayer."""
        return self.t2v_encoder(txt, vid, pos_txt, pos_vid, txt_pad, vid_pad, is_mlm, group, vid_p=vid_p,
                                out_pos=out_pos)

    def steps(self, txt, vid, pos_txt, pos_vid, txt_pad, vid_pad, is_mlm=False, group=0, vid_p=None, out_pos=None,
              join_vid_p=False, txt_p=None):
        return self.t2v_encoder.steps(txt, vid, pos_txt, pos_vid, txt_pad, vid_pad, is_mlm, group, vid_p=vid_p,
                                      out_pos=out_pos, join_vid_p=join_vid_p, txt_p=txt_p)


class EncoderLayer(nn.Module):
    """TransformerEncoderLayer.forward_post (transformer.py:637-650)."""

    def __init__(self, d, h, ff, dropout):
        super().__init__()
        self.self_attn = PackedMHAParams(d)
        self.linear1 = ParamLinear(d, ff)
        self.linear2 = ParamLinear(ff, d)
        self.norm1 = ParamLayerNorm(d)
        self.norm2 = ParamLayerNorm(d)
        self.activation = PReLUParam()
        self.nhead, self.p = h, dropout

    def forward(self, src, pos, pad, src_p=None, out_pos=None):
        sa = self.self_attn
        if src_p is None and pos is not None:
            src_p = src + pos
        x = ops.mha(src, src_p, None, None, src, sa.in_proj_weight, sa.in_proj_bias,
                    sa.out_proj.weight, sa.out_proj.bias, self.nhead, kpad=pad,
                    attn_drop=drop_state.next(self.p), out_drop=drop_state.next(self.p),
                    self_attn=True)
        x = self.norm1(x)
        y = ops.ffn(x, x, self.linear1.weight, self.linear1.bias, self.activation.weight,
                    self.linear2.weight, self.linear2.bias, mid_drop=drop_state.next(self.p),
                    out_drop=drop_state.next(self.p))
        return ops.layer_norm(y, self.norm2.weight, self.norm2.bias, add=out_pos)


class EncoderStack(nn.Module):
    def __init__(self, layer, n):
        super().__init__()
        self.layers = _clones(layer, n)

    def forward(self, src, pos, pad, src_p=None):
        n = len(self.layers)
        for i, l in enumerate(self.layers):
            op = pos if i < n - 1 else None  # the next layer's query = this output + pos, from the same kernel
            res = l(src, pos, pad, src_p=src_p, out_pos=op)
            src, src_p = res if op is not None else (res, None)
        return src


class DecoderLayer(nn.Module):
    """TransformerDecoderLayer.forward (transformer.py:723-797), conditional-DETR style cross
    attention with per-head [content || positional] queries and keys."""

    def __init__(self, d, h, ff, dropout):
        super().__init__()
        for name in ("sa_qcontent_proj", "sa_qpos_proj", "sa_kcontent_proj", "sa_kpos_proj",
                     "sa_v_proj"):
            setattr(self, name, ParamLinear(d, d))
        self.self_attn = OutProjOnly(d)
        self.norm1 = ParamLayerNorm(d)
        for name in ("ca_qcontent_proj", "ca_qpos_proj", "ca_kcontent_proj", "ca_kpos_proj",
                     "ca_v_proj", "ca_qpos_sine_proj"):
            setattr(self, name, ParamLinear(d, d))
        self.cross_attn = OutProjOnly(d)
        self.linear1 = ParamLinear(d, ff)
        self.linear2 = ParamLinear(ff, d)
        self.norm2 = ParamLayerNorm(d)
        self.norm3 = ParamLayerNorm(d)
        self.activation = PReLUParam()
        self.nhead, self.p = h, dropout

    # projections that share their input, fused by layout (mesm_amd/gradbuf.py Pack)
    PACKS = {"sa_t": ("sa_qcontent_proj", "sa_kcontent_proj", "sa_v_proj"),
             "sa_p": ("sa_qpos_proj", "sa_kpos_proj"),
             "ca_kv": ("ca_kcontent_proj", "ca_v_proj")}

    def steps(self, tgt, memory, mem_pad, pos, query_pos, qsine, is_first, pack, mem_share=None, tgt_res=None):
        """The layer as a chain (ops.lockstep).  pack(key) -> (weight, bias) views of a parameter pack of THIS layer
        (MESM.pack).  The query-side position projections of the cross attention do not depend on the self-attention
        block: they fork beside it."""
        L = ops.linear_call
        h = self.nhead
        wt, bt = pack("sa_t")
        wp, bp = pack("sa_p")
        a, qs, qpp = yield [
            ops.dec_self_attn_call(tgt, query_pos, wt, bt, wp, bp, h, drop=drop_state.next(self.p)),
            L(qsine, self.ca_qpos_sine_proj.weight, self.ca_qpos_sine_proj.bias),
            L(query_pos, self.ca_qpos_proj.weight, self.ca_qpos_proj.bias) if is_first else None]
        so = self.self_attn.out_proj
        # (tgt_res: a second alias of tgt for the residual route, ops.fork: its gradient joins the others' in one launch)
        x = yield L(a, so.weight, so.bias, residual=tgt if tgt_res is None else tgt_res, out_drop=drop_state.next(self.p))
        # (norm1's output feeds the cross-attention block and the residual of its output projection: two aliases, their
        # gradients meet in the LayerNorm backward kernel)
        tgt, tgt_r = yield ops.layer_norm_call(x, self.norm1.weight, self.norm1.bias, fork=True)
        wkv, bkv = pack("ca_kv")
        a = yield ops.dec_cross_attn_call(tgt, qs, qpp, memory, pos, mem_pad, self.ca_qcontent_proj.weight,
                                          self.ca_qcontent_proj.bias, wkv, bkv, self.ca_kpos_proj.weight,
                                          self.ca_kpos_proj.bias, is_first, h, drop=drop_state.next(self.p),
                                          mem_share=mem_share)
        co = self.cross_attn.out_proj
        x = yield L(a, co.weight, co.bias, residual=tgt_r, out_drop=drop_state.next(self.p))
        tgt = yield ops.layer_norm_call(x, self.norm2.weight, self.norm2.bias)
        y = yield ops.ffn_call(tgt, tgt, self.linear1.weight, self.linear1.bias, self.activation.weight,
                               self.linear2.weight, self.linear2.bias, mid_drop=drop_state.next(self.p),
                               out_drop=drop_state.next(self.p))
        return (yield ops.layer_norm_call(y, self.norm3.weight, self.norm3.bias))

    def forward(self, *a, **kw):
        return ops.seq(self.steps(*a, **kw))


def inverse_sigmoid(x, eps=1e-3):
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


class Decoder(nn.Module):
    """TransformerDecoder (transformer.py:280-420): DAB-style iterative reference refinement."""

    def __init__(self, layer, n, d):
        super().__init__()
        self.layers = _clones(layer, n)
        self.num_layers = n
        self.norm = ParamLayerNorm(d)
        self.query_scale = MLPHead(d, d, d, 2)
        self.ref_point_head = MLPHead(d, d, d, 2)
        self.bbox_embed = MLPHead(d, d, 2, 3)
        self.ref_anchor_head = MLPHead(d, d, 1, 2)
        self.d_model = d
        # keep_query_pos=False: ca_qpos_proj exists on layer 0 only (transformer.py:329-331)
        for i in range(1, n):
            self.layers[i].ca_qpos_proj = None

    def steps(self, memory, mem_pad, pos, refpoints_unsigmoid, pack):
        """The decoder as a chain (ops.lockstep).  What depends on a layer's output only -- the box head, the final norm
        of that output, and the NEXT layer's anchor / scale heads -- forks into shared rounds; the next layer's
        reference-point head has to wait for the box head (its sine embedding comes from the refined reference)."""
        n = memory.shape[0]
        nq = refpoints_unsigmoid.shape[0]
        d = self.d_model
        nl = len(self.layers)
        dev = memory.device
        # the two stacked outputs (transformer.py:411-415) are written in place by the kernels that produce their slices
        hs_buf = torch.empty((nl, n, nq, d), device=dev, dtype=torch.float32)
        refs_buf = torch.empty((nl, n, nq, 2), device=dev, dtype=torch.float32)
        # sigmoid(refpoints)[None].expand(n, nq, 2) into slot 0 and its sine embedding: one kernel (backward: one more,
        # which also sums the gradients of ref's three consumers)
        ref_s, ref_w, ref_u, qsine, qsine_m = ops.ref_init_sine(refpoints_unsigmoid, n, d, ops.Slot(refs_buf, 0))
        refs = [ref_s]
        out = kn.zeros((n, nq, d), dev) if torch.is_grad_enabled() else torch.zeros(n, nq, d, device=dev)
        inter = []
        # layer 0: anchor head beside the reference-point head (query_scale is 1 on layer 0, transformer.py:366-369)
        (query_pos, anchor), _ = yield from mlp_heads_steps([(self.ref_point_head, qsine), (self.ref_anchor_head, out)])
        # qsine * (sigmoid(ref_anchor_head(out)) / ref_width): one kernel
        qsine = ops.qsine_scale(qsine_m, None, anchor, ref_w)
        # every layer reads the same memory: their d memory shares are summed by the dX GEMMs' epilogues
        mem_share = ops.GradShare(nl) if (nl > 1 and torch.is_grad_enabled() and memory.requires_grad) else None
        out_res = None
        ref = ref_u
        for li, layer in enumerate(self.layers):
            out = yield from layer.steps(out, memory, mem_pad, pos, query_pos, qsine, li == 0,
                                         lambda key, li=li: pack("dec%d.%s" % (li, key)), mem_share=mem_share,
                                         tgt_res=out_res)
            norm = ops.layer_norm_call
            if li + 1 == nl:
                # the last layer's box refinement is DEAD in the reference: new_reference_points is neither appended
                # (transformer.py:395-396) nor read again, so bbox_embed(output) of that layer reaches no output and no
                # loss -- three GEMM launches and a ref_update that this build used to run (round 6)
                inter.append((yield norm(out, self.norm.weight, self.norm.bias, slot=ops.Slot(hs_buf, li))))
                break
            # a layer's output has six consumers (box head, final norm, the next layer's anchor / scale heads, the
            # next layer's self-attention and its residual): their gradients meet in one launch (ops.fork)
            o_box, o_norm, o_anchor, o_scale, o_next, out_res = ops.fork(out, 6)
            heads = [(self.bbox_embed, o_box), (self.ref_anchor_head, o_anchor), (self.query_scale, o_scale)]
            res, ex = yield from mlp_heads_steps(
                heads, extra=[norm(o_norm, self.norm.weight, self.norm.bias, slot=ops.Slot(hs_buf, li))])
            out = o_next
            inter.append(ex[0])
            # ONE kernel: new_ref = sigmoid(bbox_embed(out) + inverse_sigmoid(ref)) into its slot, and from its detached
            # value the next layer's sine embedding and qsine * query_scale(out) * (sigmoid(ref_anchor_head(out)) / width)
            new_ref, qsine_raw, qsine = ops.ref_step(res[0], ref, res[2], res[1], d, ops.Slot(refs_buf, li + 1))
            refs.append(new_ref)
            ref = new_ref.detach()
            (query_pos,), _ = yield from mlp_heads_steps([(self.ref_point_head, qsine_raw)])
        return ops.stacked(hs_buf, inter), ops.stacked(refs_buf, refs)

    def forward(self, *a, **kw):
        return ops.seq(self.steps(*a, **kw))


class Transformer(nn.Module):
    """Transformer (transformer.py:119-205): 2 self-attention encoder layers over
    [global token || video] + the moment-query decoder."""

    def __init__(self, d, h, num_queries, enc_layers, dec_layers, ff, dropout):
        super().__init__()
        self.encoder = EncoderStack(EncoderLayer(d, h, ff, dropout), enc_layers)
        self.decoder = Decoder(DecoderLayer(d, h, ff, dropout), dec_layers, d)
        _xavier_(self)
        self.d_model, self.nhead = d, h
        self.dim_feedforward, self.dropout = ff, dropout
        self.num_queries = num_queries

    def forward(self, src, vid_pad, query_embed, pos, g_tok, g_pos, run_decoder=True, n_dec=None, pack=None):
        """src (N, L, d); vid_pad (N, L) True = padding.  The global token is prepended as a
        MASKED key (transformer.py:185-186): it pools, nobody attends to it.  n_dec: only the first
        n_dec rows of the batch go through the decoder (the negative pass stacked behind them
        stops after the encoder: its decoder output is discarded at model.py:295)."""
        n = src.shape[0]
        # [global token ; video], its position embeddings, their sum (the first layer's query) and the key
        # padding mask [True ; vid_pad] from one launch (transformer.py:185-188, model.py:236-238)
        x, p, xp, pad = ops.prepend(g_tok, src, ptok=g_pos, pos=pos, pad=vid_pad, first_pad=True)
        mem = self.encoder(x, p, pad, src_p=xp)
        nd = n if n_dec is None else n_dec
        if not run_decoder:
            mem_g, mem_l = ops.split_token(mem)
            return None, None, mem_l, mem_g
        # memory_global, memory_local and the decoder's (positive-half) copy of memory_local, one launch
        mem_g, mem_l, mem_d = ops.split_token(mem, nd)
        hs, refs = self.decoder(mem_d, vid_pad[:nd].contiguous(), pos[:nd].contiguous(), query_embed, pack)
        return hs, refs, mem_l, mem_g
