"""MESM (model/model.py:16-394 of the reference) on the gfx950 kernels.

Same constructor semantics, call signature, returned dict and state_dict keys as the
reference module, so `runner.build_model` / train.py / eval.py can use it unchanged
(SURVEY.md §8b).  Differences that do not change any output or gradient:
  * dead compute of the reference is skipped: the negative pass runs the encoder
    only (its decoder output is discarded at model.py:295), SegSenRecon's unused position
    embedding is not computed;
  * all host-side decisions (group splits, negative indices, MLM word choice, gather maps) are
    made once per call in `make_plan` from CPU copies of the masks, so the device work is a
    fixed sequence of kernels (capturable in a HIP graph for a fixed batch shape).
"""
import contextlib
import os
import weakref

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import draws
from . import kernels as kn
from . import ops
from .autograph import AutoGraph, AutoOutputs
from .gradbuf import GradBuffer
from .layers import (LinearLayer, MLPHead, ParamLayerNorm, ParamLinear, T2VLayer, T2VStack,
                     inverse_sigmoid)
from .ops import drop_state
from .text_encoder import CLIPTextEncoder, GloveTextEncoder


_SCOPES = os.environ.get("MESM_SCOPES") == "1"
_MLM_HEAD_LAG = int(os.environ.get("MESM_MLM_HEAD_LAG", "1"))  # (A/B switch: 0 = the round-5 schedule)


def _scope(name):
    """Profiler region (tools/count_kernels.py); a no-op unless MESM_SCOPES=1."""
    return torch.profiler.record_function("R:" + name) if _SCOPES else contextlib.nullcontext()


def _pool_spent(_g):
    kn.zero_pool.backward_reached_inputs()


class TrainablePositionalEncoding(nn.Module):
    """position_encoding.py:10-32: dropout(LayerNorm(x + E[0:L])).  Bypassed while use_txt_pos=False
    (model.py:169-172, every shipped config), like in the reference."""

    def __init__(self, max_pos, d, dropout):
        super().__init__()
        self.position_embeddings = nn.Embedding(max_pos, d)
        self.LayerNorm = ParamLayerNorm(d)
        self.p = dropout

    def forward(self, x):
        L = x.shape[1]
        if L > self.position_embeddings.num_embeddings:
            raise IndexError("txt_position_embed: %d positions, sequence of %d"
                             % (self.position_embeddings.num_embeddings, L))  # nn.Embedding's own error class
        ln = self.LayerNorm
        return ops.layer_norm(x + self.position_embeddings.weight[:L], ln.weight, ln.bias,
                              drop=drop_state.next(self.p))


class SegSenRecon(nn.Module):
    """SS-MESM reconstructor (model.py:437-503)."""

    def __init__(self, input_dropout, d, h, n_layers, ff, dropout):
        super().__init__()
        self.masked_sent_token = nn.Parameter(torch.zeros(d))
        self.recon_trans = T2VStack(T2VLayer(d, h, ff, dropout), n_layers)
        self.output_sent_proj = nn.Sequential(LinearLayer(d, d, input_dropout, True),
                                              LinearLayer(d, d, input_dropout, False))


class Plan:
    """Host-side decisions of one forward call, as device index tensors."""
    pass


class MESM(nn.Module):
    def __init__(self, text_encoder, enhance_encoder, t2v_encoder, transformer,
                 vid_position_embed, txt_position_embed, txt_dim, vid_dim, num_queries,
                 input_dropout, aux_loss=False, max_video_l=75, max_words_l=32, normalize_txt=True,
                 use_txt_pos=False, span_loss_type="l1", n_input_proj=2, rec_fw=False,
                 vocab_size=1111, rec_ss=False, num_recss_layers=2, share_MLP=True):
        super().__init__()
        if text_encoder is not None and not isinstance(text_encoder, (CLIPTextEncoder, GloveTextEncoder)):
            raise NotImplementedError("text_encoder must be a mesm_amd CLIPTextEncoder / GloveTextEncoder or None")
        if span_loss_type != "l1":
            raise NotImplementedError("span_loss_type 'ce' raises in the reference as well")
        self.text_encoder = text_encoder  # frozen (model.py:30-33)
        if text_encoder is not None:
            for p in text_encoder.parameters():
                p.requires_grad_(False)
        self.enhance_encoder = enhance_encoder
        self.t2v_encoder = t2v_encoder
        self.transformer = transformer
        self.vid_position_embed = vid_position_embed  # None: the sine kernel has no parameters
        self.txt_position_embed = txt_position_embed
        self.num_queries = num_queries
        d = transformer.d_model
        self.hidden_dim = d
        self.span_loss_type = span_loss_type
        self.max_video_l, self.max_words_l = max_video_l, max_words_l
        self.normalize_txt = normalize_txt
        self.use_txt_pos = use_txt_pos
        self.n_input_proj = n_input_proj
        self.aux_loss = aux_loss
        self.share_MLP = share_MLP
        self.span_embed = MLPHead(d, d, 2, 3)
        self.class_embed = ParamLinear(d, 2)
        self.query_embed = nn.Embedding(num_queries, 2)
        relu = [True] * 3
        relu[n_input_proj - 1] = False
        dims_t = [txt_dim, d, d]
        dims_v = [vid_dim, d, d]
        self.input_txt_proj = nn.Sequential(*[LinearLayer(dims_t[i], d, input_dropout, relu[i])
                                              for i in range(n_input_proj)])
        self.input_vid_proj = nn.Sequential(*[LinearLayer(dims_v[i], d, input_dropout, relu[i])
                                              for i in range(n_input_proj)])
        self.saliency_proj1 = ParamLinear(d, d)
        self.saliency_proj2 = ParamLinear(d, d)
        self.global_rep_token = nn.Parameter(torch.randn(d))
        self.global_rep_pos = nn.Parameter(torch.randn(d))
        self.rec_fw = rec_fw
        # model.py:73-79: CLIP's BPE vocabulary carries three special ids, the GloVe / feature path one
        num_classes = vocab_size + 3 if isinstance(text_encoder, CLIPTextEncoder) else vocab_size + 1
        if rec_fw:
            self.masked_token = nn.Parameter(torch.zeros(txt_dim))
            self.unknown_token = nn.Parameter(torch.zeros(txt_dim))
            self.output_txt_proj = nn.Sequential(LinearLayer(d, d, input_dropout, True),
                                                 ParamLinear(d, num_classes))
        self.rec_ss = rec_ss
        if rec_ss:
            self.ss_reconstructor = SegSenRecon(input_dropout, d, transformer.nhead, num_recss_layers,
                                                transformer.dim_feedforward, transformer.dropout)
        self._gradbuf = None
        self._flat_params = None
        self._step = 0
        self._flat_checked = -1
        self._auto = None

    # ------------------------------------------------------------------ infrastructure
    def gradbuf(self):
        if self._gradbuf is None:
            packs = {}
            for i, layer in enumerate(self.transformer.decoder.layers):
                pre = "transformer.decoder.layers.%d." % i
                for key, mods in layer.PACKS.items():  # projections of one input, fused by layout (gradbuf.Pack)
                    for kind in ("weight", "bias"):
                        packs["dec%d.%s.%s" % (i, key, kind)] = [pre + m + "." + kind for m in mods]
            self._gradbuf = GradBuffer([(n, p) for n, p in self.named_parameters() if p.requires_grad], packs)
        return self._gradbuf

    def parameters(self, recurse=True):
        """nn.Module.parameters with the (fixed) parameter list remembered: the module-tree walk costs 0.5 ms per call over
        this model's 273 + frozen parameters, and the reference's loop calls it every step (clip_grad_norm_(model.parameters(),
        ...), train.py:70-71).  Same objects, same order; forgotten whenever the module tree or the tensors may change."""
        if not recurse:
            return super().parameters(recurse)
        lst = self.__dict__.get("_param_list")
        if lst is None:
            lst = list(super().parameters(True))
            self.__dict__["_param_list"] = lst
        return iter(lst)

    def _apply(self, fn, *a, **kw):
        self.__dict__.pop("_param_list", None)
        self.__dict__["_addr_gen"] = self.__dict__.get("_addr_gen", 0) + 1  # (tensors may move: graphs re-check addresses)
        return super()._apply(fn, *a, **kw)

    def register_parameter(self, name, param):
        self.__dict__.pop("_param_list", None)
        return super().register_parameter(name, param)

    def add_module(self, name, module):
        self.__dict__.pop("_param_list", None)
        return super().add_module(name, module)

    def __setattr__(self, name, value):
        if isinstance(value, (nn.Module, nn.Parameter)) or name in self.__dict__.get("_modules", ()) \
                or name in self.__dict__.get("_parameters", ()):
            self.__dict__.pop("_param_list", None)
        super().__setattr__(name, value)

    def zero_grad(self, set_to_none=True):
        """nn.Module.zero_grad for the trainable parameters the flat gradient buffer knows (the frozen text encoder never
        gets gradients): the module-tree walk of the stock method costs 0.8 ms per eager step over 273 parameters"""
        gb = self._gradbuf
        if not set_to_none or gb is None:
            return super().zero_grad(set_to_none=set_to_none)
        for p in gb.params:
            p.grad = None

    def pack(self, key):
        """(weight, bias) views of a parameter pack in the flat parameter buffer (see gradbuf.Pack)."""
        gb = self.gradbuf()
        # (the 273-parameter address check of flat_params() once per step, not once per pack: 6 packs per forward)
        fp = self._flat_params if (self._flat_checked == self._step and self._flat_params is not None) else self.flat_params()
        self._flat_checked = self._step
        return gb.packs[key + ".weight"].weight(fp), gb.packs[key + ".bias"].weight(fp)

    def flat_params(self):
        """Move every trainable parameter into ONE flat fp32 buffer with the layout of the flat gradient
        buffer (gradbuf.py); every `param.data` becomes a view, so modules, state_dict() and checkpoints
        are unaffected.  The fused optimizer (optim.FlatAdamW) updates this buffer in one launch.  Done
        eagerly by build_model, i.e. BEFORE any HIP-graph capture: a captured step bakes the parameter
        addresses in, and GraphedStep refuses to replay once they have moved."""
        gb = self.gradbuf()
        dev = gb.params[0].device
        fp = self._flat_params
        if fp is not None and fp.device == dev and all(
                p.data_ptr() == fp.data_ptr() + 4 * off for p, off in zip(gb.params, gb.offsets)):
            return fp
        self.__dict__["_addr_gen"] = self.__dict__.get("_addr_gen", 0) + 1
        fp = torch.zeros(gb.numel, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, off in zip(gb.params, gb.offsets):
                view = fp[off:off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
        self._flat_params = fp
        return fp


    def _begin(self, device, is_training):
        if not device.type == "cuda":
            raise kn._lib.MesmError(
                "mesm_amd.MESM runs on an MI355X only (got %s); the CPU oracle in oracle/ is a "
                "test checker, not a fallback" % device)
        kn._lib.set_device_index(device.index if device.index is not None else torch.cuda.current_device())
        gb = self.gradbuf()
        gb.ensure(device)
        if torch.is_grad_enabled():
            gb.begin_step()
        self._step += 1
        drop_state.begin(self.training, torch.initial_seed() + self._step)
        if torch.is_grad_enabled():
            kn.zero_pool.begin(device)  # ONE fill for every zero-initialised scratch tensor of the step
        else:
            kn.zero_pool.idle()

    def _proj(self, seq, x):
        for m in seq:
            x = m(x)
        return x

    # ------------------------------------------------------------------ text encoding (model.py:103-152)
    def CLIP_encode_text(self, words_id, words_mask):
        """model.py:103-134: fp16 CLIP transformer -> fp32, first max_words_l tokens, pads zeroed, sentence =
        masked mean of the un-normalised words, both L2-normalised (eps 1e-5).  (The reference's detour through
        "cuda" for CPU inputs, quirk Q10, has no counterpart: this model only runs on the GPU.)"""
        hid = self.text_encoder(words_id)["last_hidden_state"]
        Lw = min(self.max_words_l, hid.shape[1])
        words, sent = kn.text_pool(hid, words_mask, Lw, self.normalize_txt)
        return words, sent, words_id[:, :Lw], words_mask[:, :Lw]

    def GloVe_encode_text(self, words_id, words_mask):
        """model.py:136-143."""
        emb = self.text_encoder(words_id)
        return kn.text_pool(emb, words_mask, emb.shape[1], self.normalize_txt)

    # ------------------------------------------------------------------ host RNG draws (draws.py)
    @staticmethod
    def draw_neg_index(groups):
        """sample_outclass_neg (utils/data_utils.py:113-124), the reference's RNG stream (draws.neg_index_reference)."""
        return draws.neg_index_reference(groups)

    @staticmethod
    def draw_masked_words(words_mask_cpu, words_weight):
        """_mask_words (model.py:361-384), the reference's RNG stream (draws.masked_words_reference)."""
        return draws.masked_words_reference(words_mask_cpu, words_weight)

    # ------------------------------------------------------------------ host-side plan
    real_groups = staticmethod(draws.real_groups)

    @staticmethod
    def draw_neg_padded(groups, n_valid):
        """the step's negative draw (draws.MODE: vectorized by default, MESM_DRAWS=reference for the reference's
        stream) over the REAL groups; padding pairs point at pair 0 (their rows are never read by a loss)"""
        return torch.from_numpy(draws.neg_index(groups, n_valid))

    def plan_arrays(self, vm, wm, groups, dataset_name, is_training, clip_mask=None, neg_index=None,
                    masked_words=None, words_weight=None, Lc_cap=None, Lss_cap=None, M_cap=None, n_valid=None):
        """All data-dependent host decisions of model.py:184-207, :260, :307-325 as numpy arrays
        ({name: array}, meta) -- pure host arithmetic on the (small) masks, no device work.

        vm (N, Lv) / wm (N, Lw) bool arrays (video / word validity), groups = queries per video group.
        Lc_cap / Lss_cap: pad the GT-clip gather (MLM branch) / the group-video gather (SS branch) to a fixed key
        length; the padding slots are masked keys, so results do not change, and a HIP graph captured with such
        a plan replays for every batch that fits (graphed.py).  A batch that does not fit raises ValueError."""
        vm, wm = np.asarray(vm, dtype=bool), np.asarray(wm, dtype=bool)
        N, Lv = vm.shape
        arr, meta = {}, {"groups": list(groups)}
        if neg_index is None:
            neg_index = draws.neg_index(groups, n_valid)
        arr["neg_index"] = np.asarray(neg_index, dtype=np.int64)
        # the key-padding forms of the two masks (what every attention takes) ride in the plan: no inversion launches
        arr["vpad"], arr["wpad"] = ~vm, ~wm
        if self.rec_ss:
            emask = np.concatenate([np.ones((N, 1), dtype=bool), wm], axis=1)  # [recon ; words] (model.py:221-224)
            arr["emask"] = emask
        if n_valid is not None:
            # the number of REAL pairs as a device scalar: modulus of the attention mask quirk, extent of every loss
            arr["n_valid"] = np.asarray([n_valid], dtype=np.int32)
        if self.rec_ss:
            M = max(groups)
            if M_cap is not None:  # sentence slots per pair padded to a fixed extent (masked queries / keys)
                if M > M_cap:
                    raise ValueError("make_plan: a video group has %d queries > M_cap %d" % (M, M_cap))
                M = M_cap
            starts = np.concatenate([[0], np.cumsum(groups)])
            gid = np.repeat(np.arange(len(groups)), groups)          # group of every pair
            slot = np.arange(N) - starts[gid]                        # position of the pair inside its group
            cols = np.arange(M)[None, :]
            sent_mask = cols < np.asarray(groups)[gid][:, None]
            arr["sent_src"] = np.where(sent_mask, starts[gid][:, None] + cols, 0).astype(np.int64)
            arr["sent_mask"] = sent_mask
            arr["sent_pad"] = ~sent_mask
            arr["sent_loc"] = cols == slot[:, None]
            arr["sent_slot"] = slot.astype(np.int64)
            arr["rows"] = np.arange(N, dtype=np.int64)
            # the masked slot of every pair in the (N * M)-row output of the reconstructor, and its inverse map
            ridx = np.arange(N) * M + slot
            rinv = np.full(N * M, -1, dtype=np.int64)
            rinv[ridx] = np.arange(N)
            arr["recon_idx"], arr["recon_inv"] = ridx.astype(np.int64), rinv
            if dataset_name == "qvhighlights":
                # all valid clips of the group's segments, concatenated, once per query (model.py:190-195)
                flat_valid = np.flatnonzero(vm.reshape(-1))
                offs = np.concatenate([[0], np.cumsum(vm.sum(1))])
                seg_len = offs[starts[1:]] - offs[starts[:-1]]       # clips per group
                Lss = int(seg_len.max())
                if Lss_cap is not None:
                    if Lss > Lss_cap:
                        raise ValueError("make_plan: the longest group video has %d clips > Lss_cap %d" % (Lss, Lss_cap))
                    Lss = Lss_cap
                c = np.arange(Lss)[None, :]
                vid_mask = c < seg_len[gid][:, None]
                src_pos = np.minimum(offs[starts[:-1]][gid][:, None] + c, len(flat_valid) - 1)
                arr["vid_src"] = np.where(vid_mask, flat_valid[src_pos], 0).astype(np.int64)
                arr["vid_mask"], arr["vid_pad"] = vid_mask, ~vid_mask
                # every group is one full-length pair (the common QVH case): the gather is the identity
                meta["vid_identity"] = bool(Lss == Lv and vid_mask.all()
                                            and np.array_equal(arr["vid_src"].reshape(-1), np.arange(N * Lv)))
                meta["has_vid_src"] = True
            elif dataset_name in ("charades", "charades-cg", "charades-cd", "tacos"):
                meta["vid_identity"], meta["has_vid_src"] = False, False
            else:
                raise NotImplementedError
        if self.rec_fw and is_training:
            cm = np.asarray(clip_mask, dtype=bool)
            lens = cm.sum(1)
            Lc = int(lens.max())
            if Lc_cap is not None:
                if Lc > Lc_cap:
                    raise ValueError("make_plan: a pair has %d ground-truth clips > Lc_cap %d" % (Lc, Lc_cap))
                Lc = Lc_cap
            flat = np.flatnonzero(cm.reshape(-1))
            offs = np.concatenate([[0], np.cumsum(lens)])
            c = np.arange(Lc)[None, :]
            cmask = c < lens[:, None]
            src = np.where(cmask, flat[np.minimum(offs[:-1][:, None] + c, max(len(flat) - 1, 0))], 0).astype(np.int64)
            # inverse of the GT-clip gather: source row (pair, clip) -> its slot in (N, Lc), -1 = not gathered
            # (the padding slots point at row 0 but are invalid: they must not claim it)
            cinv = np.full(N * Lv, -1, dtype=np.int64)
            cinv[src[cmask]] = np.arange(N * Lc).reshape(N, Lc)[cmask]
            arr["clip_src"], arr["clip_inv"], arr["clip_mask"], arr["clip_pad"] = src, cinv, cmask, ~cmask
            if masked_words is None:
                masked_words = draws.masked_words(wm, words_weight)
            arr["masked_words"] = np.asarray(masked_words, dtype=bool)
        return arr, meta

    @staticmethod
    def plan_from(views, meta):
        pl = Plan()
        for k, v in views.items():
            setattr(pl, k, v)
        pl.groups = meta["groups"]
        pl.vid_identity = meta.get("vid_identity", False)
        if not meta.get("has_vid_src", False):
            pl.vid_src = None
        return pl

    @torch.no_grad()
    def make_plan(self, video_mask, words_mask, num_clips, dataset_name, is_training, words_weight=None,
                  clip_mask=None, neg_index=None, masked_words=None, device=None, Lc_cap=None, Lss_cap=None,
                  n_valid=None):
        """plan_arrays on host copies of the masks, uploaded to `device` in one transfer (arena.Arena)."""
        from .arena import Arena
        device = device or video_mask.device
        _np = lambda t: None if t is None else (t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t))
        arr, meta = self.plan_arrays(_np(video_mask), _np(words_mask), [int(g) for g in num_clips.tolist()],
                                     dataset_name, is_training, clip_mask=_np(clip_mask), neg_index=_np(neg_index),
                                     masked_words=_np(masked_words), words_weight=words_weight, Lc_cap=Lc_cap,
                                     Lss_cap=Lss_cap, n_valid=n_valid)
        pl = self.plan_from(Arena(arr, device).views, meta)
        return pl

    # ------------------------------------------------------------------ forward
    def autograph(self, enabled=None, pad=None, pairs=None, max_graphs=None):
        """Graph replay behind the unchanged `model(...) / criterion(...) / loss.backward()` sequence (autograph.py):
        on by default (MESM_AUTOGRAPH=0 turns it off); `pad=(max_v_l, max_words_l), pairs=8` pads batches like
        graphed.StepCache so that a loader's ever-changing shapes replay from a handful of graphs."""
        if self._auto is None:
            self._auto = AutoGraph(self)
        return self._auto.configure(enabled, pad, pairs, max_graphs)

    def forward(self, video_feat, video_mask, words_id, words_mask, words_weight, num_clips, **kwargs):
        dev = video_feat.device
        is_training = kwargs["is_training"]
        auto = self._auto if self._auto is not None else self.autograph()
        if auto.eligible(self, {"video_feat": video_feat}, kwargs):
            batch = dict(video_feat=video_feat, video_mask=video_mask, words_id=words_id, words_mask=words_mask,
                         words_weight=words_weight, num_clips=num_clips)
            batch.update({k: v for k, v in kwargs.items() if k not in ("dataset_name", "is_training", "plan")})
            res = auto.forward(batch, kwargs["dataset_name"])
            if res is not None:
                return res
        elif torch.is_grad_enabled() and not auto.busy:
            auto.gen += 1
        side = auto.eager_stream(dev)
        if side is not None:
            # an eager visit of a shape that may be captured later runs on THE capture stream of the process: autograd pins
            # every AccumulateGrad node to the stream of its first use, and a capture on another stream would record the
            # hand-over as a second hardware queue (graphed.capture_stream)
            cur = torch.cuda.current_stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                out = self._forward(video_feat, video_mask, words_id, words_mask, words_weight, num_clips, **kwargs)
            cur.wait_stream(side)
            out._mesm_side = side
            return out
        return self._forward(video_feat, video_mask, words_id, words_mask, words_weight, num_clips, **kwargs)

    def _forward(self, video_feat, video_mask, words_id, words_mask, words_weight, num_clips, **kwargs):
        dev = video_feat.device
        is_training = kwargs["is_training"]
        auto = self._auto
        self._begin(dev, is_training)
        d, h = self.hidden_dim, self.transformer.nhead
        N, Lv = video_mask.shape

        with _scope("text"):
            if isinstance(self.text_encoder, CLIPTextEncoder):
                words, sent, words_id, words_mask = self.CLIP_encode_text(words_id, words_mask)
            elif isinstance(self.text_encoder, GloveTextEncoder):
                words, sent = self.GloVe_encode_text(words_id, words_mask)
            else:
                if words_id.dim() != 3:
                    raise NotImplementedError("token ids need a text encoder; pass (N, Lw, Dt) word features")
                # post_process_text (model.py:145-152), one kernel; word features carry no gradient
                words, words_mask, sent = kn.text_prep(words_id, self.normalize_txt)

        plan = kwargs.get("plan")
        if plan is None:
            plan = self.make_plan(video_mask, words_mask, num_clips, kwargs["dataset_name"], is_training,
                                  words_weight=words_weight, clip_mask=kwargs.get("clip_mask"),
                                  neg_index=kwargs.get("neg_index"),
                                  masked_words=kwargs.get("masked_words"), device=dev, n_valid=kwargs.get("_n_real"))

        # The step's independent chains run SIDE BY SIDE (ops.lockstep): every round, the blocks the chains are at
        # become one autograd node whose launch phases are shared -- the small problems of the SS-MESM / MLM stacks
        # ride in the launches of the 4800-row enhance stack instead of queueing behind them as ~5 us launches of
        # their own, forward and backward.  Stage A: the input projections of every modality.  Stage B: the enhance
        # stack beside the SS-MESM stack, the MLM stack beside the SS-MESM layers the enhance stack leaves alone.
        # Stage C: the t2v stack beside the MLM head and the (loss-free) sentence projection.
        mlm = self.rec_fw and is_training
        enc = self.enhance_encoder
        ni = plan.neg_index

        def proj_chain(seq, x):
            for m in seq:
                x = yield from m.steps(x)
            return x

        # batches padded to a captured pair capacity: the real pair count is a device scalar of the plan
        kn.set_mask_mod(getattr(plan, "n_valid", None))
        with _scope("inproj"):
            vid_pad = plan.vpad if getattr(plan, "vpad", None) is not None else (~video_mask).contiguous()
            words_pad = plan.wpad if getattr(plan, "wpad", None) is not None else (~words_mask).contiguous()
            vpos = kn.sine_pos(video_mask, d)
            chains = [proj_chain(self.input_vid_proj, video_feat), proj_chain(self.input_txt_proj, words)]
            if self.rec_ss:
                if plan.vid_src is not None and not plan.vid_identity:
                    bvid = video_feat.reshape(N * Lv, -1)[plan.vid_src] * plan.vid_mask.unsqueeze(-1)
                    bvid_pad = plan.vid_pad
                elif plan.vid_src is not None:
                    bvid, bvid_pad = video_feat, plan.vid_pad  # 27 MB gather + mask multiply skipped
                else:
                    bvid, bvid_pad = video_feat, vid_pad
                # the group's sentences per pair (zeros in the padding slots), model.py:199 split_expand_and_pad
                bsent = kn.gather_rows_fwd(sent, plan.sent_src, plan.sent_mask)[0]
                chains += [proj_chain(self.input_vid_proj, bvid), proj_chain(self.input_txt_proj, bsent)]
            if mlm:
                chains += [proj_chain(self.input_txt_proj, self.unknown_token.view(1, 1, -1)),
                           proj_chain(self.input_txt_proj, self.masked_token.view(1, 1, -1))]
            res = ops.lockstep(chains)
            pv, pw = res[0], res[1]
            if pv.requires_grad and not torch.cuda.is_current_stream_capturing():
                # the backward has reached the input side: the pool-backed activations of this forward are spent
                # (kernels.ZeroPool.fwd_live; under capture the step is one forward + one backward by construction)
                pv.register_hook(_pool_spent)
            if self.rec_ss:
                bvid, bsent = res[2], res[3]
            if mlm:
                unk, msk = res[-2], res[-1]
            tpos = self.txt_position_embed(pw) if self.use_txt_pos else None  # model.py:169-172
            defer_enhance = False

        # The positive and the negative pass (model.py:260-299) run the SAME weights over the same
        # video with different queries: they are stacked along the batch (rows [0, N) positive,
        # [N, 2N) negative) so every layer is one launch over 2N rows instead of two over N --
        # these kernels are far too small to fill 256 CUs, so rows are what buys efficiency.
        # The Q1 mask rule wraps inside each group of N rows (group=N).  The decoder runs on the
        # positive half only: the reference discards the negative decoder output (model.py:295).
        out = {}
        with _scope("enhance"):
            # one launch builds every stacked tensor of the stage: [x ; x] for the video side, [x ; x[neg_index]]
            # for the words (neg_words_feat = expanded_words_feat[neg_index][:, 1:] = projected words of the
            # negative query; the SS token is stripped again, model.py:264-266)
            # ... and the stage's other assembly steps -- the first query pv2 + position (formed without autograd: the
            # consumer block folds its gradient into d pv2, join_vid_p), the SS-MESM query tokens, the MLM branch's
            # token replacement and ground-truth clip gathers -- are independent of it and of each other: ONE autograd
            # node whose kernels leave in one grouped assembly launch, forward and backward (ops.glue_block)
            # (the projected video / words have three consumers each -- the stacked copy, the MLM branch, and the rec_ss
            # loss resp. the expanded words: aliases whose gradients meet in one launch, ops.fork)
            n_pv = 1 + (1 if mlm else 0) + (1 if self.rec_ss else 0)
            pv_s, pv_g, pv_o = ops.fork(pv, 3) if n_pv == 3 else (pv, pv, pv)
            pw_s, pw_t, pw_p = ops.fork(pw, 3) if n_pv == 3 else (pw, pw, pw)
            prep = [ops.stack_rows_call([pv_s, vpos, vid_pad, pw_s, words_pad], [0, 0, 0, 1, 1], ni),
                    ops.add_tile_call(pv.detach(), vpos, 2)]
            if self.rec_ss:
                # the pair's own sentence slot is replaced by the learned token (model.py:493-501)
                prep.append(ops.token_mix_call(bsent, plan.sent_loc, self.ss_reconstructor.masked_sent_token))
            if mlm:
                # FW-MESM masked-language-model branch (model.py:307-332): unknown words, then the drawn positions,
                # become learned tokens (model.py:361-394); the ground-truth clips of every pair re-padded to Lc, and
                # their position embeddings (model.py:312-325)
                prep += [ops.token_mix_call(pw_t, kwargs["unknown_mask"], unk, plan.masked_words, msk),
                         ops.gather_rows2_call(pv_g.reshape(N * Lv, d), plan.clip_src, plan.clip_inv, plan.clip_mask),
                         ops.gather_rows2_call(vpos.reshape(N * Lv, d), plan.clip_src, plan.clip_inv, plan.clip_mask),
                         # clips + their positions in one pass: the key-side input of the MLM blocks (a plain operand)
                         ops.gather_add_call(pv.detach().reshape(N * Lv, d), vpos.reshape(N * Lv, d), plan.clip_src,
                                             plan.clip_mask)]
            prep = ops.par(prep)
            (pv2, vpos2, vid_pad2, pw2, wpad2), pvp2 = prep[0], prep[1]
            if self.rec_ss:
                q_tok = prep[2]
            if mlm:
                w, cfeat, cpos, cfeat_p = prep[-4:]
            stage, names = [], []
            tpos2 = None
            if self.rec_fw:
                if self.use_txt_pos and not self.rec_ss:
                    tpos2 = ops.stack_rows([tpos], [1], ni)[0]  # txt_position ; txt_position[neg_index]
                elif self.use_txt_pos:
                    defer_enhance = True  # the negative half takes positions 1.. of the EXPANDED words (below)
                # every block hands its output + position embedding to the next one (second output of its last
                # LayerNorm), so only this very first query is formed by an element-wise launch
                if not defer_enhance:
                    stage.append(enc.steps(pw2, pv2, tpos2, vpos2, wpad2, vid_pad2, group=N, vid_p=pvp2, out_pos=vpos2,
                                           join_vid_p=True))
                    names.append("E")
            if self.rec_ss:
                stage.append(self.ss_reconstructor.recon_trans.steps(bvid, q_tok, None, None, bvid_pad, plan.sent_pad))
                names.append("S")
            if mlm:
                m_chain = enc.steps(cfeat, w, cpos, tpos, plan.clip_pad, words_pad, is_mlm=True, txt_p=cfeat_p)  # pos_vid = txt_position
                # beside the SS-MESM layers that outlast the enhance stack (3 rounds per layer)
                lag = 3 * len(enc.t2v_encoder.layers) if ("E" in names and "S" in names) else 0
                stage.append(ops.delayed(m_chain, lag))
                names.append("M")
            res = dict(zip(names, ops.lockstep(stage)))
            if "E" in res:
                enhanced2, enhanced2_p = res["E"]
                enhanced = enhanced2[:N]
            elif not self.rec_fw:
                enhanced2, enhanced2_p = pv2, pvp2
                enhanced = pv
            else:
                enhanced2 = enhanced2_p = enhanced = None  # deferred (use_txt_pos with the sentence token)

        with _scope("ss"):
            if self.rec_ss:
                rec = res["S"]
                # the masked slot of every pair, L2-normalised (model.py:485-486): one kernel
                recon = ops.gather_rows2(rec.reshape(-1, d), plan.recon_idx, plan.recon_inv, normalize=True)
                # [recon ; words] and its padding mask (model.py:221-224), one launch
                ewords, epad = ops.prepend(recon, pw_p, pad=words_pad, first_pad=False)
                emask = plan.emask if getattr(plan, "emask", None) is not None else ~epad
            else:
                ewords, emask = pw, words_mask
                epad = words_pad

        with _scope("t2v"):
            pre = None
            if self.use_txt_pos:
                # expanded_txt_position (model.py:225-226) and its negative gather (:263); with the sentence token in
                # front the negative words of the enhance stage keep positions 1.. of it (:267), so that stage
                # could not run before the SS branch
                etpos = self.txt_position_embed(ewords)
                ewords2, epad2, etpos2 = ops.stack_rows([ewords, epad, etpos], [1, 1, 1], ni)
                if defer_enhance:
                    tpos2 = torch.cat([tpos, etpos2[N:, 1:]], 0)
                    pre = enc.steps(pw2, pv2, tpos2, vpos2, wpad2, vid_pad2, group=N, vid_p=pvp2, out_pos=vpos2,
                                    join_vid_p=True)
            else:
                ewords2, epad2 = ops.stack_rows([ewords, epad], [1, 1], ni)
                etpos2 = None

            def t2v_chain():
                e2, e2p = enhanced2, enhanced2_p
                if pre is not None:
                    e2, e2p = yield from pre
                # (without the enhance stage the t2v stack's first query is the autograd-free pv2 + position)
                enc2 = yield from self.t2v_encoder.steps(ewords2, e2, etpos2, vpos2, epad2, vid_pad2, group=N, vid_p=e2p,
                                                         join_vid_p=e2p is pvp2)
                return e2, enc2

            def mlm_head_chain():
                hid = yield from self.output_txt_proj[0].steps(res["M"])
                head = self.output_txt_proj[1]
                return (yield ops.linear_call(hid, head.weight, head.bias))

            stage, names = [t2v_chain()], ["T"]
            if mlm:
                # one round late: the head's LayerNorm then shares the launch of the t2v layer's FFN LayerNorm and its
                # vocabulary product the launch of the second layer's projections (started with the stack, none of its three
                # launches had a partner of its kind, forward or backward)
                stage.append(ops.delayed(mlm_head_chain(), _MLM_HEAD_LAG))
                names.append("H")
            if self.rec_ss:
                # projed_recon_feat feeds no loss (criterion.py:246-255) but is part of the returned dict
                stage.append(proj_chain(self.ss_reconstructor.output_sent_proj, recon))
                names.append("R")
            res2 = dict(zip(names, ops.lockstep(stage)))
            e2_, encoded2 = res2["T"]
            if enhanced is None:
                enhanced = e2_[:N]
            if self.rec_ss:
                projed_recon = res2["R"]
        with _scope("transformer"):
            hs, refs, memory2, memory_g2 = self.transformer(
                encoded2, vid_pad2, self.query_embed.weight, vpos2, self.global_rep_token,
                self.global_rep_pos, n_dec=N, pack=self.pack)
        with _scope("heads"):
            # independent projections share one grouped launch: class head, first span-head layer,
            # and the two saliency projections (model.py:301-302, both passes in one go)
            def heads_chain():
                l0 = self.span_embed.layers[0]
                logits, sp, sa, sb = yield [
                    ops.linear_call(hs, self.class_embed.weight, self.class_embed.bias),
                    ops.linear_call(hs, l0.weight, l0.bias, relu=True),
                    ops.linear_call(memory2, self.saliency_proj1.weight, self.saliency_proj1.bias),
                    ops.linear_call(memory_g2, self.saliency_proj2.weight, self.saliency_proj2.bias)]
                for i_, l_ in enumerate(self.span_embed.layers[1:], 1):
                    sp = yield ops.linear_call(sp, l_.weight, l_.bias, relu=i_ < len(self.span_embed.layers) - 1)
                return logits, sp, sa, sb

            logits, sp, sa, sb = ops.seq(heads_chain())
            spans = ops.ref_update(sp, refs)  # sigmoid(span_embed(hs) + inverse_sigmoid(refs)), model.py:250
            sal2 = ops.rowdot(sa, sb, 1.0 / float(np.sqrt(d)))

        def layer_of(stack, k):
            # the reference's per-layer tensors, tagged with the stacked tensor they are a view of: the criterion
            # block then reads the stack itself and returns ONE gradient for it (criterion.py: _mesm_stack)
            v = stack[k]
            v._mesm_stack = (stack, k)
            return v

        nl = logits.shape[0]
        sal_p, sal_n = sal2[:N], sal2[N:]
        sal_p._mesm_stack, sal_n._mesm_stack = (sal2, 0), (sal2, 1)
        out.update({"pred_logits": layer_of(logits, nl - 1), "pred_spans": layer_of(spans, nl - 1),
                    "saliency_scores": sal_p, "neg_saliency_scores": sal_n})
        if self.aux_loss:
            out["aux_outputs"] = [{"pred_logits": layer_of(logits, k), "pred_spans": layer_of(spans, k)}
                                  for k in range(nl - 1)]

        if mlm:
            out["recfw_words_logit"] = res2["H"]
            out["words_mask"] = words_mask
        if self.rec_ss:
            out.update({"projed_video_feat": pv_o, "recon_feat": recon, "projed_recon_feat": projed_recon,
                        "expanded_words_feat": ewords, "expanded_words_mask": emask,
                        "enhanced_video_feat": enhanced, "projed_words_feat": pw})
        if not auto.busy:  # (the criterion finds the model on its outputs: autograph.py)
            out = AutoOutputs(out)
            out._mesm_model = weakref.ref(self)
        return out
