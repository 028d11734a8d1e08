"""Factory functions with the reference's signatures (runner.py:190-352): the drop-in boundary.

    model = build_model(args)            # nn.Module, same forward signature / state_dict keys
    criterion = build_criterion(args)    # nn.Module -> (loss_dict, total), .weight_dict
    optimizer, sched = build_optimizer(args, model)

`args` is the reference's option namespace (utils/config.py); only the fields the reference
factories read are used.
"""
import logging

import torch

from .criterion import Criterion, HungarianMatcher
from .layers import T2VEncoder, Transformer
from .model import MESM, TrainablePositionalEncoding

logger = logging.getLogger(__name__)


def build_enhance_encoder(args):
    """runner.py:190-210: T2VEncoder, or the TwoMLP variant when share_MLP is false."""
    return T2VEncoder(args.hidden_dim, args.nheads, args.num_recfw_layers, args.dim_feedforward,
                      args.dropout, two_mlp=not args.share_MLP)


def build_t2v_encoder(args):
    """runner.py:213-222."""
    return T2VEncoder(args.hidden_dim, args.nheads, args.t2v_layers, args.dim_feedforward, args.dropout)


def build_transformer(args):
    """runner.py:225-236 (activation 'prelu', return_intermediate_dec=True)."""
    if args.pre_norm:
        raise NotImplementedError("self.normalize_before is True")
    t = Transformer(args.hidden_dim, args.nheads, args.num_queries, args.enc_layers, args.dec_layers,
                    args.dim_feedforward, args.dropout)
    torch.nn.init.zeros_(t.decoder.bbox_embed.layers[-1].bias)
    return t


def build_position_encoding(args):
    """runner.py:239-252.  The sine encoding is a parameter-free kernel (mesm_sine_pos_fwd)."""
    if args.position_embedding not in ("v2", "sine"):
        raise ValueError(f"not supported {args.position_embedding}")
    txt_pos = TrainablePositionalEncoding(args.max_words_l + 1 if args.rec_ss else args.max_words_l,
                                          args.hidden_dim, args.input_dropout)
    return None, txt_pos


def build_model(args, vocab=None):
    """runner.py:255-298."""
    logger.info("Building model...")
    from .text_encoder import build_CLIP_text_encoder, build_GloVe_text_encoder
    if args.tokenizer_type == "GloVeSimple":
        text_encoder = build_GloVe_text_encoder(args.text_model_path, vocab)
    elif args.tokenizer_type == "CLIP":
        text_encoder = build_CLIP_text_encoder(args.text_model_path)
    elif args.tokenizer_type == "GloVeNLTK":
        text_encoder = None if args.load_vocab_pkl else build_GloVe_text_encoder(args.text_model_path, vocab)
    else:
        raise NotImplementedError
    vid_pos, txt_pos = build_position_encoding(args)
    model = MESM(text_encoder=text_encoder, t2v_encoder=build_t2v_encoder(args),
                 enhance_encoder=build_enhance_encoder(args), transformer=build_transformer(args),
                 vid_position_embed=vid_pos, txt_position_embed=txt_pos, txt_dim=args.t_feat_dim,
                 vid_dim=args.v_feat_dim, num_queries=args.num_queries, input_dropout=args.input_dropout,
                 aux_loss=args.aux_loss, max_video_l=args.max_video_l, max_words_l=args.max_words_l,
                 normalize_txt=args.normalize_txt, use_txt_pos=args.use_txt_pos,
                 span_loss_type=args.span_loss_type, n_input_proj=args.n_input_proj, rec_fw=args.rec_fw,
                 vocab_size=args.vocab_size, rec_ss=args.rec_ss, num_recss_layers=args.num_recss_layers,
                 share_MLP=args.share_MLP)
    model.to(args.device)
    if torch.device(args.device).type == "cuda":
        model.flat_params()  # final parameter addresses before any optimizer / HIP-graph capture
    return model


def build_matcher(args):
    """runner.py:301-306."""
    return HungarianMatcher(cost_span=args.set_cost_span, cost_giou=args.set_cost_giou,
                            cost_class=args.set_cost_class, span_loss_type=args.span_loss_type,
                            max_v_l=args.max_video_l, multi_clip=args.dataset_name in ["qvhighlights"])


def build_criterion(args):
    """runner.py:309-345."""
    logger.info("Building criterion...")
    matcher = build_matcher(args)
    losses = ["span", "label", "saliency"]
    weight_dict = {"loss_span": args.loss_span_coef, "loss_giou": args.loss_giou_coef,
                   "loss_label": args.loss_label_coef, "loss_saliency": args.loss_saliency_coef}
    if args.aux_loss:
        aux = {}
        for i in range(args.dec_layers - 1):
            aux.update({k + f"_{i}": v for k, v in weight_dict.items() if k != "loss_saliency"})
        weight_dict.update(aux)
    if args.rec_fw:
        losses.append("rec_fw")
        weight_dict["loss_rec_fw"] = args.loss_recfw_coef
    if args.rec_ss:
        losses.append("rec_ss")
        weight_dict["loss_rec_ss"] = args.loss_recss_coef
    criterion = Criterion(matcher=matcher, weight_dict=weight_dict, losses=losses, eos_coef=args.eos_coef,
                          span_loss_type=args.span_loss_type, max_video_l=args.max_video_l,
                          rank_coef=args.rank_coef, use_triplet=args.use_triplet,
                          saliency_margin=args.saliency_margin,
                          multi_clip=args.dataset_name in ["qvhighlights"], gamma=args.iou_gamma,
                          recss_tau=args.recss_tau)
    criterion.to(args.device)
    return criterion


def build_optimizer(opt, model):
    """runner.py:348-352: AdamW(lr, weight_decay) + StepLR(lr_drop, gamma), here on the flat parameter /
    gradient buffers (mesm_amd/optim.py): `optimizer.step()` is one launch for all tensors, and
    `optimizer.step(grad_clip=opt.grad_clip)` also absorbs train.py:70-71's clip_grad_norm_."""
    from .optim import build_optimizer as _build
    return _build(opt, model)
