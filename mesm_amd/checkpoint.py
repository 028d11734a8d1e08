"""Checkpoint interop with the reference.

    train.py:185-223   save format: {"model": state_dict minus text_encoder.*, "optimizer": AdamW.state_dict(),
                       "lr_scheduler": StepLR.state_dict(), "epoch": int, "opt": Namespace}
    train.py:117-125   resume: model.load_state_dict(ckpt["model"]); with --resume_all also optimizer, scheduler,
                       start_epoch = epoch + 1
    eval.py:513-521    inference: the frozen text encoder's own weights are merged back before the strict load
    utils/model_utils.py:20-36   state_dict_without_module / merge_state_dict_with_module

The parameter names of mesm_amd.MESM are the reference's, so a released
`model_*_best.ckpt` loads strictly; the optimizer state is torch.optim.AdamW's per-parameter layout in
`named_parameters()` order of the trainable tensors, which FlatAdamW reads into / writes from its flat buffers.
"""
from collections import OrderedDict

import torch


def state_dict_without_module(model, module_name):
    return OrderedDict((k, v) for k, v in model.state_dict().items() if module_name not in k)


def merge_state_dict_with_module(state_dict, module_state_dict, module_name):
    state_dict.update(OrderedDict((module_name + "." + k, v) for k, v in module_state_dict.items()))
    return state_dict


def save_checkpoint(path, model, optimizer, lr_scheduler, epoch, opt):
    """train.py:185-192 / 207-214."""
    ckpt = {"model": state_dict_without_module(model, "text_encoder"), "optimizer": optimizer.state_dict(),
            "epoch": epoch, "opt": opt}
    if lr_scheduler is not None:
        ckpt["lr_scheduler"] = lr_scheduler.state_dict()
    torch.save(ckpt, path)
    return ckpt


def load_checkpoint(path_or_dict, model, optimizer=None, lr_scheduler=None, resume_all=False):
    """Load a reference-format checkpoint into a mesm_amd model (and, with resume_all, the optimizer and
    scheduler).  Returns the epoch to continue from (train.py:124: epoch + 1 when resume_all, else None).
    The checkpoint stores the argparse Namespace `opt`, hence weights_only=False (as torch 1.11 loaded it)."""
    ckpt = path_or_dict if isinstance(path_or_dict, dict) else torch.load(path_or_dict, map_location="cpu",
                                                                           weights_only=False)
    sd = OrderedDict(ckpt["model"])
    if getattr(model, "text_encoder", None) is not None:  # eval.py:515-518
        sd = merge_state_dict_with_module(sd, model.text_encoder.state_dict(), "text_encoder")
    model.load_state_dict(sd)
    start = None
    if resume_all:
        if optimizer is not None:
            optimizer.load_state_dict(ckpt["optimizer"])
        if lr_scheduler is not None and "lr_scheduler" in ckpt:
            lr_scheduler.load_state_dict(ckpt["lr_scheduler"])
        start = ckpt["epoch"] + 1
    return start
