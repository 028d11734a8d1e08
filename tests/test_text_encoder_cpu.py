"""The CPU restatement of the frozen text encoders (oracle/clip_text_oracle.py) against the outputs of the
real reference (tests/golden/clip_text_tiny.npz, tools/gen_golden_r2.py): CLIPTextEncoder.forward in fp16
and MESM.CLIP_encode_text.  fp16 tolerance: 1e-3 of the tensor scale (one fp16 ulp is 4.9e-4 relative)."""
import os

import numpy as np
import torch

from golden_io import GOLDEN
from oracle import clip_text_oracle as C

TOL16 = 1e-3


def load():
    z = np.load(os.path.join(GOLDEN, "clip_text_tiny.npz"))
    sd = {k[3:]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("sd.")}
    t = lambda k: torch.from_numpy(z[k].copy())
    return sd, z, t


def rel(a, b):
    return float((a.double() - b.double()).abs().max()) / max(float(b.double().abs().max()), 1e-3)


def test_state_dict_has_the_reference_dtypes():
    sd, _, _ = load()
    assert sd["transformer.resblocks.0.attn.in_proj_weight"].dtype == torch.float16
    assert sd["transformer.resblocks.0.mlp.c_fc.bias"].dtype == torch.float16
    assert sd["text_projection"].dtype == torch.float16
    assert sd["token_embedding.weight"].dtype == torch.float32
    assert sd["transformer.resblocks.0.ln_1.weight"].dtype == torch.float32


def test_clip_text_forward_matches_reference():
    sd, z, t = load()
    hid = C.clip_text_forward(sd, t("ids"))
    assert hid.dtype == torch.float16 and hid.shape == t("hidden").shape
    assert rel(hid, t("hidden")) < TOL16


def test_clip_encode_text_matches_reference():
    sd, z, t = load()
    wf, sf, wid, wm = C.clip_encode_text(sd, t("ids"), t("mask"), int(z["max_words_l"]))
    assert torch.equal(wid, t("words_id_cut")) and torch.equal(wm, t("words_mask_cut"))
    assert rel(wf, t("words_feat")) < TOL16 and rel(sf, t("sentence_feat")) < TOL16
    # pads are exact zeros, valid words unit-norm
    assert float(wf[~wm].abs().max()) == 0.0
    assert torch.allclose(wf[wm].norm(dim=-1), torch.ones(int(wm.sum())), atol=1e-5)
