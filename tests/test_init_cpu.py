"""A21: the initialiser distributions of build_model against what the reference's constructors leave behind
(transformer.py:78-81, 168-171, 306-331; model.py:19-101): xavier-uniform on every matrix of the three transformer
stacks (bounds sqrt(6 / (fan_in + fan_out))), nn.Linear defaults outside them, the bbox head's last bias zero and its
weight NOT zero (the xavier pass runs after the zeroing), PReLU slopes 0.25, learned tokens zero, N(0, 1) global
token / query embedding."""
import math

import torch

from mesm_amd import build_model, synthetic


def _model():
    torch.manual_seed(3)
    return build_model(synthetic.make_args("C3a", device="cpu"))


def test_transformer_matrices_are_xavier_uniform():
    m = _model()
    seen = 0
    for name, p in m.named_parameters():
        if p.dim() < 2 or not name.startswith(("enhance_encoder.", "t2v_encoder.", "transformer.")):
            continue
        fan_out, fan_in = p.shape[0], p.shape[1]
        bound = math.sqrt(6.0 / (fan_in + fan_out))
        assert float(p.abs().max()) <= bound * (1 + 1e-6), name
        if p.numel() >= 4096:  # uniform(-b, b): std = b / sqrt(3), mean 0
            assert abs(float(p.std()) / (bound / math.sqrt(3)) - 1) < 0.05, name
            assert abs(float(p.mean())) < 0.05 * bound, name
            assert float(p.abs().max()) > 0.9 * bound, name
        seen += 1
    assert seen > 60


def test_decoder_heads_and_scalars():
    m = _model()
    dec = m.transformer.decoder
    last = dec.bbox_embed.layers[-1]
    assert float(last.bias.abs().max()) == 0.0            # transformer.py:320-321
    assert float(last.weight.abs().max()) > 0.0           # ... but the later xavier pass re-fills the weight
    for name, p in m.named_parameters():
        if name.endswith("activation.weight"):
            assert p.shape == (1,) and float(p) == 0.25, name
    assert float(m.masked_token.abs().max()) == 0.0 and float(m.unknown_token.abs().max()) == 0.0
    assert float(m.ss_reconstructor.masked_sent_token.abs().max()) == 0.0
    assert 0.7 < float(m.global_rep_token.std()) < 1.3 and 0.7 < float(m.global_rep_pos.std()) < 1.3
    assert m.query_embed.weight.shape == (10, 2)
    # decoder layers > 0 have no ca_qpos_proj (keep_query_pos=False, transformer.py:329-331)
    assert dec.layers[0].ca_qpos_proj is not None and all(l.ca_qpos_proj is None for l in dec.layers[1:])


def test_linear_defaults_outside_the_stacks():
    m = _model()
    for name in ("input_vid_proj.0.net.1", "saliency_proj1", "class_embed", "span_embed.layers.0"):
        mod = m.get_submodule(name)
        fan_in = mod.weight.shape[1]
        bound = 1.0 / math.sqrt(fan_in)  # kaiming_uniform(a = sqrt(5)) and the bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
        assert float(mod.weight.abs().max()) <= bound * (1 + 1e-6), name
        assert float(mod.bias.abs().max()) <= bound * (1 + 1e-6), name
    ln = m.input_vid_proj[0].LayerNorm
    assert float((ln.weight - 1).abs().max()) == 0.0 and float(ln.bias.abs().max()) == 0.0


def test_stacked_outputs_in_place_host_logic():
    """ops.Slot / ops.stacked (the decoder's stacked outputs written in place): plain torch on the host -- the buffer comes back as the
    stack, the gradient reaches every producer as its slice, a tensor that does not live in its slot is refused."""
    import pytest
    import torch
    from mesm_amd import ops
    buf = torch.zeros(3, 4, 5)
    xs = [torch.randn(4, 5, requires_grad=True) for _ in range(3)]

    class Into(torch.autograd.Function):  # a producer that writes its result into a slot (what the kernels do)
        @staticmethod
        def forward(ctx, x, slot):
            slot.t.data.copy_(2.0 * x)  # (like a kernel: through the storage, not through torch's in-place machinery)
            return slot.t.view_as(slot.t)

        @staticmethod
        def backward(ctx, g):
            return 2.0 * g, None

    ys = [Into.apply(x, ops.Slot(buf, k)) for k, x in enumerate(xs)]
    st = ops.stacked(buf, ys)
    assert st.data_ptr() == buf.data_ptr() and torch.equal(st, torch.stack([2.0 * x.detach() for x in xs]))
    g = torch.randn(3, 4, 5)
    st.backward(g)
    for k, x in enumerate(xs):
        assert torch.equal(x.grad, 2.0 * g[k])
    with pytest.raises(AssertionError):
        ops.stacked(torch.zeros(3, 4, 5), [x.detach() for x in xs])
