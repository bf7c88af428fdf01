"""The attention block against the reference's OWN source: tests/golden/attention.npz holds what the reference's vendored
`src/diffusers/models/attention_processor.py` (`Attention` :128-297 + `AttnProcessor2_0` :1256-1341, `my_get_processor`
:37-125) computed on the seeded cases of tests/attention_cases.py (generator: tests/golden/make_attention_golden.py).

CPU: the oracle's attention restatements (oracle/diffusers_ref.py::Attention, oracle/sd_unet_ref.py::CrossAttention) in
fp64 against the fixture - outputs and input gradients to 1e-6 relative (the fixture is an fp64 result stored in fp32),
parameter gradients (norm and a random projection) to 1e-9 relative; the LoRA key grammar this build writes against the
processor `my_get_processor` builds for ragged ranks.
GPU: the HIP attention block (gad/nn.py::Attention, gad/sd.py::CrossAttention through the C ABI) against the same
fixture - outputs 2e-5, input gradients 1e-4 of the gradient's scale, parameter-gradient norms 2e-4 relative (fp32
products; the reference's own fp32 run is 0.4-2.8e-6 away from its fp64 run, stored as `out32_maxdiff`)."""
import os

import numpy as np
import pytest
import torch

import attention_cases as AC

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "attention.npz"))


def test_case_tensors_reproduce_the_checksums_the_fixture_was_made_with():
    for case in AC.CASES:
        for k, v in AC.checksums(case).items():
            assert abs(float(GOLD[f"{case}/sum/{k}"]) - v) <= 1e-9 * max(1.0, abs(v)), (case, k)


def _oracle_module(case):
    from oracle.diffusers_ref import Attention, LoRALinearLayer
    from oracle.sd_unet_ref import CrossAttention
    c = AC.CASES[case]
    if len(c["shape"]) == 4:
        m = Attention(c["query_dim"], c["heads"], c["dim_head"], c["eps"], c["groups"],
                      rescale_output_factor=c.get("rescale", 1.0), residual_connection=c["residual"], bias=c["bias"])
    else:
        m = CrossAttention(c["query_dim"], c.get("cross_dim"), c["heads"], c["dim_head"])
    for proj, lin in AC.projections(m):
        if "ranks" in c:
            lin.set_lora_layer(LoRALinearLayer(lin.in_features, lin.out_features, c["ranks"][proj], c.get("network_alpha")))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in AC.weights(case).items()})
    return m.double()


def _check_param_grads(case, named_grads, rel, floor):
    """`floor`: the norm below which a gradient is rounding noise on both sides (`to_k.bias`: softmax is invariant to a
    per-query shift of the scores, so its gradient is analytically zero)."""
    for k, g in named_grads.items():
        want_norm, want_dot = GOLD[f"{case}/grad/{k}"]
        g = g.detach().double().cpu()
        dot = float((g * torch.from_numpy(AC.probe(case, k, tuple(g.shape)))).sum())
        assert abs(float(g.norm()) - want_norm) <= rel * max(want_norm, floor), (case, k, float(g.norm()), want_norm)
        assert abs(dot - want_dot) <= rel * max(want_norm, floor) * np.sqrt(g.numel()), (case, k, dot, want_dot)


@pytest.mark.parametrize("case", list(AC.CASES))
def test_oracle_attention_matches_the_reference_source(case):
    m = _oracle_module(case)
    x, ctx, g = (None if a is None else torch.from_numpy(a).double() for a in AC.inputs(case))
    x.requires_grad_(True)
    scale = AC.CASES[case].get("scale", 1.0)
    out = m(x, scale=scale) if ctx is None and x.dim() == 4 else m(x, ctx, scale)
    (out * g).sum().backward()
    want, want_dx = GOLD[f"{case}/out"], GOLD[f"{case}/dx"]
    assert np.abs(out.detach().numpy() - want).max() <= 1e-6 * np.abs(want).max()
    assert np.abs(x.grad.numpy() - want_dx).max() <= 1e-6 * np.abs(want_dx).max()
    _check_param_grads(case, {k: p.grad for k, p in m.named_parameters()}, 1e-9, 1e-3)


@pytest.mark.parametrize("case", [c for c in AC.CASES if "ranks" in AC.CASES[c]])
def test_lora_key_grammar_matches_my_get_processor(case):
    """`unet.attn_processors` -> `save_attn_procs` serialises, per attention, the processor `my_get_processor` returns:
    `to_{q,k,v,out}_lora.{down,up}.weight` with PER-PROJECTION ranks (the fix of src/utils.py:84-96).  The product's
    `lora_state_dict` must write those keys, shapes and values for the same attention."""
    assert str(GOLD[f"{case}/proc/class"]) == "LoRAAttnProcessor2_0"
    w = AC.weights(case)
    keys = [str(k) for k in GOLD[f"{case}/proc/keys"]]
    assert keys == sorted(f"{p}_lora.{h}.weight" for p in ("to_q", "to_k", "to_v", "to_out") for h in ("down", "up"))
    for k in keys:
        proj, half, _ = k.split(".")
        name = {"to_out_lora": "to_out.0"}.get(proj, proj[: -len("_lora")])
        mine = w[f"{name}.lora_layer.{half}.weight"]
        *shape, total = GOLD[f"{case}/proc/{k}"]
        assert tuple(int(s) for s in shape) == mine.shape, (k, shape, mine.shape)
        assert abs(total - float(mine.astype(np.float64).sum())) < 1e-9
    # and the writer of this build (host code, no kernels involved) emits exactly that grammar
    from gad.sd import CrossAttention, UNet2DConditionModel
    from gad.nn import LoRALinearLayer
    c = AC.CASES[case]
    holder = torch.nn.Module()
    holder.attn1 = CrossAttention(c["query_dim"], c.get("cross_dim"), c["heads"], c["dim_head"])
    for proj, lin in AC.projections(holder.attn1):
        lin.set_lora_layer(LoRALinearLayer(lin.weight.shape[1], lin.weight.shape[0], c["ranks"][proj], c.get("network_alpha")))
    missing, unexpected = holder.attn1.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=False)
    assert not missing and not unexpected
    holder.attention_modules = lambda: UNet2DConditionModel.attention_modules(holder)
    sd = UNet2DConditionModel.lora_state_dict(holder)
    assert sorted(sd) == sorted("attn1.processor." + k for k in keys)
    for k in keys:
        *shape, total = GOLD[f"{case}/proc/{k}"]
        t = sd["attn1.processor." + k]
        assert tuple(t.shape) == tuple(int(s) for s in shape) and abs(float(t.double().sum()) - total) < 1e-9


# --------------------------------------------------------------------------------------------------------------------
# GPU: the HIP block against the reference-made fixture
# --------------------------------------------------------------------------------------------------------------------
GPU_CASES = [c for c in AC.CASES if "rescale" not in AC.CASES[c]]       # every model config has rescale_output_factor 1


def _product_module(case, dev):
    from gad.nn import Attention, LoRALinearLayer
    from gad.sd import CrossAttention
    c = AC.CASES[case]
    if len(c["shape"]) == 4:
        m = Attention(c["query_dim"], c["heads"], c["dim_head"], c["eps"], c["groups"])
    else:
        m = CrossAttention(c["query_dim"], c.get("cross_dim"), c["heads"], c["dim_head"])
    for proj, lin in AC.projections(m):
        if "ranks" in c:
            lin.set_lora_layer(LoRALinearLayer(lin.weight.shape[1], lin.weight.shape[0], c["ranks"][proj], c.get("network_alpha")))
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in AC.weights(case).items()}, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return m.to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("case", GPU_CASES)
def test_hip_attention_block_matches_the_reference_source(case):
    dev = torch.device("cuda:0")
    c = AC.CASES[case]
    m = _product_module(case, dev)
    x, ctx, g = (None if a is None else torch.from_numpy(a).to(dev) for a in AC.inputs(case))
    four_d = x.dim() == 4
    if four_d:                                     # the product's activations are NHWC
        x, g = x.permute(0, 2, 3, 1).contiguous(), g.permute(0, 2, 3, 1).contiguous()
    x.requires_grad_(True)
    out = m(x, scale=c.get("scale", 1.0)) if four_d else m(x, ctx, scale=c.get("scale", 1.0))
    (out * g).sum().backward()
    dx = x.grad
    if four_d:
        out, dx = out.permute(0, 3, 1, 2), dx.permute(0, 3, 1, 2)
    want, want_dx = GOLD[f"{case}/out"], GOLD[f"{case}/dx"]
    err = np.abs(out.detach().cpu().numpy() - want).max()
    assert err <= 2e-5, (case, err, float(GOLD[f"{case}/out32_maxdiff"]))
    err = np.abs(dx.cpu().numpy() - want_dx).max()
    assert err <= 1e-4 * np.abs(want_dx).max(), (case, err)
    _check_param_grads(case, {k: p.grad for k, p in m.named_parameters() if p.grad is not None}, 2e-4, 1.0)
    if four_d:                                     # sampling route: fused q|k|v projection, q/k/v read in place
        with torch.no_grad():
            o2 = m(x.detach()).permute(0, 3, 1, 2)
        assert np.abs(o2.cpu().numpy() - want).max() <= 2e-5
