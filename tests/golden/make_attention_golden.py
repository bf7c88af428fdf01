"""Pin the attention block to the ONE hot-path source file the reference holds: its vendored
`src/diffusers/models/attention_processor.py` (`Attention` :128-297, `AttnProcessor2_0` :1256-1341,
`my_get_processor` :37-125).  The file is imported UNCHANGED from /root/reference and executed; what it computes is
stored as tests/golden/attention.npz.  Build container only (the reference does not travel):

    python tests/golden/make_attention_golden.py

Placeholders.  The file's four top-level imports (:21-24) name the absent `diffusers==0.24.0` package; they get
placeholder modules: `USE_PEFT_BACKEND = False`, a no-op `deprecate`, `logging.get_logger`, `is_xformers_available()
-> False`, an identity `maybe_allow_in_graph`.  `diffusers.models.lora` is a STAND-IN written here from the published
semantics (not reference code): `LoRACompatibleLinear` = `nn.Linear` with a `lora_layer` attribute and
`forward(x, scale) = linear(x) + scale * lora_layer(x)`; `LoRALinearLayer(in, out, rank, network_alpha)` =
`up(down(x)) * (network_alpha / rank if network_alpha else 1)`.  So the fixture pins the attention arithmetic
(GroupNorm placement, head split, SDPA, out projection, residual, rescale) and `my_get_processor`'s per-projection
ranks to the reference's source; the LoRA linear itself stays "parity unpinned" (restated).

Inputs and weights are NOT stored: both sides rebuild them from `numpy.random.RandomState(seed)` through
`tests/attention_cases.py` (a float64 checksum of every tensor is stored to catch generator drift).  Stored per case:
the module output computed in fp64, stored rounded to fp32 for size (and how far the reference's own fp32 run is from it), and from the fp64 run the input
gradient of <out, g> (rounded likewise) plus (norm, <grad, r>) per parameter.
"""
import importlib
import logging as _pylogging
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "tests"))
import attention_cases as AC  # noqa: E402


class LoRALinearLayer(nn.Module):                       # stand-in, see the header
    def __init__(self, in_features, out_features, rank=4, network_alpha=None, device=None, dtype=None):
        super().__init__()
        self.down = nn.Linear(in_features, rank, bias=False)
        self.up = nn.Linear(rank, out_features, bias=False)
        self.network_alpha, self.rank = network_alpha, rank
        self.in_features, self.out_features = in_features, out_features

    def forward(self, x):
        y = self.up(self.down(x))
        return y * (self.network_alpha / self.rank) if self.network_alpha is not None else y


class LoRACompatibleLinear(nn.Linear):                  # stand-in, see the header
    def __init__(self, *a, lora_layer=None, **k):
        super().__init__(*a, **k)
        self.lora_layer = lora_layer

    def set_lora_layer(self, lora_layer):
        self.lora_layer = lora_layer

    def forward(self, x, scale: float = 1.0):
        y = super().forward(x)
        return y if self.lora_layer is None else y + scale * self.lora_layer(x)


def import_reference_attention():
    def placeholder(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    d = placeholder("diffusers")
    d.utils = placeholder("diffusers.utils", USE_PEFT_BACKEND=False, deprecate=lambda *a, **k: None,
                          logging=types.SimpleNamespace(get_logger=_pylogging.getLogger))
    placeholder("diffusers.utils.import_utils", is_xformers_available=lambda: False)
    placeholder("diffusers.utils.torch_utils", maybe_allow_in_graph=lambda cls: cls)
    d.models = placeholder("diffusers.models")
    placeholder("diffusers.models.lora", LoRACompatibleLinear=LoRACompatibleLinear, LoRALinearLayer=LoRALinearLayer)
    sys.path.insert(0, REF)
    return importlib.import_module("src.diffusers.models.attention_processor")      # the reference file, unmodified


def build(ref, case, dtype):
    """The reference `Attention` for one case with the shared seeded weights."""
    c = AC.CASES[case]
    attn = ref.Attention(query_dim=c["query_dim"], cross_attention_dim=c.get("cross_dim"), heads=c["heads"],
                         dim_head=c["dim_head"], bias=c["bias"], norm_num_groups=c.get("groups"), eps=c.get("eps", 1e-5),
                         residual_connection=c["residual"], rescale_output_factor=c.get("rescale", 1.0),
                         _from_deprecated_attn_block=c.get("deprecated_block", False))
    assert isinstance(attn.processor, ref.AttnProcessor2_0)
    w = AC.weights(case)
    if "ranks" in c:
        for proj, lin in AC.projections(attn):
            lin.set_lora_layer(LoRALinearLayer(lin.in_features, lin.out_features, c["ranks"][proj], c.get("network_alpha")))
    sd = attn.state_dict()
    assert set(sd) == set(w), (sorted(sd), sorted(w))
    attn.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return attn.to(dtype)


def run(attn, case, dtype):
    x, ctx, g = (None if a is None else torch.from_numpy(a).to(dtype) for a in AC.inputs(case))
    x.requires_grad_(True)
    out = attn(x, encoder_hidden_states=ctx, scale=AC.CASES[case].get("scale", 1.0))
    (out * g).sum().backward()
    return out.detach(), x.grad.detach(), {k: p.grad.detach() for k, p in attn.named_parameters()}


def main():
    ref = import_reference_attention()
    store = {}
    for case in AC.CASES:
        for k, v in AC.checksums(case).items():
            store[f"{case}/sum/{k}"] = np.float64(v)
        out32, _, _ = run(build(ref, case, torch.float32), case, torch.float32)
        a64 = build(ref, case, torch.float64)
        out64, dx64, grads = run(a64, case, torch.float64)
        store[f"{case}/out"] = out64.numpy().astype(np.float32)                              # fp64 result, stored rounded
        store[f"{case}/out32_maxdiff"] = np.float64((out32.double() - out64).abs().max())   # the reference's own fp32 noise
        store[f"{case}/dx"] = dx64.numpy().astype(np.float32)                               # fp64 result, stored rounded
        for k, gr in grads.items():
            r = AC.probe(case, k, tuple(gr.shape))
            store[f"{case}/grad/{k}"] = np.array([gr.norm().item(), (gr * torch.from_numpy(r)).sum().item()])
        print(f"{case:22s} out {tuple(out64.shape)}  |out| {out64.norm():.6f}  fp32-fp64 {float((out32.double() - out64).abs().max()):.2e}")
        if "ranks" in AC.CASES[case]:
            # my_get_processor(:37-125) with LoRA active: the per-projection ranks must survive into the processor the
            # serializer (`unet.attn_processors` -> save_attn_procs) sees.  Stored: class name, state_dict keys + shapes + sums.
            proc = ref.my_get_processor(a64, return_deprecated_lora=True)
            psd = proc.state_dict()
            store[f"{case}/proc/class"] = np.array(type(proc).__name__)
            store[f"{case}/proc/keys"] = np.array(sorted(psd))
            for k, v in psd.items():
                store[f"{case}/proc/{k}"] = np.array(list(v.shape) + [float(v.double().sum())], dtype=np.float64)
            assert ref.my_get_processor(a64) is a64.processor                       # :48-49
    np.savez_compressed(os.path.join(HERE, "attention.npz"), **store)
    print("wrote", os.path.join(HERE, "attention.npz"), os.path.getsize(os.path.join(HERE, "attention.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
