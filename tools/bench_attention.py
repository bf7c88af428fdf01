"""Fused attention kernels timed at the reference's shapes (routes: fused = the product's choice - for d >= 160 the 8-wave
split-head-dim forward, for d <= 96 the single-pass backward; 2-kernel-bwd = 4-wave forward + the recomputing dQ + dK/dV pair;
narrow-fwd = the 4-wave forward at d >= 160; 3-launch = batched GEMMs + softmax with S in HBM) (HIP events, median of interleaved rounds) next to the
three-launch route (batched Q K^T -> softmax -> P V with S in HBM): TF/s against the 157.3 TF/s f32-MFMA roofline.
usage: python tools/bench_attention.py [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
import torch
from gad import ops
dev = torch.device("cuda:0")
ROUNDS = int(sys.argv[1]) if len(sys.argv) > 1 else 5
PREC = sys.argv[2] if len(sys.argv) > 2 else "f32"
ops.set_operand_precision(PREC)
print("operand precision", PREC, flush=True)
SHAPES = [  # name, B, Tq, Tk, heads, d
    ("sd512 self 64x64", 16, 4096, 4096, 8, 40), ("sd512 cross 64x64", 16, 4096, 77, 8, 40),
    ("sd512 self 32x32", 16, 1024, 1024, 8, 80), ("sd512 self 16x16", 16, 256, 256, 8, 160),
    ("sd256 self 32x32", 64, 1024, 1024, 8, 40), ("sd256 cross 32x32", 64, 1024, 77, 8, 40),
    ("sd256 self 16x16", 64, 256, 256, 8, 80), ("sd256 self 8x8", 64, 64, 64, 8, 160),
    ("celeba 32x32", 32, 1024, 1024, 14, 32), ("celeba 16x16", 32, 256, 256, 21, 32), ("celeba-pruned 32x32", 32, 1024, 1024, 14, 23),
    ("celeba-pruned 16x16", 32, 256, 256, 21, 23), ("cifar train 16x16", 128, 256, 256, 1, 256),
    ("cifar sample 16x16", 1024, 256, 256, 1, 256), ("cifar-pruned sample", 1024, 256, 256, 1, 192),
]


def ev_time(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)


for name, B, Tq, Tk, h, d in SHAPES:
    C = h * d
    g = torch.Generator(device=dev).manual_seed(0)
    q = torch.randn(B, Tq, C, device=dev, generator=g) * 0.5
    k = torch.randn(B, Tk, C, device=dev, generator=g) * 0.5
    v = torch.randn(B, Tk, C, device=dev, generator=g)
    do = torch.randn(B, Tq, C, device=dev, generator=g)
    unit = B * h * Tq * Tk * d
    res = {}
    def pair(*a):                      # fused forward, recomputing dQ + dK/dV backward pair (the round-2 backward)
        return ops.attention_core_fused(*a)
    for route, fn in (("fused", ops.attention_core_fused), ("2-kernel-bwd", pair),
                      ("narrow-fwd", ops.attention_core_fused), ("3-launch", ops.attention_core_unfused)):
        if route == "narrow-fwd" and d < 160:
            continue                   # the 8-wave split-head-dim forward exists for d >= 160 only
        if route == "3-launch" and 4 * B * h * Tq * Tk * 3 > 40e9:
            continue
        if route == "2-kernel-bwd" and d > 96:
            continue                   # the pair IS the backward there
        tf, tb = [], []
        for r in range(ROUNDS + 1):
            qq, kk, vv = (t.clone().requires_grad_(True) for t in (q, k, v))
            out = None
            def fwd():
                global out_
                out_ = fn(qq, kk, vv, h)
            with ops.kernel_flags(narrow_attn_fwd=(route == "2-kernel-bwd" or (route == "narrow-fwd"))):
                t1 = ev_time(fwd)
            with ops.kernel_flags(two_kernel_attn_bwd=(route == "2-kernel-bwd")):
                t2 = ev_time(lambda: out_.backward(do))
            if r:
                tf.append(t1); tb.append(t2)
        res[route] = (sorted(tf)[len(tf) // 2], sorted(tb)[len(tb) // 2])
    line = f"{name:22s} B={B:4d} Tq={Tq:4d} Tk={Tk:4d} h={h:2d} d={d:3d}: "
    for route, (tf, tb) in res.items():
        line += f"{route} fwd {tf:7.3f} ms ({4 * unit / tf / 1e9:6.1f} TF/s) bwd {tb:7.3f} ms ({10 * unit / tb / 1e9:6.1f} TF/s algorithmic) | "
    print(line, flush=True)
