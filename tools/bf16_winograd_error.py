"""Why the bf16-operand mode has no Winograd route ("a bf16 Winograd F(2x2), or a documented no-Winograd decision
with numbers").  CPU emulation of what bf16-operand MFMAs would see: operands rounded to bf16 (RNE), products and sums in fp32 -
  direct   : x, w rounded                       -> conv                    (what conv3x3_patch_bf16_kernel computes)
  F(2x2)   : V = B^T x B, U = G w G^T in fp32, BOTH rounded to bf16 -> 16 products -> A^T M A in fp32
  F(4x4)   : the same with the 6 x 6 transforms (points 0, +-1, +-2, inf)
against an fp64 convolution, on SD-like shapes (unit-variance activations, weights ~ N(0, 1 / (9 Cin))).
usage: python tools/bf16_winograd_error.py"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)
bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def winograd(x, w, BT, G, AT, m, round_ops):
    n, c, h, wd = x.shape
    a = BT.shape[0]
    xp = F.pad(x, (1, 1, 1, 1))
    tiles = xp.unfold(2, a, m).unfold(3, a, m)                       # [n, c, th, tw, a, a]
    V = torch.einsum("ij,nctujk,lk->nctuil", BT.float(), tiles, BT.float())
    U = torch.einsum("ij,ocjk,lk->ocil", G.float(), w, G.float())
    if round_ops:
        V, U = bf(V), bf(U)
    M = torch.einsum("nctuil,ocil->notuil", V, U)
    Y = torch.einsum("ij,notujk,lk->notuil", AT.float(), M, AT.float())      # [n, o, th, tw, m, m]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(n, w.shape[0], h, wd)


print(f"{'shape':28s} {'form':10s} {'rms err / rms out':>18s} {'max err / rms out':>18s}")
for n, c, o, hw in [(2, 128, 128, 32), (1, 320, 320, 32), (1, 640, 640, 16), (1, 1280, 1280, 8)]:
    x = torch.randn(n, c, hw, hw)
    w = torch.randn(o, c, 3, 3) / (9 * c) ** 0.5
    ref = F.conv2d(x.double(), w.double(), padding=1)
    rms = ref.pow(2).mean().sqrt().item()
    rows = [("fp32 direct", F.conv2d(x, w, padding=1)), ("bf16 direct", F.conv2d(bf(x), bf(w), padding=1)),
            ("fp32 F(2x2)", winograd(x, w, BT2, G2, AT2, 2, False)), ("bf16 F(2x2)", winograd(x, w, BT2, G2, AT2, 2, True)),
            ("fp32 F(4x4)", winograd(x, w, BT4, G4, AT4, 4, False)), ("bf16 F(4x4)", winograd(x, w, BT4, G4, AT4, 4, True))]
    for name, y in rows:
        e = (y.double() - ref)
        print(f"N{n} {c:4d}->{o:<4d} {hw:2d}x{hw:<2d}        {name:10s} {e.pow(2).mean().sqrt().item() / rms:18.2e} {e.abs().max().item() / rms:18.2e}")
