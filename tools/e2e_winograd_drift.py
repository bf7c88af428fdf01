"""End-to-end score drift between kernel families: the SAME coalitions - same seeds, same host-drawn
randomness, same base weights - run with the planner's default routes (Winograd F(4x4) one-launch / three-launch, F(2x2)),
with F(4x4) switched off, and with every Winograd route switched off (direct LDS-patch kernels), and the per-coalition model
behaviours (FID / IS / precision / recall under the stand-in feature extractor, unlearn.py:807-837) side by side.
usage: python tools/e2e_winograd_drift.py full|reduced [n_seeds] -> text on stdout
  full:    BASELINE configs[1] at full size (gd_steps 1000 @ B = 128, 10 240 samples x 100 DDIM steps, widths [128, 256, 256, 256])
  reduced: the shape of tests/test_gpu_winograd_drift.py (widths [64, 128, 128, 128], 40 steps @ B = 64, 512 samples x 20 steps)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
import torch
from gad import ops
from gad.coalition import CoalitionEngine

mode = sys.argv[1] if len(sys.argv) > 1 else "reduced"
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if mode == "full":
    kw = dict(gd_steps=1000, n_samples=10240)
else:
    os.environ.setdefault("GAD_SYNTH_SCALE", "0.2")
    kw = dict(gd_steps=40, n_samples=512, num_inference_steps=20, fuse=8,
              unet_overrides=dict(block_out_channels=(64, 128, 128, 128)))
eng = CoalitionEngine("cifar100", device="cuda:0", **kw)
FAMILIES = [("default (Winograd F(4x4) / F(2x2) where planned)", {}), ("no F(4x4) (F(2x2) + direct)", dict(no_wino4=True)),
            ("direct kernels only", dict(no_wino=True))]
res = {}
for name, flags in FAMILIES:
    for s in range(nseeds):
        t0 = time.time()
        with ops.kernel_flags(**flags):
            r = eng.run_coalition(s, verbose=False)
        res[(name, s)] = r
        print(f"{name:52s} seed {s}: fid {r.fid_value:.6f}  is {r.inception_score:.6f}  precision {r.precision:.6f}  recall {r.recall:.6f}  "
              f"loss_last {r.loss_last:.6f}  ({time.time() - t0:.1f} s)", flush=True)
base = FAMILIES[-1][0]
print()
for name, _ in FAMILIES[:-1]:
    for s in range(nseeds):
        a, b = res[(name, s)], res[(base, s)]
        print(f"{name:52s} seed {s} vs direct: dFID {a.fid_value - b.fid_value:+.6f} ({abs(a.fid_value - b.fid_value) / abs(b.fid_value):.2e} rel)  "
              f"dIS {a.inception_score - b.inception_score:+.6f}  dP {a.precision - b.precision:+.6f}  dR {a.recall - b.recall:+.6f}")
fids = [res[(base, s)].fid_value for s in range(nseeds)]
if nseeds > 1:
    print(f"\ncoalition-to-coalition FID gap under the direct kernels: {max(fids) - min(fids):.6f} (what the Shapley regression consumes)")
