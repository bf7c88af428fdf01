"""Run complete coalitions on the GPU box (engine end-to-end) and write the jsonl db."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
import torch
from gad.coalition import CoalitionEngine, run_sharded
gd, ns, nseeds = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
out = sys.argv[4]
t0 = time.time()
eng = CoalitionEngine("cifar100", device="cuda:0", gd_steps=gd, n_samples=ns)
print(f"engine ready in {time.time()-t0:.1f}s", flush=True)
if os.path.exists(out): os.remove(out)
t0 = time.time()
recs = run_sharded(eng, list(range(nseeds)), db_path=out, verbose=True)
dt = time.time() - t0
print(f"{nseeds} coalitions in {dt:.1f}s -> {nseeds/dt*3600:.2f} coalitions/hour", flush=True)
rows = [json.loads(l) for l in open(out)]
print({k: (v if not isinstance(v, list) else len(v)) for k, v in rows[0].items()})
