#!/usr/bin/env python3
"""bench.py - Shapley coalitions/hour on the CIFAR-20 DDPM sFT cycle (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W            (N > 1: this process only launches the N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --workload cifar20-pruned | sd256 | sd512 | celeba | celeba-pruned [--precision bf16]   (one labelled line)
With no --workload the line is the headline (cifar20) and, at N = 1, carries the other workloads of the path as
`"secondary": {"cifar20-pruned": {...}, "sd512": {...}, "celeba": {...}, "celeba-pruned": {...}}`, each timed the same way
(barrier-bracketed, HIP-event kernel brackets) with its own `roofline`; `--no-secondary` skips them.

Workload (config.workload): one coalition of the reference's CIFAR-20 configuration =
gd_steps=1000 fine-tuning steps at B=128 (noise, antithetic t, add_noise, U-Net fwd, MSE, bwd,
clip 1.0, Adam 1e-4, EMA) + 10 240 samples x 100 DDIM steps (U-Net fwd + scheduler step;
reference batches of 32 with per-batch CPU-generator noise, 32 of them fused per launch), fp32.
ONE BENCH STEP = 1/1000 of that coalition, in the coalition's own proportions:
    1 training step (B=128)  +  1 sampler step at B=1024 (= 32 reference batches x 32 images x 1 DDIM step).
value = coalitions/hour summed over all ranks = K * world / 1000 / hours(max-over-ranks time of the K steps).
The score tail (FID features + float64 Frechet, ~1 % of the FLOPs, once per coalition) is not part of a
slice; `--full-coalition` runs one real, complete coalition (train -> EMA -> preview -> sample -> score) instead - with
the STAND-IN feature extractor (gad/scoring.py::FeatureNet, ~0.12 GFLOP/image): the reference's InceptionV3 (FID, IS) and
VGG16 (P/R) passes over 10 240 images (fid_score.py:26-29, precision_recall.py:31,42-43; ~0.5 PFLOP = ~3 % of the
17.6 PFLOP coalition) need URL-fetched weights and are NOT part of any number printed here (`config.score_tail`).
Each rank works on its own coalition (removal_seed = rank); the only collective is the final all_gather of
the per-rank records ("scaling": "weak").
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.join(ROOT, "group-attribution-for-diffusion-models_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")

GD_STEPS, N_SAMPLES, DDIM_STEPS, TRAIN_B, SAMPLE_B, FUSE = 1000, 10240, 100, 128, 32, 32
UNET_GFLOP_PER_IMG = 12.44          # forward, SURVEY §8d (6.222 GMAC)
BF16_MFMA_PEAK_TF = 2500.0          # dense bf16 MFMA (v_mfma_f32_32x32x16_bf16)
F32_MFMA_PEAK_TF = 157.3            # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
HBM_PEAK_GBPS = 8000.0              # MI355X_MICROARCH.md: HBM3E spec peak (6.29 TB/s measured with a float4 copy)


PRUNED_WIDTHS = (96, 192, 192, 192)      # magnitude pruning at ratio 0.3 keeps 32-channel GroupNorm groups whole
WORKLOADS = {
    "cifar20": dict(kind="cifar", widths=None),
    "cifar20-pruned": dict(kind="cifar", widths=PRUNED_WIDTHS),
    "sd256": dict(kind="sd", latent=32, batch=64),
    "sd512": dict(kind="sd", latent=64, batch=16),
    "celeba": dict(kind="ldm", pruned=False, batch=32),
    "celeba-pruned": dict(kind="ldm", pruned=True, batch=32),
}
SECONDARY = ("cifar20-pruned", "sd512", "celeba", "celeba-pruned")     # timed after the headline by the default N = 1 run
SCORE_TAIL = ("score tail outside the timed slice; --full-coalition runs it with the stand-in feature extractor - the "
              "reference's InceptionV3 / VGG16 passes (~0.5 PFLOP, ~3 % of a 17.6 PFLOP coalition; URL-fetched weights) are not included")

torch = None       # imported in main(): the launcher parent of `--gpus N` must not import it


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=list(WORKLOADS), default=None,
                    help="cifar20 = BASELINE configs[1] (the headline); cifar20-pruned = the same cycle at the pruned widths "
                         "[96,192,192,192] that unlearn.py:363-367 actually fine-tunes; sd256 / sd512 = one SD-1.x LoRA (r=256) "
                         "sFT training step at B=64 @ 32x32 / B=16 @ 64x64 latents (train_text_to_image_lora.py:1215-1311); "
                         "celeba / celeba-pruned = one CelebA-HQ LDM U-Net sFT training step at B=32 on 64x64 latents "
                         "(ddpm_config.py:395-450), unpruned / head-grouped-pruned (prune.py:337-342).  Default: cifar20 + the "
                         "others as `secondary` lines")
    ap.add_argument("--no-secondary", action="store_true", help="headline only")
    ap.add_argument("--in-flight", type=int, choices=[1, 2, 3, 4], default=3,
                    help="CIFAR workloads: coalitions in flight per GPU - k > 1 (default 3, what gad.launch / run_sharded run): one coalition's "
                         "sampling phase beside the training phases of the next k - 1, each on its own HIP stream; 1: strictly sequential")
    ap.add_argument("--secondary-steps", type=int, default=10)
    ap.add_argument("--widths", choices=["full", "pruned"], default=None, help="alias: --widths pruned = --workload cifar20-pruned")
    ap.add_argument("--full-coalition", action="store_true", help="time K complete coalitions instead of slices")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket contraction launches with events")
    ap.add_argument("--no-one-stream-pass", action="store_true",
                    help="--in-flight > 1: skip the second, one-stream pass of the same K steps that gives the roofline's one_stream_* keys")
    ap.add_argument("--precision", choices=["f32", "bf16", "bf16-operands"], default="f32",
                    help="f32 = the reference's default precision and the headline number; bf16 = what the reference's SD jobs run "
                         "(--mixed_precision=fp16, setup_train_commands.py:127): half-precision ACTIVATIONS in HBM (gad/half.py) for the "
                         "SD workloads, bf16 operands elsewhere; bf16-operands = bf16 MFMA operands with fp32 storage everywhere (the "
                         "earlier rounds' mode, kept for A/B).  Separate, explicitly labelled lines, never the default")
    ap.add_argument("--no-train-rate", action="store_true",
                    help="skip the separate U-Net steps/s measurement (PMC passes: keeps the launch mix = the timed region's)")
    ap.add_argument("--gd-steps", type=int, default=GD_STEPS)
    ap.add_argument("--n-samples", type=int, default=N_SAMPLES)
    ap.add_argument("--stub", action="store_true",
                    help="CPU rehearsal of the launcher / process group / final all_gather with a stub engine over gloo "
                         "(tests only: the line is marked \"stub\": true and is not a measurement)")
    a = ap.parse_args()
    if a.widths == "pruned":
        a.workload = "cifar20-pruned"
    a.secondary = a.workload is None and not a.no_secondary and not a.full_coalition and not a.stub
    a.workload = a.workload or "cifar20"
    return a


def launch_ranks(a) -> int:
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: this process becomes the launcher.  It starts
    N fresh children (one per GPU: RANK = LOCAL_RANK = r, WORLD_SIZE = N, MASTER_ADDR = 127.0.0.1) BEFORE anything
    touches the GPU - it imports neither torch nor the package - waits for them and returns the worst exit code.
    Rank 0's child prints the JSON line on the inherited stdout."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "_gad_launch", os.path.join(ROOT, "group-attribution-for-diffusion-models_amd", "gad", "launch.py"))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)                     # stdlib only: no torch, no HIP
    log(f"launcher: starting {a.gpus} ranks (one per GPU)")
    codes = launch.spawn_workers([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], a.gpus)
    if any(codes):
        log(f"launcher: rank exit codes {codes}")
    return max(abs(c) for c in codes)


class SliceRunner:
    """Holds one coalition in flight on this rank and advances it one slice at a time."""

    def __init__(self, engine, removal_seed, in_flight=1):
        import gad
        from gad import ops
        from gad.coalition import DeviceLoader, FusedSampler, antithetic_timesteps, seed_everything
        self.ops, self.antithetic = ops, antithetic_timesteps
        self.engine, self.dev = engine, engine.device
        remaining, _ = engine.coalition(removal_seed)
        seed_everything(engine.opt_seed)
        self.model, self.ema = engine.load_base()
        self.trainer = engine.make_trainer(self.model, self.ema)
        self.loader = DeviceLoader(engine.dataset, remaining, TRAIN_B, self.dev)
        self.it = iter(self.loader)
        # sampling model = the EMA weights of the base checkpoint (same shapes as the post-sFT EMA model)
        self.smodel, _ = engine.load_base()
        self.smodel.eval()
        self.sampler = FusedSampler(self.smodel, engine.sample_scheduler, SAMPLE_B, FUSE)
        engine.sample_scheduler.set_timesteps(DDIM_STEPS)
        self.ts = engine.sample_scheduler.timesteps.tolist()
        self.noise_pool = [self.sampler.initial_noise(list(range(g * FUSE, (g + 1) * FUSE)), [SAMPLE_B] * FUSE)
                           for g in range(2)]                       # pre-staged in HBM before the timed region
        self.group, self.ti = 0, 0
        self.x = self.noise_pool[0].clone()
        self.t = torch.empty(self.x.shape[0], device=self.dev, dtype=torch.int64)
        self.images_done = 0
        self.n_t = engine.train_scheduler.config.num_train_timesteps
        self.streams = None
        if in_flight > 1:                                       # CoalitionEngine.run_pipelined's streams: [sampler, trainers ...]
            torch.cuda.synchronize(self.dev)
            self.streams = tuple(torch.cuda.Stream(self.dev) for _ in range(in_flight))
            self.trainers = [self.trainer]
            for _ in range(in_flight - 2):                      # a further training phase (another coalition's): its own model and optimizer state
                m_, e_ = engine.load_base()
                self.trainers.append(engine.make_trainer(m_, e_))
            self.turn = 0
        if self.trainer.use_graph:                              # the one-time captures belong to building the runner, not to the W + K steps
            keep = self.trainer
            for i, tr in enumerate(self.trainers if self.streams is not None else [self.trainer]):
                self.trainer = tr
                ctx = torch.cuda.stream(self.streams[1 + i]) if self.streams is not None else contextlib.nullcontext()
                with ctx:
                    for _ in range(tr.GRAPH_WARMUP + 1):
                        self.train_step()
            self.trainer = keep
            torch.cuda.synchronize(self.dev)

    def train_step(self):
        try:
            image, _ = next(self.it)
        except StopIteration:
            self.it = iter(self.loader)
            image, _ = next(self.it)
        if image.shape[0] != TRAIN_B:                                   # short last batch of an epoch: the
            self.it = iter(self.loader)                                 # reference trains on it; the bench keeps
            image, _ = next(self.it)                                    # every step at the quoted B=128
        noise = torch.randn_like(image)
        ts = self.antithetic(self.n_t, image.shape[0], self.dev)
        return self.trainer.step(image, noise, ts)

    def sampler_step(self):
        ops, sch = self.ops, self.engine.sample_scheduler
        t = self.ts[self.ti]
        self.t.fill_(t)
        with torch.no_grad():
            eps = self.smodel.forward_nhwc(self.x, self.t)
        a_t, a_p = sch.step_coefficients(t)
        ops.ddim_step_raw(self.x, eps, a_t, a_p, 1.0, out=self.x)
        self.ti += 1
        if self.ti == len(self.ts):                                      # trajectory finished: quantise, next group
            img = ops.to_image01_raw(self.x)
            self.last_images = img.mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8)
            self.images_done += self.x.shape[0]
            self.group += 1
            self.x.copy_(self.noise_pool[self.group % len(self.noise_pool)])
            self.ti = 0

    def slice(self):
        if self.streams is not None:
            # two coalitions in flight: the training phase of one beside the sampling phase of another, each on its own HIP
            # stream (independent work: the small launches of the B = 128 training step fill the tails of the sampler's large ones)
            k = self.turn = (self.turn + 1) % len(self.trainers)       # the training phases take turns (run_pipelined)
            self.trainer = self.trainers[k]
            with torch.cuda.stream(self.streams[1 + k]):
                loss = self.train_step()
            with torch.cuda.stream(self.streams[0]):
                for _ in range(N_SAMPLES * DDIM_STEPS // GD_STEPS // (SAMPLE_B * FUSE)):
                    self.sampler_step()
            return loss
        loss = self.train_step()
        for _ in range(N_SAMPLES * DDIM_STEPS // GD_STEPS // (SAMPLE_B * FUSE)):      # 1
            self.sampler_step()
        return loss


class SDRunner:
    """One SD-1.x LoRA sFT step per bench step: the body of text_to_image/train_text_to_image_lora.py:1215-1311 as the
    kept entry point runs it - batch drawn from the HBM-resident latent cache, noise, t ~ U{0..999}, add_noise,
    UNet2DConditionModel (859.5 M frozen parameters + LoRA r=256 on all 32 attentions = 51.0 M trainable) forward, MSE,
    backward (LoRA gradients only), clip 1.0, AdamW 3e-4 (cosine), wd 1e-6 (src/ddpm_config.py:624-642)."""

    def __init__(self, dev, latent, batch, seed):
        import gad
        from gad.coalition import seed_everything
        seed_everything(seed)
        with torch.device(dev):
            self.net = gad.UNet2DConditionModel(sample_size=latent)
        self.net.to(dev)
        lora = self.net.inject_lora(rank=256)
        with torch.no_grad():                       # `up` starts at zero in a fresh LoRA; a coalition resumes a trained one
            for n, p in self.net.named_parameters():
                if n.endswith("lora_layer.up.weight"):
                    p.normal_(0.0, 0.02)
        self.n_lora = sum(p.numel() for p in lora)
        self.n_base = sum(p.numel() for p in self.net.parameters()) - self.n_lora
        sched = gad.DDPMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000)
        from gad import ops
        # the half-precision step is launch-bound on the host (~2 900 launches): replay it from a hipGraph (same kernels, same order)
        self.trainer = gad.FusedTrainer(self.net, sched, None, lr=3e-4, weight_decay=1e-6, adamw=True, max_grad_norm=1.0,
                                        params=lora, lr_schedule=gad.lr_lambda("cosine", 200, 0),
                                        use_graph=ops.half_activations() and not os.environ.get("GAD_NO_TRAIN_GRAPH"))
        g = torch.Generator(device=dev).manual_seed(seed)
        n_cache = 8 * batch                          # latent cache + per-sample text embeddings, resident in HBM
        self.latents = torch.randn(n_cache, 4, latent, latent, device=dev, generator=g) * 0.8
        self.text = torch.randn(n_cache, 77, 768, device=dev, generator=g) * 0.5
        self.batch, self.dev = batch, dev
        self.perm, self.pos = torch.randperm(n_cache, device=dev), 0
        if self.trainer.use_graph:                    # the one-time capture (two eager steps, then the capturing one) belongs to building the
            for _ in range(self.trainer.GRAPH_WARMUP + 1):      # runner, not to the W warm-up / K timed steps of the contract
                self.slice()

    def slice(self):
        if self.pos + self.batch > self.perm.numel():
            self.perm, self.pos = torch.randperm(self.perm.numel(), device=self.dev), 0
        sel = self.perm[self.pos:self.pos + self.batch]
        self.pos += self.batch
        x0 = self.latents.index_select(0, sel)
        noise = torch.randn_like(x0)
        ts = torch.randint(0, 1000, (self.batch,), device=self.dev).long()
        return self.trainer.step(x0, noise, ts, self.text.index_select(0, sel))


def _threads():
    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    return max(1, min(threads, int(os.environ.get("GAD_CPU_THREADS", "16"))))   # the GPU box grants a 16-CPU share


def cpu_baseline(engine):
    """The oracle (pure-PyTorch restatement = what "the reference's CPU path" can mean here: diffusers is
    not installable) timed on the host cores on a bounded sample: 2 training steps and 3 sampler steps at
    B=16, scaled per image to one coalition."""
    from oracle import diffusers_ref as R
    threads = _threads()
    torch.set_num_threads(threads)
    log(f"cpu_baseline: oracle on {threads} threads")
    torch.manual_seed(0)
    net = R.UNet2DModel(**engine.unet_cfg)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4)
    ema = R.EMAModel(net.parameters())
    sch = R.DDPMScheduler(**engine.config["scheduler_config"])
    B = 16
    g = torch.Generator().manual_seed(0)
    x, n = torch.rand(B, 3, 32, 32, generator=g) * 2 - 1, torch.randn(B, 3, 32, 32, generator=g)
    t = R.antithetic_timesteps(torch.randint(0, 1000, (B // 2 + 1,), generator=g), 1000, B)
    R.train_step(net, opt, ema, sch, x, n, t)                          # warm-up
    t0 = time.time()
    for _ in range(2):
        R.train_step(net, opt, ema, sch, x, n, t)
    t_train_img = (time.time() - t0) / 2 / B
    dd = R.DDIMScheduler()
    dd.set_timesteps(DDIM_STEPS)
    net.eval()
    t0 = time.time()
    with torch.no_grad():
        xs = n.clone()
        for ts in dd.timesteps[:3]:
            xs = dd.step(net(xs, ts).sample, ts, xs).prev_sample
    t_fwd_img = (time.time() - t0) / 3 / B
    coalition_s = GD_STEPS * TRAIN_B * t_train_img + N_SAMPLES * DDIM_STEPS * t_fwd_img
    return {"value": 3600.0 / coalition_s, "unit": "coalitions/hour", "cores": threads, "kind": "port",
            "sample": f"oracle (PyTorch-CPU fp32 restatement) on {threads} threads: 2 train steps + 3 DDIM sampler "
                      f"steps at B={B}, scaled per image to 1000x128 train images + 10240x100 sampler images",
            "train_s_per_image": t_train_img, "sampler_s_per_image": t_fwd_img}


def cpu_baseline_sd(latent, batch):
    """The SD oracle (oracle/sd_unet_ref.py, full SD-1.x width, LoRA r=256 on every attention projection) on the host
    cores: 1 warm-up + 2 LoRA training steps at B=1, scaled per image to the workload's batch."""
    from oracle import diffusers_ref as R
    from oracle.sd_unet_ref import CrossAttention, UNet2DConditionModel
    threads = _threads()
    torch.set_num_threads(threads)
    log(f"cpu_baseline: SD oracle on {threads} threads")
    torch.manual_seed(0)
    net = UNet2DConditionModel(sample_size=latent)
    for p in net.parameters():
        p.requires_grad_(False)
    lora = []
    for m in net.modules():
        if isinstance(m, CrossAttention):
            for lin in (m.to_q, m.to_k, m.to_v, m.to_out[0]):
                layer = R.LoRALinearLayer(lin.in_features, lin.out_features, rank=256)
                torch.nn.init.normal_(layer.up.weight, std=0.02)
                lin.set_lora_layer(layer)
                lora += list(layer.parameters())
    opt = torch.optim.AdamW(lora, lr=3e-4, weight_decay=1e-6)
    sch = R.DDPMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000)
    g = torch.Generator().manual_seed(0)
    x, n = torch.randn(1, 4, latent, latent, generator=g) * 0.8, torch.randn(1, 4, latent, latent, generator=g)
    ctx, t = torch.randn(1, 77, 768, generator=g) * 0.5, torch.tensor([500])

    def step():
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(net(sch.add_noise(x, n, t), t, ctx).sample, n)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(lora, 1.0)
        opt.step()
    step()
    t0 = time.time()
    for _ in range(2):
        step()
    s_img = (time.time() - t0) / 2
    return {"value": 1.0 / (s_img * batch), "unit": "steps/s", "cores": threads, "kind": "port",
            "sample": f"SD oracle (PyTorch-CPU fp32 restatement, SD-1.x widths, LoRA r=256) on {threads} threads: 2 LoRA "
                      f"training steps at B=1 on {latent}x{latent} latents, scaled per image to B={batch}",
            "train_s_per_image": s_img}


def stub_rank(a, rank, world):
    """--stub: the N-rank path on CPU (gloo) with the engine's GPU work replaced by a deterministic function of the
    seed: launcher, rendezvous, barrier-bracketed timing, run_sharded's final all_gather and the JSON contract are the
    product's.  Not a measurement."""
    import torch.distributed as dist
    from gad.coalition import CoalitionRecord, run_sharded

    class StubEngine:
        n_groups, device = 20, torch.device("cpu")

        def run_coalition(self, seed, verbose=False):
            time.sleep(0.01)
            return CoalitionRecord(seed, 100 - seed, seed, 10.0 + 0.5 * seed, 0.1, 0.01, 0.01, 1, [seed % 20])

        def jsonl_row(self, rec):
            return dict(removal_seed=rec.removal_seed, fid_value=rec.fid_value)

    if world > 1:
        dist.barrier()
    t0 = time.time()
    recs = run_sharded(StubEngine(), list(range(a.steps * world)), db_path=os.environ.get("GAD_STUB_DB"))
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    seen = [0]
    if world > 1:
        ids = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(ids, torch.tensor([rank]))
        seen = sorted(int(i.item()) for i in ids)
    if rank == 0:
        print(json.dumps({"metric": "shapley_coalitions_per_hour", "value": len(recs) / (dt / 3600.0), "unit": "coalitions/hour",
                          "n_gpus": dist.get_world_size() if world > 1 else 1, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "none", "data": "stub", "stub": True,
                          "ranks_seen": seen, "records_gathered": sorted(r.removal_seed for r in recs),
                          "config": {"workload": "STUB engine (CPU / gloo rehearsal of the N-rank path): not a measurement"}}),
              flush=True)
    if world > 1:
        dist.destroy_process_group()


def kernel_report(prof, dt, peak_tf, precision, workload):
    """roofline of the dominant kernel family + per-family table, from the live HIP-event brackets."""
    summ = prof.summary()
    kern = {f"{k[0]}_t{k[1]}_sk{k[2]}_v{k[3]}": dict(launches=v["launches"], avg_us=v["ms"] / v["launches"] * 1e3,
                                                   tflops=v["executed"] / v["ms"] / 1e9, frac=v["executed"] / v["ms"] / 1e9 / peak_tf,
                                                   algorithmic_bytes_per_launch=v["bytes"] / v["launches"],
                                                   **({"tflops_algorithmic": v["flops"] / v["ms"] / 1e9} if v["executed"] != v["flops"] else {}))
            for k, v in summ.items()}
    dom_key = max(summ, key=lambda k: summ[k]["ms"])
    d = summ[dom_key]
    nm = dom_key[0]
    if nm.startswith("attn_"):
        dd = int(nm.rsplit('_d', 1)[1])
        kern_ = "attn_fwd" if "fwd" in nm else ("attn_bwd1" if dd <= 96 else "attn_bwd_dq + attn_bwd_dkv")     # single-pass backward up to d = 96
        kname = f"{kern_}_f32_kernel<{dd}> ({nm}, Tq={dom_key[1]}, Tk={dom_key[2]}" + (", incl. attn_delta + attn_dq_reduce" if "bwd" in nm else "") + ")"
    elif "_patch" in nm:      # the name rocprofv3 shows for it
        wo = int(nm.rsplit("_w", 1)[1])
        ni = {8: 2, 4: 8}.get(wo, 1)
        if "wgrad" in nm:
            kname = f"wgrad3x3_patch_f32_kernel<{wo}> ({nm})"
        elif "bf16" in nm:
            kname = f"conv3x3_patch_bf16_kernel<{wo}, {ni}> ({nm})"
        else:
            tiles_ = "128- then 96-channel tiles, two launches" if dom_key[1] == 224 else f"{dom_key[1]}-channel tiles"
            kname = f"conv3x3_patch_f32_kernel<{wo}, {ni}, {'true' if 'dgrad' in nm else 'false'}, {'128 | 96' if dom_key[1] == 224 else dom_key[1]}> ({nm}, {tiles_})"
    elif "_wino4" in nm:      # Winograd F(4x4,3x3): the launches of one convolution, bracketed together
        if dom_key[1] in (32, 64, 65):
            kname = (f"wino4_input_kernel + wino4_fused{'' if dom_key[1] == 64 else '2'}_kernel ({nm}, {'64 tiles x 32' if dom_key[1] == 65 else f'{dom_key[1]} tiles x 64'} channels per workgroup: all 36 "
                     "products and the whole output transform in one launch; time = both launches)")
        else:
            kname = (f"wino4_input_kernel + wino4_gemm_kernel<64, 128> / <128, 64> (six-position products; small launches: 36 batched products on "
                     f"gemm_kernel<20, 22, tile, tile, 4, 1>) + wino4_output_kernel ({nm}: the three launches are one convolution; time = all of them)")
    elif "_wino" in nm:       # Winograd F(2x2,3x3): input transform + the 16-position MFMA loop, bracketed together
        kname = (f"wino_input_kernel + wino_gemm_kernel<{'64, 128' if dom_key[1] == 128 else '128, 64'}> ({nm}: the pair is one "
                 "convolution; time = both launches)")
    elif nm.startswith("hgemm") or nm.startswith("hconv"):      # the half-precision activation path's one contraction engine
        inst = "2, 2, 2, 5, 32" if dom_key[1] == 320 else "2, 2, 2, 2, 64"
        kname = f"hgemm_kernel<{inst}> ({nm}: 128 x {dom_key[1]} tiles, bf16 operands by LDS-DMA, splitk {dom_key[2]})"
    else:
        kname = f"gemm_kernel<{nm}, tile {dom_key[1]}, splitk {dom_key[2]}>"
    if nm.startswith("attn_") and dom_key[3] == 2:
        kname = kname.replace("_f32_kernel", "_bf16_kernel<.., HIO>").replace("attn_bwd1", "attn_bwd_dq + attn_bwd_dkv")
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_summary.json" if workload == "cifar20" else f"pmc_summary_{workload}.json")
    if precision == "bf16":                       # the half-precision activation path has its own counter passes
        pmc = os.path.join(ROOT, "profiles", f"pmc_summary_{workload}_bf16.json")
    if precision in ("f32", "bf16") and os.path.exists(pmc):
        j = json.load(open(pmc))
        if j.get("dominant_key") in (None, "_".join(str(x) for x in dom_key)):      # a stored pass of another kernel says nothing here
            traffic = j.get("dominant_kernel_hbm_bytes_per_launch")
            traffic_src = (f"STORED counter pass, not measured in this run: profiles/{os.path.basename(pmc)} "
                           f"({j.get('bench_workload', {}).get('source', j.get('source', 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE'))})")
    # `achieved` / `frac`: the MFMA work actually ISSUED over the family's time (for a Winograd route: all of its launches,
    # the HBM-bound input transform included) - a fraction of a roof, <= 1.  The direct-convolution count 2 M N K that every
    # other line of this file uses is kept beside it (`algorithmic_*`): Winograd F(4x4) issues 36/144 of it, F(2x2) 16/36.
    ex_tf, alg_tf = d["executed"] / d["ms"] / 1e9, d["flops"] / d["ms"] / 1e9
    alg_bytes = d["bytes"] / d["launches"]
    roof = {"bound": "mfma", "kernel": kname, "achieved": ex_tf, "peak": peak_tf, "unit": "TFLOP/s",
            "frac": ex_tf / peak_tf, "traffic": traffic, "traffic_ratio": (traffic / alg_bytes if traffic else None),
            "traffic_source": traffic_src,
            "launches": d["launches"], "avg_launch_us": d["ms"] / d["launches"] * 1e3,
            "executed_gflop_per_launch": d["executed"] / d["launches"] / 1e9,
            "algorithmic_gflop_per_launch": d["flops"] / d["launches"] / 1e9,
            "algorithmic_tflops": alg_tf, "algorithmic_speedup_vs_direct_peak": alg_tf / peak_tf,
            "algorithmic_bytes_per_launch": alg_bytes, "share_of_step_time": d["ms"] / (dt * 1e3)}
    if "ms_input" in d:
        # the two stages of the route, each against the roof that bounds it (an event between the two launches).  Launches whose
        # V was written by the producing GroupNorm (gn_wino4_kernel: the sampler's norm -> silu -> conv halves) have no input
        # stage of their own: `stage_input_launches` of the family's `launches` ran one.
        t_in, t_rest, n_in = d["ms_input"], d["ms"] - d["ms_input"], d.get("n_input", 0)
        if n_in:
            roof.update({
                "stage_input_kernel": "wino4_input_kernel" if "_wino4" in nm else "wino_input_kernel",
                "stage_input_bound": "hbm", "stage_input_launches": n_in, "stage_input_us": t_in / n_in * 1e3,
                "stage_input_gbps": d["bytes_input"] / t_in / 1e6, "stage_input_frac_of_hbm_peak": d["bytes_input"] / t_in / 1e6 / HBM_PEAK_GBPS})
        roof.update({
            "stage_products_kernel": kname.split(" + ", 1)[1].split(" (")[0],
            "stage_products_bound": "mfma", "stage_products_us": t_rest / d["launches"] * 1e3,
            "stage_products_tflops_executed": d["executed"] / t_rest / 1e9,
            "stage_products_frac_of_mfma_peak": d["executed"] / t_rest / 1e9 / peak_tf,
            "stage_products_algorithmic_bytes_per_launch": d["bytes_rest"] / d["launches"],
            "stage_products_algorithmic_gbps": d["bytes_rest"] / t_rest / 1e6,
            "stage_products_frac_of_hbm_peak": d["bytes_rest"] / t_rest / 1e6 / HBM_PEAK_GBPS,
            "stage_note": ("input transform: x in, V out (2.25x the input for F(4x4)), a pure HBM stream - run by the route only where the "
                           "producing GroupNorm did not write V itself; products: V and U in, y out (+ residual in) - MFMA-bound by design, "
                           "its algorithmic stream rate is listed against the HBM roof too")})
        if traffic and j.get("traffic_is") == "products_stage":
            # the stored counters are of the products kernel alone: against the convolution's algorithmic bytes (x + w + y: `traffic_ratio`)
            # and against what that stage must move given that V is its input (`traffic_ratio_vs_stage_bytes`)
            roof["traffic_ratio_vs_stage_bytes"] = traffic / (d["bytes_rest"] / d["launches"])
            roof["traffic_note"] = ("HBM bytes per launch of the products kernel (the whole route where GroupNorm wrote V); launches that ran "
                                    "wino4_input_kernel add its x-in / V-out stream: profiles/pmc_summary.json lists it per launch")
    # every forward launch of the route's family, whatever form the planner gave it (one-launch / three-launch)
    fam = [v for k, v in summ.items() if k[0] == nm]
    if len(fam) > 1:
        f_ms, f_ex, f_fl = sum(v["ms"] for v in fam), sum(v["executed"] for v in fam), sum(v["flops"] for v in fam)
        roof.update({"family_launches": sum(v["launches"] for v in fam), "family_share_of_step_time": f_ms / (dt * 1e3),
                     "family_frac": f_ex / f_ms / 1e9 / peak_tf, "family_algorithmic_tflops": f_fl / f_ms / 1e9})
    all_ms = sum(v["ms"] for v in summ.values())
    all_fl = sum(v["flops"] for v in summ.values())
    all_ex = sum(v["executed"] for v in summ.values())
    table = {"tflops": all_ex / all_ms / 1e9, "tflops_algorithmic": all_fl / all_ms / 1e9, "share_of_step_time": all_ms / (dt * 1e3), "by_instance": kern}
    return roof, table, all_fl, all_ex


class LDMRunner:
    """One CelebA-HQ LDM U-Net sFT step per bench step (BASELINE configs[2]; unlearn.py:560-642 on `celeba_config`,
    ddpm_config.py:395-450): B = 32 precomputed 3x64x64 VQ latents from the HBM-resident cache, noise, antithetic t,
    add_noise, UNet2DModel [224, 448, 672, 896] (attention at 32x32 / 16x16 / 8x8, heads of dim 32) forward, MSE, backward,
    clip 1.0, Adam, EMA.  `pruned`: the head-grouped magnitude-pruned network the sFT cycle actually fine-tunes
    (prune.py:337-342 at ratio 0.3: widths [160, 320, 480, 640], the heads stay, head dim 32 -> 23)."""

    def __init__(self, dev, pruned, batch, seed):
        import gad
        from gad.coalition import antithetic_timesteps, seed_everything
        from src.ddpm_config import DDPMConfig
        from unconditional_generation.prune import pruned_head_dim, pruned_width
        seed_everything(seed)
        cfg = DDPMConfig.celeba_config
        ucfg = dict(cfg["unet_config"])
        if pruned:
            boc, hd, groups = list(ucfg["block_out_channels"]), ucfg["attention_head_dim"], ucfg["norm_num_groups"]
            ucfg["attention_layout"] = [[w // hd, pruned_head_dim(w, hd, 0.3)] for w in boc]
            ucfg["block_out_channels"] = [pruned_width(w, 0.3, groups) for w in boc]
        self.ucfg = ucfg
        with torch.device(dev):
            self.model = gad.UNet2DModel(**ucfg)
        self.model.to(dev)
        keys = ("beta_start", "beta_end", "beta_schedule", "num_train_timesteps")
        sched = gad.DDPMScheduler(**{k: v for k, v in cfg["scheduler_config"].items() if k in keys})
        self.trainer = gad.FusedTrainer(self.model, sched, gad.EMAModel(self.model.parameters()), lr=1e-4, max_grad_norm=1.0)
        g = torch.Generator(device=dev).manual_seed(seed)
        self.latents = torch.randn(8 * batch, 3, 64, 64, device=dev, generator=g)       # vqvae_output.pt stand-in, resident
        self.batch, self.dev, self.pos, self.antithetic = batch, dev, 0, antithetic_timesteps
        self.n_params = sum(p.numel() for p in self.model.parameters())

    def slice(self):
        x0 = self.latents[self.pos:self.pos + self.batch]
        self.pos = (self.pos + self.batch) % self.latents.shape[0]
        return self.trainer.step(x0, torch.randn_like(x0), self.antithetic(1000, self.batch, self.dev))


def measure(a, name, steps, warmup, env, headline):
    """Build workload `name`, run `warmup` untimed and `steps` timed steps (barrier + synchronize on both sides) and
    return rank 0's line for it (None on the other ranks)."""
    import torch.distributed as dist
    import gad
    from gad import ops
    from gad.coalition import CoalitionEngine, CoalitionRecord, gather_records
    rank, world, dev, backend = env["rank"], env["world"], env["dev"], env["backend"]
    wl = WORKLOADS[name]
    peak_tf = F32_MFMA_PEAK_TF if a.precision == "f32" else BF16_MFMA_PEAK_TF
    log(f"rank {rank}/{world} on {dev}: workload {name}, building")
    engine = None
    if wl["kind"] == "cifar":
        over = dict(block_out_channels=tuple(wl["widths"])) if wl["widths"] else None
        engine = CoalitionEngine("cifar100", device=dev, gd_steps=a.gd_steps, n_samples=a.n_samples,
                                 sample_batch=SAMPLE_B, fuse=FUSE, num_inference_steps=DDIM_STEPS, unet_overrides=over)
        n_groups = engine.n_groups
    else:
        if a.full_coalition:
            print("bench.py: --full-coalition is a CIFAR-workload option", file=sys.stderr)
            sys.exit(2)
        n_groups = 258 if wl["kind"] == "sd" else 50            # artists / celebrities (src/ddpm_config.py: DatasetStats)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    prof, dt_train, n_tr, run, dt_prof = None, 0.0, 0, None, None
    if a.full_coalition:
        for i in range(warmup):
            engine.run_coalition(10_000 + rank)
        barrier()
        t0 = time.time()
        recs = []
        def show(r):
            log(f"coalition {r.removal_seed}: |S|={r.n_remaining} train {r.total_steps_time:.1f}s sample+score {r.total_sampling_time:.1f}s fid {r.fid_value:.4f}")
        if a.in_flight > 1:
            recs = engine.run_pipelined([rank + world * i for i in range(steps)], on_record=show, n_train=a.in_flight - 1)
        else:
            for i in range(steps):                                       # (stdout carries the JSON line only)
                recs.append(engine.run_coalition(rank + world * i, verbose=False))
                show(recs[-1])
        barrier()
        dt = time.time() - t0
        units = steps * world                                           # coalitions
    else:
        if wl["kind"] == "cifar":
            run = SliceRunner(engine, removal_seed=rank, in_flight=a.in_flight)
        elif wl["kind"] == "sd":
            run = SDRunner(dev, wl["latent"], wl["batch"], seed=rank)
        else:
            run = LDMRunner(dev, wl["pruned"], wl["batch"], seed=rank)
        log("runner ready; warm-up")
        for _ in range(warmup):
            run.slice()
        log("timed region")
        barrier()
        # The half-precision SD step is launch-bound on the host side (2 900 short launches per step): two events + a plan query around
        # every contraction cost ~10 ms of host time per step there and would be what the timed region measures.  Its timed region
        # therefore runs the product path as it is, and the SAME K steps are run once more right after it with the brackets on
        # (`dt_prof`): the per-kernel table and the roofline come from that pass (as the CIFAR workloads' one_stream_* keys do).
        separate_pass = wl["kind"] == "sd" and a.precision == "bf16" and not a.no_kernel_timing
        dt_prof = None
        if not a.no_kernel_timing and not separate_pass:
            prof = ops.GemmProfiler()
            ops.PROFILER = prof
        t0 = time.time()
        for _ in range(steps):
            loss = run.slice()
        barrier()
        dt = time.time() - t0
        log(f"timed region done: {dt:.2f}s for {steps} steps")
        ops.PROFILER = None
        if separate_pass:
            graphed, run.trainer.use_graph = run.trainer.use_graph, False          # the brackets need eager launches
            prof = ops.GemmProfiler()
            ops.PROFILER = prof
            t1 = time.time()
            for _ in range(steps):
                run.slice()
            barrier()
            dt_prof = time.time() - t1
            ops.PROFILER = None
            run.trainer.use_graph = graphed
            log(f"instrumented pass done: {dt_prof:.2f}s for {steps} steps")
        if wl["kind"] == "cifar":
            units = steps * world / float(GD_STEPS)
            # second half of BASELINE's metric: U-Net training steps/s (B=128, fwd+bwd+clip+Adam+EMA), outside the timed region
            one_stream = None
            if prof is not None and run.streams is not None and not a.no_one_stream_pass:
                # the same K steps once more on ONE stream: per-kernel durations without the other stream's share of the chip
                # (in the timed region a kernel's event bracket spans time in which the other coalition's kernels ran too)
                keep, run.streams = run.streams, None
                run.slice()
                barrier()
                ops.PROFILER = one_stream = ops.GemmProfiler()
                t1 = time.time()
                for _ in range(steps):
                    run.slice()
                barrier()
                dt_one = time.time() - t1
                ops.PROFILER = None
                run.streams = keep
            n_tr = 0 if a.no_train_rate else 10
            import contextlib
            on_train_stream = torch.cuda.stream(run.streams[1]) if run.streams is not None else contextlib.nullcontext()
            if run.streams is not None:
                run.trainer = run.trainers[0]
            barrier()
            t1 = time.time()
            with on_train_stream:                                       # (the stream whose allocator pool holds the trainer's blocks)
                for _ in range(n_tr):
                    run.train_step()
            barrier()
            dt_train = time.time() - t1
            n_rem = len(run.loader.x)
        else:
            units = steps * world                                       # training steps
            n_rem = run.latents.shape[0]
        recs = [CoalitionRecord(rank, n_rem, 0, float("nan"), float(loss.item()), dt, dt, steps, [])]
    # the single data-path collective: per-coalition records to every rank (rank 0 would write the jsonl)
    ranks_seen = [0]
    if world > 1:
        cdev = dev if backend == "nccl" else torch.device("cpu")
        packed = gather_records([r.pack(n_groups) for r in recs], CoalitionRecord.NSCALAR + n_groups, cdev)
        assert len(packed) == world * len(recs)
        ids = [torch.zeros(1, device=cdev, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(ids, torch.tensor([rank], device=cdev, dtype=torch.int64))     # device all_gather of the rank ids
        ranks_seen = sorted(int(i.item()) for i in ids)
        tmax = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        if n_tr:
            tmax.fill_(dt_train)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt_train = float(tmax.item())
    if rank != 0:
        return None

    f32 = a.precision == "f32"
    half_act = a.precision == "bf16" and wl["kind"] == "sd"           # the SD U-Net has the half-precision activation path (gad/half.py)
    dtype = ("f32" if f32 else
             "bf16 activations + gradients in HBM, bf16 MFMA operands, f32 accumulate / statistics / LoRA master weights / optimizer "
             "(the reference's --mixed_precision=fp16 arithmetic with bf16 as the 16-bit type)" if half_act else
             "bf16 operands, f32 accumulate/storage (NOT the reference default)")
    par = {"coalitions_in_flight": world, "parallelism": f"coalition-per-gpu x{world}"}
    if wl["kind"] == "cifar" and a.in_flight > 1:
        par = {"coalitions_in_flight": a.in_flight * world,
               "parallelism": (f"coalition-per-gpu x{world}, {a.in_flight} coalitions in flight per GPU: the sampling phase of one beside the "
                               f"training phase(s) of the next {a.in_flight - 1}, each on its own HIP stream, one optimizer step and one "
                               "DDIM step enqueued per turn (gad.coalition.CoalitionEngine.run_pipelined; --in-flight 1 = strictly sequential)")}
    if wl["kind"] == "cifar":
        widths = list(wl["widths"]) if wl["widths"] else [128, 256, 256, 256]
        nparam = sum(p.numel() for p in run.model.parameters()) if not a.full_coalition else None
        out = {"metric": "shapley_coalitions_per_hour", "value": units / (dt / 3600.0), "unit": "coalitions/hour"}
        workload = (f"CIFAR-20 DDPM sFT coalition (BASELINE configs[1]{'' if not wl['widths'] else ', PRUNED widths - the shape unlearn.py:363-367 fine-tunes'}): "
                    f"gd_steps=1000 @B=128 + 10240 samples x 100 DDIM steps @B=32 (32 batches fused/launch), UNet2DModel widths {widths}"
                    + (f" {nparam / 1e6:.2f}M params" if nparam else "") + " fp32" + ("" if f32 else " storage, bf16 MFMA operands") + "; "
                    + ("step = one complete coalition" if a.full_coalition else
                       "step = 1/1000 coalition = 1 train step + 1 sampler step @B=1024"
                       + (f", enqueued on {a.in_flight} HIP streams (the train step belongs to a later coalition: independent work)" if a.in_flight > 1 else "")))
        config = {"workload": workload, "score_tail": SCORE_TAIL, **par}
    elif wl["kind"] == "sd":
        out = {"metric": "sd_lora_unet_train_steps_per_sec", "value": units / dt, "unit": "steps/s"}
        config = {"workload": (f"SD-1.x LoRA sFT training step (BASELINE configs[3]/[4] body, train_text_to_image_lora.py:1215-1311): "
                               f"B={wl['batch']} x 4x{wl['latent']}x{wl['latent']} latents ({8 * wl['latent']}x{8 * wl['latent']} images), ctx [B,77,768], "
                               f"UNet2DConditionModel {run.n_base / 1e6:.1f}M frozen + LoRA r=256 on 32 attentions ({run.n_lora / 1e6:.1f}M trainable), "
                               f"AdamW + clip, " + ("fp32" if f32 else "bf16 activations (gad/half.py), fp32 LoRA / optimizer" if half_act
                                                    else "fp32 storage, bf16 MFMA operands") + "; step = one training step"),
                  "images_per_s": units * wl["batch"] / dt, **par}
    else:
        out = {"metric": "ldm_unet_train_steps_per_sec", "value": units / dt, "unit": "steps/s"}
        lay = run.ucfg.get("attention_layout")
        config = {"workload": (f"CelebA-HQ LDM sFT training step (BASELINE configs[2] body, unlearn.py:560-642 on celeba_config): "
                               f"B={wl['batch']} x 3x64x64 latents, UNet2DModel widths {list(run.ucfg['block_out_channels'])} "
                               + (f"head-grouped-pruned (prune.py:337-342, ratio 0.3), attention [heads, dim] per level {lay}, " if lay
                                  else f"attention heads of dim {run.ucfg['attention_head_dim']}, ")
                               + f"{run.n_params / 1e6:.1f}M params, Adam + clip + EMA, fp32" + ("" if f32 else " storage, bf16 MFMA operands")
                               + "; step = one training step"),
                  "images_per_s": units * wl["batch"] / dt, **par}
    out.update({"n_gpus": dist.get_world_size() if world > 1 else 1, "steps": steps, "warmup": warmup,
                "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": dtype, "data": "synthetic", "config": config})
    if world > 1:
        out["rccl_ranks_seen"] = ranks_seen                       # from the collective, not from the flag
        out["dist_backend"] = dist.get_backend()
    # published reference figures (empirical_verification.ipynb:128,132): 3.27 coalitions per GPU-hour
    # (fp32, pruned CIFAR model, unnamed GPU); 0.74 SD LoRA steps/s at B=64 @256^2 (fp16, RTX 6000) -> per-GPU ratios
    if wl["kind"] == "cifar" and f32:
        out["vs_baseline"] = out["value"] / world / 3.27
    if name == "sd256" and not f32:
        out["vs_baseline"] = out["value"] / world / 0.74
    if wl["kind"] == "cifar" and (a.gd_steps != GD_STEPS or a.n_samples != N_SAMPLES):      # a reduced workload is not the metric
        config["workload"] += f" -- OVERRIDDEN: gd_steps={a.gd_steps}, n_samples={a.n_samples} (not the BASELINE workload)"
        out["vs_baseline"] = None
    if prof is not None:
        torch.cuda.synchronize(dev)
        out["roofline"], out["contraction_kernels"], all_fl, all_ex = kernel_report(prof, dt_prof if dt_prof else dt, peak_tf, a.precision, name)
        if dt_prof:
            out["roofline"]["measured"] = ("per-launch HIP-event brackets in a SEPARATE pass of the same K steps right after the timed region: this step is "
                                           "launch-bound on the host, and the brackets (two events + a plan query per contraction) cost about 10 ms of host "
                                           "time per step - the timed region (`value`, `ms_per_step`) runs the product path without them; shares are of the "
                                           "instrumented pass's step time")
            out["instrumented_pass"] = {"ms_per_step": dt_prof / steps * 1e3, "steps": steps}
            config["train_step_graph"] = bool(getattr(run.trainer, "_graph", None)) and not run.trainer._graph_failed
        out["unet_tflops_per_gpu"] = all_fl / dt / 1e12            # algorithmic FLOPs (direct-form count) of every contraction / attention launch
        out["unet_tflops_executed_per_gpu"] = all_ex / dt / 1e12   # the MFMA work issued (Winograd launches: 36/144 or 16/36 of the direct count)
        out["path_mfma_frac"] = out["unet_tflops_executed_per_gpu"] / peak_tf        # whole path, executed work / step time / peak: <= 1
        out["path_mfma_frac_algorithmic"] = out["unet_tflops_per_gpu"] / peak_tf    # the same on the direct-form count (> executed where Winograd runs)
        if wl["kind"] == "cifar" and one_stream is not None:
            r1, _, _, ex1 = kernel_report(one_stream, dt_one, peak_tf, a.precision, name)
            roof = out["roofline"]
            roof["measured"] = ("timed region, several coalitions in flight: a kernel's event bracket there includes the time the other stream's "
                                "kernels held part of the chip, so `frac` is the kernel's rate WHILE SHARING; the one_stream_* keys are the "
                                "same K steps run again on one stream right after the timed region (kernel quality without sharing)")
            for k in ("frac", "achieved", "avg_launch_us", "share_of_step_time", "stage_input_us", "stage_input_gbps", "stage_input_frac_of_hbm_peak",
                      "stage_products_us", "stage_products_tflops_executed", "stage_products_frac_of_mfma_peak", "family_frac"):
                if k in r1:
                    roof["one_stream_" + k] = r1[k]
            roof["one_stream_kernel"] = r1["kernel"][:60]
            out["one_stream"] = {"ms_per_step": dt_one / steps * 1e3, "value": units / dt_one * 3600.0,
                                 "path_mfma_frac": ex1 / dt_one / 1e12 / peak_tf}
    elif name == "cifar20" and not a.full_coalition:
        fl = (3 * UNET_GFLOP_PER_IMG * TRAIN_B + UNET_GFLOP_PER_IMG * N_SAMPLES * DDIM_STEPS / GD_STEPS) * 1e9
        out["unet_tflops_per_gpu"] = fl * steps / dt / 1e12
        out["path_mfma_frac_algorithmic"] = out["unet_tflops_per_gpu"] / peak_tf     # (--no-kernel-timing: no executed count is taken)
    if n_tr:
        out["unet_train_steps_per_s"] = {"value": n_tr * world / dt_train, "batch_per_gpu": TRAIN_B, "n_gpus": world,
                                         "ms_per_step": dt_train / n_tr * 1e3, "reference": 3.81,   # 3.81 steps/s, 1 GPU
                                         "meaning": ("one model, one GPU" if world == 1 else
                                                     f"{world} INDEPENDENT replicas (one coalition's model per GPU, no gradient "
                                                     f"exchange), summed - not a data-parallel rate")}
    if headline and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(engine) if wl["kind"] == "cifar" else cpu_baseline_sd(wl["latent"], wl["batch"]) \
            if wl["kind"] == "sd" else None
    return out


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a))                                      # parent: never touches the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:                                                # never silently benchmark a different N
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch N ranks with --gpus N (torch.distributed.run "
              f"--nproc-per-node N, or plain `python bench.py --gpus N`)", file=sys.stderr)
        sys.exit(2)
    global torch
    import torch
    import torch.distributed as dist
    backend = os.environ.get("GAD_DIST_BACKEND", "gloo" if a.stub else "nccl")   # "gloo" only to rehearse the N>1 path
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if a.stub:
        if world > 1:
            dist.init_process_group(backend)
        return stub_rank(a, rank, world)
    if os.environ.get("GAD_SHARE_GPU0"):
        local = 0
    elif local >= torch.cuda.device_count():
        print(f"bench.py: rank {rank} wants cuda:{local} but only {torch.cuda.device_count()} GPUs are visible",
              file=sys.stderr)
        sys.exit(2)
    if world > 1:
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)

    import gad
    gad.set_operand_precision(a.precision)
    env = dict(rank=rank, world=world, dev=dev, backend=backend)
    out = measure(a, a.workload, a.steps, a.warmup, env, headline=True)
    if a.secondary and world == 1:
        # the other workloads of the path, timed the same way after the headline (N = 1 only: the N > 1 runs stay short)
        import gc
        out["secondary"] = {}
        import copy
        # (name, precision): the fp32 workloads, then the SD steps in the arithmetic the reference's SD jobs actually run
        # (--mixed_precision=fp16 -> half-precision activations, gad/half.py), labelled by their key and their `dtype`
        todo = [(n, a.precision, n) for n in SECONDARY]
        if a.precision == "f32":
            todo += [("sd512", "bf16", "sd512-bf16"), ("sd256", "bf16", "sd256-bf16")]
        for name, prec, key in todo:
            gc.collect()
            torch.cuda.empty_cache()
            a2 = copy.copy(a)
            a2.precision = prec
            try:
                gad.set_operand_precision(prec)
                sec = measure(a2, name, a.secondary_steps, a.warmup, env, headline=False)
                sec.pop("contraction_kernels", None)                    # the per-instance table of the headline is enough
                out["secondary"][key] = sec
            except Exception as e:                                      # a secondary line never costs the headline
                out["secondary"][key] = {"error": f"{type(e).__name__}: {e}"}
                log(f"secondary workload {key} failed: {type(e).__name__}: {e}")
            finally:
                gad.set_operand_precision(a.precision)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
