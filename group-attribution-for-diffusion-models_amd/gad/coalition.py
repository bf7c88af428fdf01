"""The per-coalition sFT cycle (reference unconditional_generation/unlearn.py:267-969,
method "gd", removal_dist "shapley") as an in-process engine, and the one-coalition-per-GPU
scheduler that replaces the reference's SLURM job array
(text_to_image/experiments/setup_unlearn_commands.py:160-214, unlearn.job:7-22).

One process per GPU; rank r runs the coalitions whose ``removal_seed % world == r``; the
only exchange is one ``all_gather`` (RCCL over xGMI) of fixed-size score records at the end.
"""
from __future__ import annotations

import json
import os
import time
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import ops
from .nn import UNet2DModel
from .pipelines import DDPMPipeline
from .schedulers import DDIMScheduler, DDPMScheduler
from .training import EMAModel, FusedTrainer


def seed_everything(seed: int):
    """lightning.seed_everything as used at unlearn.py:359."""
    import random
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def antithetic_timesteps(n_train_timesteps: int, batch: int, device, generator=None) -> torch.Tensor:
    """main.py:684-696 / unlearn.py:591-603: t1 ~ U{0..N-1}^(B//2+1); t = cat([t1, N - t1 - 1])[:B]."""
    t1 = torch.randint(0, n_train_timesteps, (batch // 2 + 1,), device=device, generator=generator).long()
    return torch.cat([t1, n_train_timesteps - t1 - 1], dim=0)[:batch]


class DeviceLoader:
    """DataLoader(Subset(dataset, remaining_idx), batch_size, shuffle=True) with the whole subset
    resident in HBM (10 000 CIFAR images = 123 MB): per epoch one device randperm, batches are
    index_select views; RandomHorizontalFlip is applied per sample on the device.  The last batch
    of an epoch is short (drop_last=False), as in the reference (unlearn.py:373-379)."""

    def __init__(self, dataset, idx: Sequence[int], batch_size: int, device, flip=None, generator=None):
        self.generator = generator                              # None: the device's default generator (seed_everything)
        self.x = dataset.device_tensor(device, idx)             # [n,3,H,W] in [-1,1]
        self.labels = torch.as_tensor([dataset.targets[i] for i in idx], device=device)
        if flip is None:                                        # the dataset's own transform chain decides (datasets.py:412-477)
            flip = bool(getattr(dataset, "flip", True))
        self.bs, self.flip, self.device = batch_size, flip, device

    def __len__(self):
        return (self.x.shape[0] + self.bs - 1) // self.bs

    def __iter__(self):
        n = self.x.shape[0]
        perm = torch.randperm(n, device=self.device, generator=self.generator)
        for s in range(0, n, self.bs):
            sel = perm[s:s + self.bs]
            xb = self.x.index_select(0, sel)
            if self.flip:
                m = torch.rand(xb.shape[0], device=self.device, generator=self.generator) < 0.5
                xb = torch.where(m[:, None, None, None], xb.flip(-1), xb)
            yield xb, self.labels.index_select(0, sel)


class FusedSampler:
    """generate_images (src/diffusion_utils.py:319-357) with several reference batches fused into
    one U-Net launch: every reference batch b of `batch_size` images still draws its initial noise
    from ``torch.Generator().manual_seed(b)`` on the host (bit-identical noise), but `fuse` of them
    are stacked along N so the contraction kernels see M = fuse*batch_size*H*W rows.  The U-Net
    has no cross-sample op (GroupNorm and attention are per sample), so every sample sees the same arithmetic (up to
    the fp32 summation order of the tile / split plan picked for the launch width)."""

    def __init__(self, unet: UNet2DModel, scheduler: DDIMScheduler, batch_size=32, fuse=32):
        self.unet, self.sch, self.bs, self.fuse = unet, scheduler, batch_size, fuse
        self.device = unet.device

    def initial_noise(self, counters: Sequence[int], sizes: Sequence[int]) -> torch.Tensor:
        cfg = self.unet.config
        ss = cfg.sample_size
        parts = [torch.randn((n, cfg.in_channels, ss, ss), generator=torch.Generator().manual_seed(c),
                             dtype=torch.float32) for c, n in zip(counters, sizes)]
        return ops.nchw_to_nhwc_raw(torch.cat(parts, 0).to(self.device))

    def denoise_steps(self, x: torch.Tensor, num_inference_steps: int):
        """Generator form of `denoise`: yields after every enqueued DDIM step (the pipelined scheduler interleaves another
        coalition's training steps there), returns the NHWC images in [0,1]."""
        sch = self.sch
        sch.set_timesteps(num_inference_steps)
        clip = float(sch.config.clip_sample_range) if sch.config.clip_sample else 0.0
        t = torch.empty(x.shape[0], device=x.device, dtype=torch.int64)
        for ts in sch.timesteps.tolist():
            with torch.no_grad():
                t.fill_(ts)
                eps = self.unet.forward_nhwc(x, t)
                a_t, a_p = sch.step_coefficients(ts)
                ops.ddim_step_raw(x, eps, a_t, a_p, clip, out=x)
            yield
        with torch.no_grad():
            return ops.to_image01_raw(x)

    def denoise(self, x: torch.Tensor, num_inference_steps: int) -> torch.Tensor:
        """x NHWC noise -> NHWC images in [0,1] after the full DDIM trajectory."""
        return _drain(self.denoise_steps(x, num_inference_steps))

    def generate_steps(self, n_samples: int, num_inference_steps: int):
        """Generator form of `generate` (yields once per DDIM step of every fused launch group)."""
        sizes = [self.bs] * (n_samples // self.bs)
        if n_samples % self.bs:
            sizes.append(n_samples % self.bs)
        out = []
        for g0 in range(0, len(sizes), self.fuse):
            cs = list(range(g0, min(g0 + self.fuse, len(sizes))))
            img = yield from self.denoise_steps(self.initial_noise(cs, [sizes[c] for c in cs]), num_inference_steps)
            with torch.no_grad():
                q = img.mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8)       # .mul(255).add_(0.5).clamp_(0,255) -> uint8
                out.append(q.permute(0, 3, 1, 2).float().div_(255))             # ToTensor(): /255
        return torch.cat(out, 0)

    def generate(self, n_samples: int, num_inference_steps: int) -> torch.Tensor:
        """-> float tensor [n,3,H,W] holding k/255 values (uint8 round trip of :344-355), on device."""
        return _drain(self.generate_steps(n_samples, num_inference_steps))


def _drain(gen):
    """Run a step generator to its end and hand back its return value."""
    while True:
        try:
            next(gen)
        except StopIteration as e:
            return e.value


@dataclass
class CoalitionRecord:
    """What one coalition contributes to the .jsonl "db" (unlearn.py:960-968) - and the payload of
    the final all-gather: 11 fixed float64 scalars, `n_extra` engine-defined behaviour scalars (`extra`: CelebA entropy
    and cluster counts, SD aesthetic / CLIP quantiles ...; the engine names them in `extra_keys`), then the contributor mask."""
    removal_seed: int
    n_remaining: int
    n_removed: int
    fid_value: float
    loss_last: float
    total_steps_time: float
    total_sampling_time: float
    trained_steps: int
    remaining_classes: List[int] = field(default_factory=list)
    inception_score: float = float("nan")
    precision: float = float("nan")
    recall: float = float("nan")
    extra: List[float] = field(default_factory=list)

    NSCALAR = 11        # float64 scalars ahead of the extras and the contributor mask in the packed record

    def pack(self, n_groups: int, n_extra: int = 0) -> torch.Tensor:
        if len(self.extra) != n_extra:
            raise ValueError(f"record carries {len(self.extra)} extra scalars, the engine declares {n_extra}")
        v = torch.zeros(self.NSCALAR + n_extra + n_groups, dtype=torch.float64)
        v[:self.NSCALAR] = torch.tensor([self.removal_seed, self.n_remaining, self.n_removed, self.fid_value,
                                         self.loss_last, self.total_steps_time, self.total_sampling_time,
                                         self.trained_steps, self.inception_score, self.precision, self.recall],
                                        dtype=torch.float64)
        if n_extra:
            v[self.NSCALAR:self.NSCALAR + n_extra] = torch.tensor([float("nan") if x is None else float(x) for x in self.extra],
                                                                  dtype=torch.float64)
        for c in self.remaining_classes:
            v[self.NSCALAR + n_extra + int(c)] = 1.0
        return v

    @classmethod
    def unpack(cls, v: torch.Tensor, n_extra: int = 0):
        s = v[:cls.NSCALAR].tolist()
        extra = v[cls.NSCALAR:cls.NSCALAR + n_extra].tolist()
        mask = v[cls.NSCALAR + n_extra:]
        return cls(int(s[0]), int(s[1]), int(s[2]), s[3], s[4], s[5], s[6], int(s[7]),
                   [i for i in range(mask.numel()) if mask[i] > 0.5], s[8], s[9], s[10], extra)


class CoalitionEngine:
    """Everything one rank needs to run coalitions back to back on its GPU."""

    def __init__(self, dataset_name="cifar100", device="cuda:0", base_state: Optional[dict] = None,
                 gd_steps: Optional[int] = None, n_samples=10240, sample_batch=32, fuse=32,
                 num_inference_steps=100, opt_seed=42, by_class=True, preview=True,
                 unet_overrides: Optional[dict] = None, feature_dims=2048):
        from src.datasets import create_dataset
        from src.ddpm_config import DDPMConfig
        from .scoring import extractor_tag  # noqa: F401

        self.device = torch.device(device)
        self.dataset_name = dataset_name
        cfg_name = {"cifar100": "cifar100_config", "cifar": "cifar_config", "cifar2": "cifar2_config",
                    "toy2": "cifar100_config"}[dataset_name]
        self.config = {**getattr(DDPMConfig, cfg_name)}
        self.unet_cfg = dict(self.config["unet_config"])
        if unet_overrides:
            self.unet_cfg.update(unet_overrides)
        self.gd_steps = gd_steps if gd_steps is not None else self.config["training_steps"]["gd"]
        self.n_samples, self.sample_batch, self.fuse = n_samples, sample_batch, fuse
        # coalitions in flight on this GPU: 1 = strictly one after the other; k > 1 = the sampling phase of one beside the training
        # phases of the next k - 1, each on its own HIP stream (run_pipelined; what run_sharded / gad.launch use)
        self.in_flight = int(os.environ.get("GAD_IN_FLIGHT", "3"))
        # the training step's add_noise -> forward -> loss -> backward replayed from a hipGraph (FusedTrainer(use_graph=True): the step is ~1 500
        # short launches at B = 128 - host-bound when enqueued one by one; same kernels, same order, bit-identical records)
        self.train_graph = os.environ.get("GAD_TRAIN_GRAPH", "0") == "1"
        self.num_inference_steps, self.opt_seed, self.by_class, self.preview = num_inference_steps, opt_seed, by_class, preview
        self.dataset = create_dataset(dataset_name, train=True)
        self.n_groups = len(set(self.dataset.targets))
        self.train_scheduler = DDPMScheduler(**self.config["scheduler_config"])
        self.sample_scheduler = DDIMScheduler()            # build_pipeline: DDPMPipeline(unet, DDIMScheduler()) :311
        # the "pruned + fine-tuned" starting point every coalition resumes from (load_ckpt_model :111-205)
        if base_state is None:
            torch.manual_seed(0)
            m = UNet2DModel(**self.unet_cfg)
            ema = EMAModel(m.parameters())
            ema.optimization_step = 10000                   # as after prune_fine_tune (ddpm_config.py:207-215)
            base_state = {"unet": {k: v.clone() for k, v in m.state_dict().items()}, "unet_ema": ema.state_dict()}
        self.base_state = base_state
        from . import scoring
        self.feature_net = scoring.default_extractor(feature_dims, self.device)
        scoring._REF_STATS["net"] = self.feature_net          # one extractor for every score of this process
        self.opt_kwargs = dict(self.config["optimizer_config"]["kwargs"])
        self.adamw = self.config["optimizer_config"]["class_name"] == "AdamW"

    # -- pieces ---------------------------------------------------------------------------------
    def coalition(self, removal_seed: int):
        from src.datasets import remove_data_by_shapley
        return remove_data_by_shapley(self.dataset, seed=removal_seed, by_class=self.by_class)

    def load_base(self):
        model = UNet2DModel(**self.unet_cfg)
        model.load_state_dict(self.base_state["unet"])
        model.to(self.device)
        ema = EMAModel(model.parameters())                  # defaults, then overridden (diffusion_utils.py:193-198)
        ema.load_state_dict(self.base_state["unet_ema"])
        return model, ema

    def make_trainer(self, model, ema):
        kw = self.opt_kwargs
        return FusedTrainer(model, self.train_scheduler, ema, lr=kw.get("lr", 1e-4),
                            weight_decay=kw.get("weight_decay", 0.0), adamw=self.adamw, max_grad_norm=1.0,
                            use_graph=getattr(self, "train_graph", False))

    def score(self, images01: torch.Tensor) -> dict:
        """fid_value, is, precision, recall (unlearn.py:807-837) under the stand-in feature net."""
        from .scoring import global_scores_against_dataset
        return global_scores_against_dataset(images01, self.dataset, self.device, 512, self.feature_net.dims)

    # -- the cycle -------------------------------------------------------------------------------
    def train_phase(self, removal_seed: int, own_generator=False):
        """Generator: the sparsified fine-tuning of one coalition (unlearn.py:548-644) on the CURRENT stream, yielding after every
        enqueued optimizer step; returns the hand-over state for `sample_phase`.  No host synchronisation inside.
        `own_generator`: draw from a device generator of this phase's own, seeded like `seed_everything` seeds the default one
        (same Philox stream, same draws) - for phases that run beside other phases and must not share the default generator."""
        remaining_idx, removed_idx = self.coalition(removal_seed)
        gen = None
        if own_generator:
            gen = torch.Generator(device=self.device).manual_seed(self.opt_seed)
        else:
            seed_everything(self.opt_seed)                                # unlearn.py:359
        model, ema = self.load_base()
        trainer = self.make_trainer(model, ema)
        loader = DeviceLoader(self.dataset, remaining_idx, self.config["batch_size"], self.device, generator=gen)
        n_t = self.train_scheduler.config.num_train_timesteps
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        steps = 0
        loss = torch.zeros(1, device=self.device)
        while steps < self.gd_steps:                                      # unlearn.py:558-642
            for image, _ in loader:
                noise = torch.randn(image.shape, device=image.device, dtype=image.dtype, generator=gen) if gen is not None else torch.randn_like(image)
                ts = antithetic_timesteps(n_t, image.shape[0], self.device, generator=gen)
                loss = trainer.step(image, noise, ts)
                steps += 1
                yield
                if steps == self.gd_steps:
                    break
        # EMA weights are used for inference (unlearn.py:751-753); the fine-tuned ones are not kept
        model.flat[0].copy_(trainer.ema_flat)
        ops.WEIGHT_EPOCH[0] += 1                                         # flat copy: no per-parameter version bump
        model.eval()
        ev1.record()
        return dict(removal_seed=removal_seed, remaining_idx=remaining_idx, removed_idx=removed_idx, model=model, loss=loss,
                    steps=steps, train_events=(ev0, ev1))

    def sample_phase(self, st: dict, preview_generator=None):
        """Generator: sampling + scoring of a fine-tuned coalition (unlearn.py:751-837) on the CURRENT stream, yielding after
        every enqueued DDIM step; returns the CoalitionRecord (its score tail synchronises with the host)."""
        model = st["model"]
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        if self.preview:                                                  # unlearn.py:761-765 (global RNG)
            pipe = DDPMPipeline(model, self.sample_scheduler)
            pipe.use_graph = False                                        # one-off call: not worth a capture
            pipe(batch_size=self.config["n_samples"], num_inference_steps=self.num_inference_steps,
                 output_type="tensor", generator=preview_generator)
            yield
        sampler = FusedSampler(model, self.sample_scheduler, self.sample_batch, self.fuse)
        images = yield from sampler.generate_steps(self.n_samples, self.num_inference_steps)
        sc = self.score(images)
        ev1.record()
        ev1.synchronize()
        remaining_idx, removed_idx = st["remaining_idx"], st["removed_idx"]
        return CoalitionRecord(st["removal_seed"], len(remaining_idx), len(removed_idx), sc["fid_value"], float(st["loss"].item()),
                               st["train_events"][0].elapsed_time(st["train_events"][1]) / 1e3, ev0.elapsed_time(ev1) / 1e3,
                               st["steps"], sorted(set(int(self.dataset.targets[i]) for i in remaining_idx)),
                               sc["is"], sc["precision"], sc["recall"])

    def run_coalition(self, removal_seed: int, verbose=False) -> CoalitionRecord:
        """One coalition, its two phases back to back on the current stream."""
        rec = _drain(self.sample_phase(_drain(self.train_phase(removal_seed))))
        if verbose:
            print(f"[coalition {removal_seed}] |S|={rec.n_remaining} train {rec.total_steps_time:.1f}s "
                  f"sample+score {rec.total_sampling_time:.1f}s fid {rec.fid_value:.4f}", flush=True)
        return rec

    def run_pipelined(self, seeds: Sequence[int], on_record=None, on_error=None, verbose=False, n_train: int = 2) -> List[CoalitionRecord]:
        """SEVERAL coalitions in flight on this GPU: while coalition i samples (10 240 images x 100 DDIM steps in launches of 1024
        images: long kernels that fill the chip) the next coalitions fine-tune (B = 128: short launches that leave CUs idle), each
        phase on its own HIP stream - the training steps' small launches fill the tails of the sampler's large ones and of each
        other (profiles/r04_two_streams.txt: 91.6 -> 81.7 ms per 1/1000 coalition with one training stream, -> ~80 with two).
        One host thread enqueues everything: per turn one DDIM step and one optimizer step (the training phases, up to
        `n_train`, take turns; a new one starts when the others are past their first half, so that trained models arrive at the
        rate the sampler consumes them), never more than `ahead` turns ahead of the GPU.
        Every coalition computes exactly what `run_coalition` computes - same kernels, same order on its stream, its own
        workspace (ops.workspace is per stream), its own device generator seeded like `seed_everything` seeds the default one;
        the preview draw of the sampling phase (unlearn.py:761-765: its images are discarded) takes a generator of its own -
        records are bit-identical to the sequential run's (tests/test_gpu_engine.py::test_pipelined_coalitions_equal_sequential).
        `on_record(rec)` is called as each coalition finishes (durable append); a coalition whose phase raises is reported
        through `on_error(seed, exc)` and dropped, the others carry on."""
        dev = self.device
        torch.cuda.synchronize(dev)
        if getattr(self, "_streams", None) is None or len(self._streams) != n_train + 1:
            # one set per engine: ops.workspace keeps 1 GiB of scratch per stream
            self._streams = tuple(torch.cuda.Stream(dev) for _ in range(n_train + 1))
        s_samp, s_train = self._streams[0], self._streams[1:]
        todo = list(seeds)
        recs, ahead, marks = [], 8, []
        slots = [None] * n_train              # running training phases: dict(gen, seed, done)
        ready = []                            # trained states waiting for the sampler (FIFO)
        samp, samp_seed, last = None, None, -1

        def fail(seed, exc):
            if on_error is None:
                raise exc
            on_error(seed, exc)
        while todo or any(slots) or samp is not None or ready:
            if samp is None and ready:
                st = ready.pop(0)
                s_samp.wait_stream(st.pop("stream"))                      # the EMA weights were written on that training stream
                samp_seed = st["removal_seed"]
                g = torch.Generator(device=dev).manual_seed(1_000_003 * self.opt_seed + samp_seed)
                with torch.cuda.stream(s_samp):
                    samp = self.sample_phase(st, preview_generator=g)
            running = [t for t in slots if t]
            if todo and len(running) + len(ready) < n_train and all(2 * t["done"] >= self.gd_steps for t in running):
                k = slots.index(None)
                seed = todo.pop(0)
                with torch.cuda.stream(s_train[k]):
                    slots[k] = dict(gen=self.train_phase(seed, own_generator=True), seed=seed, done=0)
            order = [(last + 1 + i) % n_train for i in range(n_train)]
            k = next((i for i in order if slots[i]), None)
            if k is not None:                                             # one optimizer step, the training phases taking turns
                last, t = k, slots[k]
                try:
                    with torch.cuda.stream(s_train[k]):
                        next(t["gen"])
                    t["done"] += 1
                except StopIteration as e:
                    slots[k] = None
                    e.value["stream"] = s_train[k]
                    ready.append(e.value)
                except Exception as e:                                    # this coalition only
                    slots[k] = None
                    fail(t["seed"], e)
            if samp is not None:
                try:
                    with torch.cuda.stream(s_samp):
                        next(samp)
                except StopIteration as e:
                    samp = None
                    recs.append(e.value)
                    if verbose:
                        r = e.value
                        print(f"[coalition {r.removal_seed}] |S|={r.n_remaining} train {r.total_steps_time:.1f}s sample+score "
                              f"{r.total_sampling_time:.1f}s (each beside other phases) fid {r.fid_value:.4f}", flush=True)
                    if on_record is not None:
                        on_record(e.value)
                except Exception as e:
                    samp = None
                    fail(samp_seed, e)
            # bounded run-ahead: wait for the turn `ahead` turns back on every stream
            ev = tuple(torch.cuda.Event() for _ in self._streams)
            for e_, st_ in zip(ev, self._streams):
                e_.record(st_)
            marks.append(ev)
            if len(marks) > ahead:
                for e_ in marks.pop(0):
                    e_.synchronize()
        torch.cuda.synchronize(dev)
        return recs

    def jsonl_row(self, rec: CoalitionRecord, extra: Optional[dict] = None) -> dict:
        """Keys lds.py reads (lds.py:203-257): dataset, removal_dist, method, exp_name, removal_seed,
        remaining_idx, fid_value, gd_steps, total_steps_time, total_sampling_time."""
        from .scoring import extractor_tag
        remaining_idx, removed_idx = self.coalition(rec.removal_seed)
        row = dict(dataset=self.dataset_name, method="gd", removal_dist="shapley", removal_seed=rec.removal_seed,
                   datamodel_alpha=None, exp_name=f"gd_shapley_seed_{rec.removal_seed}", gd_steps=self.gd_steps,
                   opt_seed=self.opt_seed, n_samples=self.n_samples, num_inference_steps=self.num_inference_steps,
                   model_behavior="global", fid_value=rec.fid_value, precision=rec.precision, recall=rec.recall,
                   total_steps_time=rec.total_steps_time, **{"is": rec.inception_score},
                   trained_steps=rec.trained_steps, remaining_idx=np.asarray(remaining_idx).tolist(),
                   removed_idx=np.asarray(removed_idx).tolist(), device=str(self.device),
                   total_sampling_time=rec.total_sampling_time,
                   feature_extractor=extractor_tag(self.feature_net))   # stand-in rows must never pass for Inception rows
        if extra:
            row.update(extra)
        return row


def shard_seeds(seeds: Sequence[int], rank: int, world: int) -> List[int]:
    """Static partition: rank r owns the coalitions with removal_seed % world == r."""
    return [s for s in seeds if s % world == rank]


def gather_records(local: List[torch.Tensor], width: int, device, group=None) -> List[torch.Tensor]:
    """The single data-path collective: all_gather of [max_local, width] float64 records
    (padded with seed = -1).  Works for nccl(=RCCL) and gloo."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n_local = torch.tensor([len(local)], device=device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    cap = max(int(c.item()) for c in counts)
    buf = torch.full((max(cap, 1), width), -1.0, dtype=torch.float64, device=device)
    for i, v in enumerate(local):
        buf[i] = v.to(device)
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    out = []
    for b, c in zip(bufs, counts):
        out += [b[i].cpu() for i in range(int(c.item()))]
    return sorted(out, key=lambda v: v[0].item())


def _rank_shard(db_path: str, rank: int) -> str:
    return f"{db_path}.rank{rank}"


def _read_rows(path: str, bad: Optional[list] = None) -> List[dict]:
    """Rows of a jsonl file that carry a `removal_seed`; lines that do not parse (the torn last line of a killed writer:
    that seed is simply redone) or have no seed are counted into `bad`."""
    rows = []
    if os.path.exists(path):
        with open(path) as f:
            for line in f:
                line = line.strip()
                if not line:
                    continue
                try:
                    row = json.loads(line)
                    int(row["removal_seed"])
                except (ValueError, KeyError, TypeError):
                    if bad is not None:
                        bad.append(line)
                    continue
                rows.append(row)
    return rows


def _append_row(path: str, row: dict):
    """One finished coalition = one durable line (the reference's `open(args.db, "a+")` per job, unlearn.py:960-968)."""
    with open(path, "a+") as f:
        f.write(json.dumps(row, default=str) + "\n")
        f.flush()
        os.fsync(f.fileno())


def _shard_paths(db_path: str) -> List[str]:
    import glob
    return sorted(glob.glob(glob.escape(db_path) + ".rank*[0-9]"))


def finished_seeds(db_path: Optional[str]) -> set:
    """Seeds with a row in the db or in any rank's shard (shards of a run that died before its merge count)."""
    done = set()
    if db_path:
        for path in [db_path] + _shard_paths(db_path):
            done.update(int(r["removal_seed"]) for r in _read_rows(path))
    return done


def merge_shards(db_path: str, extra_rows: Sequence[dict] = (), keep_ranks: Sequence[int] = ()) -> List[int]:
    """Rank 0's consolidation: rows of every rank shard and of `extra_rows` (records that arrived through the
    all_gather) that are not in the db yet are appended in seed order; then the shards are removed.  A row the owning
    rank wrote itself (its shard) wins over one rebuilt from the gathered scalars.  A shard is renamed before it is read
    (a writer that is still alive re-creates `<db>.rank<r>` and loses nothing); shards of `keep_ranks` - ranks that did
    not reach the rendezvous and are not known to be dead - are read but left in place.  Returns the seeds appended."""
    have = {int(r["removal_seed"]) for r in _read_rows(db_path)}
    keep_paths = {_rank_shard(db_path, r) for r in keep_ranks}
    new, consumed = {}, []
    for path in _shard_paths(db_path):
        src = path
        if path not in keep_paths:
            src = f"{path}.merging{os.getpid()}"
            os.rename(path, src)
        bad = []
        for r in _read_rows(src, bad):
            new.setdefault(int(r["removal_seed"]), r)
        if bad:                                            # never drop lines this function could not account for
            print(f"[merge_shards] {path}: {len(bad)} unreadable line(s) kept in {src}", flush=True)
        elif src != path:
            consumed.append(src)
    for r in extra_rows:
        new.setdefault(int(r["removal_seed"]), r)
    seeds = sorted(s for s in new if s not in have)
    for s in seeds:
        _append_row(db_path, new[s])
    for src in consumed:
        os.remove(src)
    return seeds


def _rendezvous(db_path: Optional[str], tag: str, timeout_s: float):
    """Rendezvous ahead of the final collective that cannot hang on a dead rank and cannot split the ranks: every rank
    posts an arrival key in the process group's store; RANK 0 ALONE decides - it waits for every arrival, not waiting
    for a rank the launcher has declared dead (tombstone `<db>.rank<r>.dead`, written by gad.launch when a child exits
    non-zero) nor past `timeout_s` from its own arrival - and publishes the decision under `gad/<tag>/go` ("gather" or
    "skip:<missing ranks>"); every other rank polls that one key and has NO deadline of its own: it leaves only when rank
    0 is known to be dead (tombstone) or the store it hosts is gone, and then says so first (`gad/<tag>/<r>/left`), which
    rank 0 - should it be alive after all - reads before it decides, counting that rank as missing.  (A deadline of their
    own let a late rank 0 publish "gather" to peers that had already given up, and then sit alone in the collective.)
    The store is polled with the non-blocking `check` (False = not there yet; an exception = the store is gone, i.e. rank
    0, which hosts it, died: skip the collective) and a sleep between polls.  Returns (gather: bool, missing ranks)."""
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    store = dist.distributed_c10d._get_default_store()

    def dead(r):
        return bool(db_path) and os.path.exists(f"{_rank_shard(db_path, r)}.dead")

    t0 = time.time()
    try:
        store.set(f"gad/{tag}/{rank}", "1")
        if rank == 0:
            missing = []
            for r in range(1, world):
                while not store.check([f"gad/{tag}/{r}"]):
                    if dead(r) or time.time() - t0 > timeout_s:
                        missing.append(r)
                        break
                    time.sleep(0.2)
            missing += [r for r in range(1, world) if r not in missing and store.check([f"gad/{tag}/{r}/left"])]
            store.set(f"gad/{tag}/go", "gather" if not missing else "skip:" + ",".join(map(str, sorted(missing))))
            return not missing, sorted(missing)
        while not store.check([f"gad/{tag}/go"]):
            if dead(0):
                store.set(f"gad/{tag}/{rank}/left", "1")
                return False, [0]
            time.sleep(0.2)
        go = store.get(f"gad/{tag}/go").decode()
    except (RuntimeError, OSError) as e:                            # DistStoreError / DistNetworkError: the store's host is gone
        print(f"[rank {rank}] rendezvous store unreachable ({type(e).__name__}: {e}): skipping the collective", flush=True)
        return False, [0]
    if go == "gather":
        return True, []
    return False, [int(r) for r in go.split(":", 1)[1].split(",")]


def run_sharded(engine, seeds: Sequence[int], db_path: Optional[str] = None, verbose=False, retries: int = 1,
                rendezvous_timeout_s: Optional[float] = None):
    """The one-coalition-per-GPU scheduler that replaces the reference's SLURM job array
    (text_to_image/experiments/setup_unlearn_commands.py:133-154,160-214; unlearn.job:15,17 `--requeue`,
    `--open-mode=append`).

    * rank r of the default process group (or a lone process) owns the seeds with ``seed % world == r`` that have no
      row yet - neither in the db nor in any rank's shard (idempotent re-entry);
    * every finished coalition is appended at once, flushed and fsynced, to this rank's shard ``<db>.rank<r>``: a
      fault, OOM or kill on any rank loses at most the coalition that rank was running;
    * a coalition that raises is recorded in ``<db>.failed`` (seed, rank, error, traceback) and retried `retries`
      times after the rank's other seeds, with freshly built model / optimizer state (run_coalition builds them per
      call); what still fails is left for the next entry (a requeued launch);
    * the only data-path collective is one all_gather of the fixed-size records at the end, entered only when every
      rank is known to have arrived - decided once, by rank 0, and published to all (`_rendezvous`); rank 0 then appends
      the rows to the db in seed order and removes the shards.  If a rank is missing, every survivor skips the collective
      and rank 0 merges from the shards (a missing rank that is not known to be dead keeps its shard).
    Returns the records every rank knows at the end (all of this run's when the gather ran, else its own)."""
    import traceback
    import torch.distributed as dist

    dist_on = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist_on else (0, 1)
    if db_path and os.path.dirname(db_path):
        os.makedirs(os.path.dirname(db_path), exist_ok=True)     # a fresh --db under a directory that does not exist yet
    done = finished_seeds(db_path)
    mine = shard_seeds([s for s in seeds if s not in done], rank, world)
    shard = _rank_shard(db_path, rank) if db_path else None
    recs, failed = [], []
    todo, attempt = list(mine), 0

    def note_failure(s, e, tb):
        if db_path:
            _append_row(db_path + ".failed", dict(removal_seed=s, rank=rank, attempt=attempt, error=repr(e), traceback=tb))
        if verbose:
            print(f"[rank {rank}] coalition {s} failed (attempt {attempt}): {e!r}", flush=True)
    if todo and getattr(engine, "in_flight", 1) > 1 and hasattr(engine, "run_pipelined"):
        # two coalitions in flight per GPU (CoalitionEngine.run_pipelined): rows are appended as coalitions finish; one that
        # raises is recorded and goes to the sequential retry loop below
        again = []

        def on_record(rec):
            recs.append(rec)
            if shard:
                _append_row(shard, engine.jsonl_row(rec))

        def on_error(s, e):
            note_failure(s, e, "".join(traceback.format_exception(type(e), e, e.__traceback__)))
            again.append(s)
        engine.run_pipelined(todo, on_record=on_record, on_error=on_error, verbose=verbose, n_train=engine.in_flight - 1)
        attempt, todo, failed = 1, (again if retries >= 1 else []), again
    while todo:
        again = []
        for s in todo:
            try:
                rec = engine.run_coalition(s, verbose=verbose)
            except Exception as e:                               # this coalition only; the rank carries on
                note_failure(s, e, traceback.format_exc())
                again.append(s)
                continue
            recs.append(rec)
            if shard:
                _append_row(shard, engine.jsonl_row(rec))
        attempt += 1
        todo = again if attempt <= retries else []
        failed = again
    n_extra = len(getattr(engine, "extra_keys", ()))
    width = CoalitionRecord.NSCALAR + n_extra + engine.n_groups
    packed = [r.pack(engine.n_groups, n_extra) for r in recs]
    gathered, missing = True, []
    if dist_on:
        timeout = rendezvous_timeout_s if rendezvous_timeout_s is not None else float(os.environ.get("GAD_SHARD_TIMEOUT", 2 * 3600))
        _RDV[0] += 1
        gathered, missing = _rendezvous(db_path, f"sharded{_RDV[0]}", timeout)
        if not gathered:
            print(f"[rank {rank}] ranks {missing} did not reach the final all_gather: merging from the rank shards",
                  flush=True)
        else:
            packed = gather_records(packed, width, engine.device if dist.get_backend() == "nccl" else "cpu")
    all_recs = [CoalitionRecord.unpack(v, n_extra) for v in packed]
    if rank == 0 and db_path:
        # rows of the gathered records where this rank can rebuild them from the scalars (the CIFAR engine); an engine whose
        # rows carry more than the record (entry-point cycles: per-image behaviours, args) answers None for coalitions it
        # did not run itself - those rows come from the owning rank's shard
        rows = [engine.jsonl_row(r) for r in all_recs] if (gathered and dist_on) else []
        maybe_alive = [r for r in (missing if dist_on else []) if not os.path.exists(f"{_rank_shard(db_path, r)}.dead")]
        merge_shards(db_path, [r for r in rows if r is not None], keep_ranks=maybe_alive)
    if failed and verbose:
        print(f"[rank {rank}] seeds still failing after {retries} retries: {failed}", flush=True)
    return all_recs


_RDV = [0]
