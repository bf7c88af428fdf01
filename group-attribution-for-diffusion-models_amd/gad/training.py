"""EMAModel (diffusers.training_utils API used at main.py:342-351,424,725 and
src/diffusion_utils.py:193-198) and the fused training step
(reference hot loop unconditional_generation/main.py:681-725 / unlearn.py:588-636)."""
from __future__ import annotations

import torch

from . import ops


SLOT = 8      # slices start at multiples of 8 elements: 32-B aligned in fp32, 16-B aligned in a bf16 shadow of the buffer


def _slot(n):
    return (n + SLOT - 1) // SLOT * SLOT


def flatten_params(params):
    """Re-home the given parameters (and their .grad) in one contiguous fp32 buffer each, keeping conv weights'
    [Cout,KH,KW,Cin] storage.  Every slice starts 32-B aligned (so the same offsets are 16-B aligned in a bf16 copy of
    the whole buffer: `ops.bf16_weight`).  Returns (flat, flat_grad)."""
    params = list(params)
    dev = params[0].device
    sizes = [_slot(p.numel()) for p in params]
    flat = torch.zeros(sum(sizes), device=dev, dtype=torch.float32)
    gflat = torch.zeros_like(flat)
    off = 0
    with torch.no_grad():
        for p, sz in zip(params, sizes):
            n = p.numel()

            def view_like(buf, p=p, off=off, n=n):
                if p.ndim == 4:
                    o, i, kh, kw = p.shape
                    return buf[off:off + n].view(o, kh, kw, i).permute(0, 3, 1, 2)
                return buf[off:off + n].view(p.shape)

            v = view_like(flat)
            v.copy_(p)
            p.data = v
            p.grad = None
            p._gad_sink = view_like(gflat)               # gradient kernels write here directly (ops._sink)
            p._gad_sink_epoch = -1
            p._gad_flat = (flat, off, n)                 # where the parameter lives (one bf16 cast serves all of them)
            off += sz
    flat._gad_params = [(p, p._gad_flat[1], p._gad_flat[2]) for p in params]     # ops.bf16_weight snapshots their versions
    return flat, gflat


def flat_views(params, buf):
    """Logical-shape views of `buf` (a flat buffer laid out by flatten_params) for each parameter, in order."""
    off = 0
    for p in params:
        n = p.numel()
        if p.ndim == 4:
            o, i, kh, kw = p.shape
            yield buf[off:off + n].view(o, kh, kw, i).permute(0, 3, 1, 2)
        else:
            yield buf[off:off + n].view(p.shape)
        off += _slot(n)


def adam_state_to_torch(params, m, v, step, hp):
    """Flat Adam moments -> the dict ``torch.optim.Adam(W).state_dict()`` produces (the `optimizer` entry of the
    reference's ckpt_steps_*.pt, main.py:827-840), so either side can resume the other's checkpoint."""
    params = list(params)
    state = {i: {"step": torch.tensor(float(step)), "exp_avg": a.detach().cpu().contiguous(),
                 "exp_avg_sq": b.detach().cpu().contiguous()}
             for i, (a, b) in enumerate(zip(flat_views(params, m), flat_views(params, v)))} if step > 0 else {}
    group = {"lr": hp["lr"], "betas": tuple(hp["betas"]), "eps": hp["eps"], "weight_decay": hp["weight_decay"],
             "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
             "fused": None, "params": list(range(len(params)))}
    return {"state": state, "param_groups": [group]}


def adam_state_from_torch(sd, params, m, v):
    """Inverse of adam_state_to_torch: fills the flat m / v in place and returns the step count.  Also accepts the
    flat layout earlier builds of this engine wrote ({"step","exp_avg","exp_avg_sq"})."""
    if "state" not in sd:
        m.copy_(sd["exp_avg"].to(m.device))
        v.copy_(sd["exp_avg_sq"].to(v.device))
        return int(sd["step"])
    params = list(params)
    ids = [i for g in sd["param_groups"] for i in g["params"]]
    if len(ids) != len(params):
        raise ValueError(f"optimizer state covers {len(ids)} parameters, the model has {len(params)}")
    step = 0
    m.zero_()
    v.zero_()
    for i, a, b, p in zip(ids, flat_views(params, m), flat_views(params, v), params):
        st = sd["state"].get(i)
        if st is None:
            continue
        if tuple(st["exp_avg"].shape) != tuple(p.shape):
            raise ValueError(f"optimizer state {i}: shape {tuple(st['exp_avg'].shape)} != parameter {tuple(p.shape)}")
        a.copy_(st["exp_avg"].to(m.device))
        b.copy_(st["exp_avg_sq"].to(v.device))
        step = max(step, int(float(st["step"])))
    return step


def lr_lambda(name, num_training_steps, num_warmup_steps=0):
    """diffusers.optimization.get_scheduler multipliers used by the reference: "constant" (main.py) and
    "cosine" (text_to_image LoRA trainer, ddpm_config.py:637); SURVEY A.12."""
    import math
    if name == "constant":
        return lambda step: 1.0
    if name == "cosine":
        def f(step):
            if step < num_warmup_steps:
                return step / max(1, num_warmup_steps)
            prog = (step - num_warmup_steps) / max(1, num_training_steps - num_warmup_steps)
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
        return f
    raise NotImplementedError(f"lr_scheduler={name}")


class EMAModel:
    def __init__(self, parameters, decay=0.9999, min_decay=0.0, update_after_step=0, use_ema_warmup=False,
                 inv_gamma=1.0, power=2 / 3, model_cls=None, model_config=None, **unused):
        parameters = list(parameters)
        self.shadow_params = [p.clone().detach() for p in parameters]
        self.temp_stored_params = None
        self.decay, self.min_decay, self.update_after_step = decay, min_decay, update_after_step
        self.use_ema_warmup, self.inv_gamma, self.power = use_ema_warmup, inv_gamma, power
        self.optimization_step = 0
        self.cur_decay_value = None
        self.model_cls, self.model_config = model_cls, model_config
        self._flat = None

    def get_decay(self, optimization_step):
        step = max(0, optimization_step - self.update_after_step - 1)
        if step <= 0:
            return 0.0
        cur = 1 - (1 + step / self.inv_gamma) ** -self.power if self.use_ema_warmup else (1 + step) / (10 + step)
        return max(min(cur, self.decay), self.min_decay)

    def next_decay(self):
        """Advance the step counter and return the decay of this update (used by the fused optimizer)."""
        self.optimization_step += 1
        self.cur_decay_value = self.get_decay(self.optimization_step)
        return self.cur_decay_value

    def bind_flat(self, model):
        """Re-home the shadow parameters in one flat buffer laid out like model.flat so the EMA update
        fuses into the optimizer pass."""
        flat, _ = model.flat
        sflat = torch.empty_like(flat)
        off = 0
        new = []
        for s, p in zip(self.shadow_params, model.parameters()):
            n = p.numel()
            if p.ndim == 4:
                o, i, kh, kw = p.shape
                v = sflat[off:off + n].view(o, kh, kw, i).permute(0, 3, 1, 2)
            else:
                v = sflat[off:off + n].view(p.shape)
            v.copy_(s.to(v.device))
            new.append(v)
            off += _slot(n)
        self.shadow_params = new
        self._flat = sflat
        return sflat

    @torch.no_grad()
    def step(self, parameters):
        parameters = list(parameters)
        decay = self.next_decay()
        for s, p in zip(self.shadow_params, parameters):
            if s.device != p.device:
                raise ops._capi.GadError("EMAModel.step: shadow and model parameters live on different devices; "
                                         "call ema_model.to(device) first")
            if p.requires_grad:
                if p.is_cuda:
                    if s.is_contiguous() and p.is_contiguous():
                        ops.ema_update_raw(s, p.detach(), decay)
                    else:   # channels_last conv weights: same storage order on both sides
                        ops.ema_update_raw(s.permute(0, 2, 3, 1), p.detach().permute(0, 2, 3, 1), decay)
                else:
                    s.sub_((1 - decay) * (s - p))
            else:
                s.copy_(p)

    def copy_to(self, parameters):
        for s, p in zip(self.shadow_params, list(parameters)):
            p.data.copy_(s.to(p.device).data)
        ops.WEIGHT_EPOCH[0] += 1          # p.data.copy_ bumps no version counter: weight-derived caches must refresh

    def store(self, parameters):
        self.temp_stored_params = [p.detach().cpu().clone() for p in parameters]

    def restore(self, parameters):
        if self.temp_stored_params is None:
            raise RuntimeError("This ExponentialMovingAverage has no `store()`ed weights to `restore()`")
        for c, p in zip(self.temp_stored_params, parameters):
            p.data.copy_(c.data)
        self.temp_stored_params = None
        ops.WEIGHT_EPOCH[0] += 1

    def to(self, device=None, dtype=None):
        self.shadow_params = [p.to(device=device, dtype=dtype) if p.is_floating_point() else p.to(device=device)
                              for p in self.shadow_params]
        self._flat = None

    def state_dict(self):
        return {"decay": self.decay, "min_decay": self.min_decay, "optimization_step": self.optimization_step,
                "update_after_step": self.update_after_step, "use_ema_warmup": self.use_ema_warmup,
                "inv_gamma": self.inv_gamma, "power": self.power,
                "shadow_params": [s.detach().clone().contiguous() for s in self.shadow_params]}

    def load_state_dict(self, sd):
        for k in ("decay", "min_decay", "optimization_step", "update_after_step", "use_ema_warmup", "inv_gamma",
                  "power"):
            setattr(self, k, sd.get(k, getattr(self, k)))
        sp = sd.get("shadow_params")
        if sp is not None:
            if len(sp) != len(self.shadow_params):
                raise ValueError("shadow_params length mismatch")
            for dst, src in zip(self.shadow_params, sp):
                dst.copy_(src.to(dst.device))


class FusedTrainer:
    """One sFT / training step = add_noise -> U-Net fwd -> MSE (+grad) -> U-Net bwd ->
    [sum g^2] -> fused clip+Adam(W)+EMA, every kernel from libgad_hip.so, no host sync.
    Mirrors the loop body at main.py:681-725 with the batch, noise and timesteps supplied by
    the caller (so the RNG policy stays in the entry point)."""

    def __init__(self, model, scheduler, ema: EMAModel | None, lr=1e-4, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, adamw=False, max_grad_norm=1.0, loss_sign=1.0, params=None, lr_schedule=None, use_graph=False):
        """`use_graph`: capture add_noise -> forward -> loss -> backward of one step into a hipGraph after `GRAPH_WARMUP` eager steps and
        replay it from then on (the optimizer launch stays eager: its step count and learning rate are host scalars).  For steps that are
        launch-bound on the host - the half-precision SD LoRA step issues ~2 900 short launches - : the same kernels in the same order,
        bit-identical results, no per-launch host work.  Shapes must not change between steps; a capture that fails falls back to eager."""
        self.model, self.scheduler, self.ema = model, scheduler, ema
        self.use_graph, self._graph, self._graph_failed = use_graph, None, False
        if params is None:                      # train everything (DDPM); else only `params` (LoRA: base frozen)
            self.flat, self.gflat = model.flatten_parameters() if model._flat is None else model.flat
            self.params = list(model.parameters())
        else:
            if ema is not None:
                raise ValueError("EMA over a parameter subset is not used by the reference's LoRA trainer")
            self.params = list(params)
            self.flat, self.gflat = flatten_params(self.params)
        self.base_lr, self.lr_schedule = lr, lr_schedule
        self._sinks_checked, self._unwritten = False, []
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.ema_flat = ema.bind_flat(model) if ema is not None else None
        self.hp = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, adamw=adamw)
        self.max_grad_norm, self.loss_sign = max_grad_norm, loss_sign
        self.step_count = 0
        self._sumsq = torch.zeros(1, device=self.flat.device)
        self.last_loss = None

    GRAPH_WARMUP = 2

    def step(self, image_nchw, noise_nchw, timesteps, *model_args, loss_weights=None):
        """`loss_weights` [B]: per-sample weights of the squared error (min-SNR-gamma weighting,
        train_text_to_image_lora.py:1276-1298): loss = mean_b w_b * mean_chw (eps_hat - eps)^2."""
        if (self.use_graph and not self._graph_failed and self.step_count >= self.GRAPH_WARMUP
                and all(torch.is_tensor(a_) for a_ in model_args)):
            out = self._step_graphed(image_nchw, noise_nchw, timesteps, model_args, loss_weights)
            if out is not None:
                return out
        return self._step_eager(image_nchw, noise_nchw, timesteps, *model_args, loss_weights=loss_weights)

    def _forward_backward(self, image_nchw, noise_nchw, timesteps, model_args, loss_weights):
        """add_noise -> U-Net -> loss (+ gradient) -> backward into the flat gradient buffer; -> loss [1]"""
        model = self.model
        noisy = self.scheduler.add_noise(image_nchw, noise_nchw, timesteps)
        eps = model(noisy, timesteps, *model_args).sample
        loss, d = ops.mse_fwd_bwd_raw(eps.contiguous(), noise_nchw.contiguous(), grad_scale=self.loss_sign)
        if loss_weights is not None:           # tiny [B,4,h,w] tensors: plain torch, off the hot path
            w = loss_weights.to(d.dtype).view(-1, 1, 1, 1)
            d = d * w
            loss = ((eps.detach() - noise_nchw).square().mean(dim=(1, 2, 3)) * w.view(-1)).mean().view(1) * self.loss_sign
        ops.begin_backward_step()        # every parameter's first gradient of this step overwrites its flat slot
        try:
            eps.backward(d)              # parameter gradients land in gflat; autograd sees None for them
        finally:
            ops.end_backward_step()
        if not self._sinks_checked:      # a parameter that got no gradient would keep a stale slot: zero it each step
            ep = ops._SINK_EPOCH[0]
            self._unwritten = [p._gad_sink for p in self.params if getattr(p, "_gad_sink_epoch", -1) != ep]
            self._sinks_checked = True
        for v in self._unwritten:
            v.zero_()
        return loss

    def _step_eager(self, image_nchw, noise_nchw, timesteps, *model_args, loss_weights=None):
        loss = self._forward_backward(image_nchw, noise_nchw, timesteps, model_args, loss_weights)
        self.optimizer_step()
        self.last_loss = loss
        return loss

    def _step_graphed(self, image_nchw, noise_nchw, timesteps, model_args, loss_weights):
        g = self._graph
        sig = (tuple(image_nchw.shape), tuple(timesteps.shape), tuple(tuple(a_.shape) for a_ in model_args), loss_weights is not None)
        if g is not None and g["sig"] != sig:
            return None                                  # shapes changed: this step runs eagerly
        if g is None:
            try:
                st = dict(sig=sig, image=torch.empty_like(image_nchw), noise=torch.empty_like(noise_nchw), t=torch.empty_like(timesteps),
                          args=[torch.empty_like(a_) for a_ in model_args],
                          w=torch.empty_like(loss_weights) if loss_weights is not None else None)
                for dst, src in ((st["image"], image_nchw), (st["noise"], noise_nchw), (st["t"], timesteps), *zip(st["args"], model_args)):
                    dst.copy_(src)
                if st["w"] is not None:
                    st["w"].copy_(loss_weights)
                # derived weights (bf16 shadows of the LoRA buffer, ...) are refreshed by launches that key on the buffer's epoch: bump it so
                # that the refresh is captured - a replay must re-derive them from the CURRENT weights
                self.flat._gad_epoch = getattr(self.flat, "_gad_epoch", 0) + 1
                # scratch of the recorded launches: this graph's own (two trainers' graphs may replay on two streams at once)
                st["ws"] = torch.empty(ops.WS_BYTES, dtype=torch.uint8, device=self.flat.device)
                graph = torch.cuda.CUDAGraph()
                prev, ops.WS_OVERRIDE[0] = ops.WS_OVERRIDE[0], st["ws"]
                try:
                    with torch.cuda.graph(graph):
                        st["loss"] = self._forward_backward(st["image"], st["noise"], st["t"], st["args"], st["w"])
                finally:
                    ops.WS_OVERRIDE[0] = prev
                st["graph"] = graph
                self._graph = g = st
                # (the capture recorded the launches without running them: this step is done by the first replay below)
            except Exception as e:                       # noqa: BLE001 - any capture failure means "run eagerly", never a lost step
                self._graph_failed = True
                self._graph = None
                import warnings
                warnings.warn(f"FusedTrainer: hipGraph capture of the training step failed ({type(e).__name__}: {e}); running eagerly")
                torch.cuda.synchronize()
                return None
        else:
            g["image"].copy_(image_nchw)
            g["noise"].copy_(noise_nchw)
            g["t"].copy_(timesteps)
            for dst, src in zip(g["args"], model_args):
                dst.copy_(src)
            if g["w"] is not None:
                g["w"].copy_(loss_weights)
        g["graph"].replay()
        self.optimizer_step()
        self.last_loss = g["loss"].clone()          # (the static tensor is overwritten by the next replay)
        return self.last_loss

    def optimizer_step(self):
        self.step_count += 1
        sumsq = None
        if self.max_grad_norm is not None:
            sumsq = ops.sumsq_raw(self.gflat, out=self._sumsq)
        decay = self.ema.next_decay() if self.ema is not None else 0.0
        if self.lr_schedule is not None:        # LambdaLR semantics: step k (0-based) runs at base_lr * f(k)
            self.hp["lr"] = self.base_lr * self.lr_schedule(self.step_count - 1)
        ops.clip_adam_ema_raw(self.flat, self.gflat, self.m, self.v, self.ema_flat, sumsq,
                              max_norm=self.max_grad_norm or 0.0, step=self.step_count, ema_decay=decay, **self.hp)

    def grad_norm(self):
        return self._sumsq.sqrt()

    def state_dict(self):
        """Optimizer state in torch.optim.Adam's own state_dict format (the `optimizer` entry of
        ckpt_steps_*.pt, main.py:827-840): the reference can resume it and vice versa."""
        return adam_state_to_torch(self.params, self.m, self.v, self.step_count, self.hp)

    def load_state_dict(self, sd):
        self.step_count = adam_state_from_torch(sd, self.params, self.m, self.v)
