"""Structural magnitude pruning of a trained UNet2DModel (reference unconditional_generation/prune.py:318-421),
torch_pruning-free.

The reference hands the model to torch_pruning 1.3.2 (`MagnitudePruner(ch_sparsity=ratio, iterative_steps=1,
ignored_layers=[conv_out])`, :344-352) and pickles the resulting nn.Module (:416-421).  torch_pruning is not
installable offline, so the dependency analysis is restated here for the UNet2DModel topology:

  * a *channel space* is a set of tensor dimensions that must keep the same channels: the residual stream of a
    stage (conv_in / conv2 / conv_shortcut / to_out outputs joined by identity shortcuts and attention
    residuals), each ResnetBlock2D's internal width (conv1 out + time_emb_proj out + norm2 + conv2 in), the
    attention q/k width, the attention v width, the two time-embedding widths; consumers (norm1, conv1 in,
    shortcut in, q/k/v in, down/upsampler in, the slices of the skip concatenation) follow their producer;
  * importance of a channel = mean over the space's conv / linear weights of the squared L2 norm of the
    channel's slice, normalised by the space's mean (MagnitudeImportance p=2, group_reduction="mean",
    normalizer="mean");
  * spaces that pass through a GroupNorm lose the same number of channels from every norm group, so the
    widths stay multiples of the group count (128 -> 96, 256 -> 192 at ratio 0.3).
`--pruner taylor | diff-pruning` (reference :320-332, :358-378): first-order Taylor importance |sum w g| / sum |w g| from
gradients accumulated over the training timesteps on one batch, through the same channel spaces and selection rule.
Deviation kept on purpose: spaces without a GroupNorm (time embedding, attention q/k and v) are cut to the width
the model constructor derives from `block_out_channels` (4 x width0, stage width) rather than to
int(w * (1 - ratio)), so the pruned network is described by a plain `unet_config` and its state_dict - the
checkpoint stores those two instead of a pickled module.  Parity-unpinned."""
import argparse
import os
import sys

import numpy as np
import torch

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import src.constants as constants  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Pruning diffusion models")
    p.add_argument("--load", type=str, default=None, help="directory of the pre-trained model checkpoints")
    p.add_argument("--dataset", type=str, default=None, choices=constants.DATASET + ["toy2"])
    p.add_argument("--outdir", type=str, default=constants.OUTDIR)
    p.add_argument("--opt_seed", type=int, default=42)
    p.add_argument("--pruning_ratio", type=float, default=0.3)
    p.add_argument("--pruner", type=str, default="magnitude", choices=["magnitude", "random", "reinit", "taylor", "diff-pruning"])
    p.add_argument("--batch_size", type=int, default=None, help="batch of the Taylor gradient accumulation (default: the config's)")
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--thr", type=float, default=0.05)
    p.add_argument("--trained_steps", type=int, default=None)
    return p.parse_args(argv)


def pruned_width(w, ratio, groups):
    """Channels kept by MagnitudePruner for a space that passes through GroupNorm(groups)."""
    n_pruned = w - int(w * (1 - ratio))
    return w - (n_pruned // groups) * groups


def pruned_head_dim(c, head_dim, ratio):
    """Head dim a [heads * head_dim]-wide q / k / v projection keeps: the pruner removes int-truncated `ratio` of the
    channels, the same number - and the same in-head positions - from every head."""
    heads = c // head_dim
    n_pruned = c - int(c * (1 - ratio))
    return head_dim - n_pruned // heads


class Space:
    """One tied channel dimension: `members` = (param name, dim, offset) with `width` channels each."""

    def __init__(self, name, width, gn_groups=None, target=None, heads=None):
        self.name, self.width, self.gn, self.target = name, width, gn_groups, target
        self.heads = heads         # head-grouped space: the same in-head channels are kept in every head
        self.members = []          # every tensor dim that is sliced with this space's kept indices
        self.scored = []           # subset of members that are conv / linear weights (importance)
        self.keep = None

    def add(self, param, dim, offset=0, score=False):
        self.members.append((param, dim, offset))
        if score:
            self.scored.append((param, dim, offset))


def build_spaces(cfg, ratio=0.0):
    """Walk the UNet2DModel topology and return the list of channel spaces, each with the width
    it keeps at `ratio`."""
    boc = list(cfg["block_out_channels"])
    G = cfg.get("norm_num_groups", 32)
    lpb = cfg.get("layers_per_block", 2)
    spaces = []

    def new(name, width, gn=True, target=None, heads=None):
        s = Space(name, width, G if gn else None, target if target is not None else pruned_width(width, ratio, G), heads)
        spaces.append(s)
        return s

    temb_target = 4 * pruned_width(boc[0], ratio, G)          # time_embed_dim = 4 * block_out_channels[0]
    t1 = new("temb_hidden", boc[0] * 4, gn=False, target=temb_target)
    t2 = new("temb_out", boc[0] * 4, gn=False, target=temb_target)
    for suffix in ("weight", "bias"):
        t1.add(f"time_embedding.linear_1.{suffix}", 0, score=suffix == "weight")
        t2.add(f"time_embedding.linear_2.{suffix}", 0, score=suffix == "weight")
    t1.add("time_embedding.linear_2.weight", 1, score=True)
    # the sinusoidal projection has block_out_channels[0] features: linear_1's input follows the new width
    # (the constructor re-derives the frequencies for the narrower embedding)
    t0 = new("temb_in", boc[0], gn=False, target=pruned_width(boc[0], ratio, G))
    t0.add("time_embedding.linear_1.weight", 1, score=True)

    def produce(space, prefix):                      # conv / linear output channels + bias
        space.add(prefix + ".weight", 0, score=True)
        space.add(prefix + ".bias", 0)

    def consume(parts, prefix, norm=False):          # parts: [(space, offset)] along the input-channel dim
        for sp, off in parts:
            if norm:
                sp.add(prefix + ".weight", 0, off)
                sp.add(prefix + ".bias", 0, off)
            else:
                sp.add(prefix + ".weight", 1, off, score=True)

    def resnet(prefix, parts, cin, cout):
        consume(parts, prefix + ".norm1", norm=True)
        consume(parts, prefix + ".conv1")
        inner = new(prefix + ".inner", cout)
        produce(inner, prefix + ".conv1")
        produce(inner, prefix + ".time_emb_proj")
        t2.add(prefix + ".time_emb_proj.weight", 1, score=True)
        consume([(inner, 0)], prefix + ".norm2", norm=True)
        consume([(inner, 0)], prefix + ".conv2")
        if cin != cout:
            out = new(prefix + ".out", cout)
            consume(parts, prefix + ".conv_shortcut")
            produce(out, prefix + ".conv_shortcut")
        else:
            assert len(parts) == 1
            out = parts[0][0]
        produce(out, prefix + ".conv2")
        return out

    head_dim = cfg.get("attention_head_dim")

    def attention(prefix, stream, c):
        consume([(stream, 0)], prefix + ".group_norm", norm=True)
        if head_dim is None:                         # one head of the full width (CIFAR): plain magnitude pruning
            qk, v = new(prefix + ".qk", c, gn=False), new(prefix + ".v", c, gn=False)
        else:                                        # multi-head (CelebA): channel_groups[to_q/k/v] = heads (prune.py:337-342)
            heads = c // head_dim
            tgt = heads * pruned_head_dim(c, head_dim, ratio)
            qk = new(prefix + ".qk", c, gn=False, target=tgt, heads=heads)
            v = new(prefix + ".v", c, gn=False, target=tgt, heads=heads)
        for proj, sp in (("to_q", qk), ("to_k", qk), ("to_v", v)):
            consume([(stream, 0)], f"{prefix}.{proj}")
            produce(sp, f"{prefix}.{proj}")
        consume([(v, 0)], prefix + ".to_out.0")
        produce(stream, prefix + ".to_out.0")

    cur = new("conv_in", boc[0])
    produce(cur, "conv_in")
    skips = [(cur, boc[0])]
    width = boc[0]
    for i, typ in enumerate(cfg["down_block_types"]):
        cin, cout = width, boc[i]
        for j in range(lpb):
            cur = resnet(f"down_blocks.{i}.resnets.{j}", [(cur, 0)], cin if j == 0 else cout, cout)
            if typ == "AttnDownBlock2D":
                attention(f"down_blocks.{i}.attentions.{j}", cur, cout)
            skips.append((cur, cout))
        width = cout
        if i != len(boc) - 1:
            consume([(cur, 0)], f"down_blocks.{i}.downsamplers.0.conv")
            cur = new(f"down_blocks.{i}.down", cout)
            produce(cur, f"down_blocks.{i}.downsamplers.0.conv")
            skips.append((cur, cout))
    cur = resnet("mid_block.resnets.0", [(cur, 0)], width, width)
    if cfg.get("add_attention", True):
        attention("mid_block.attentions.0", cur, width)
    cur = resnet("mid_block.resnets.1", [(cur, 0)], width, width)
    rev = list(reversed(boc))
    for i, typ in enumerate(cfg["up_block_types"]):
        cout = rev[i]
        for j in range(lpb + 1):
            sk, wk = skips.pop()
            cur = resnet(f"up_blocks.{i}.resnets.{j}", [(cur, 0), (sk, width)], width + wk, cout)
            width = cout
            if typ == "AttnUpBlock2D":
                attention(f"up_blocks.{i}.attentions.{j}", cur, cout)
        if i != len(boc) - 1:
            consume([(cur, 0)], f"up_blocks.{i}.upsamplers.0.conv")
            cur = new(f"up_blocks.{i}.up", cout)
            produce(cur, f"up_blocks.{i}.upsamplers.0.conv")
    consume([(cur, 0)], "conv_norm_out", norm=True)
    consume([(cur, 0)], "conv_out")                  # conv_out's own output channels are never pruned
    assert not skips
    return spaces


def channel_scores(space, sd, grads=None, multivariable=True):
    """Importance of the space's channels, mean over its scored tensors, normalised by its mean.
    grads is None: squared L2 magnitude (MagnitudeImportance p = 2).  Else first-order Taylor (reference prune.py:320-332):
    |sum_j w_j g_j| over a channel's slice (`--pruner taylor`, TaylorImportance(multivariable=True)) or sum_j |w_j g_j|
    (`--pruner diff-pruning`, multivariable=False)."""
    acc = []
    for name, dim, off in space.scored:
        w = sd[name].double()
        w = w.transpose(0, dim).reshape(w.shape[dim], -1)[off:off + space.width]
        if grads is None:
            acc.append(w.pow(2).sum(dim=1))
        else:
            g = grads[name].double()
            g = g.transpose(0, dim).reshape(g.shape[dim], -1)[off:off + space.width]
            acc.append((w * g).sum(dim=1).abs() if multivariable else (w * g).abs().sum(dim=1))
    s = torch.stack(acc).mean(dim=0)
    return (s / s.mean().clamp_min(1e-300)).numpy()


def select_channels(space, sd, target, mode="magnitude", rng=None, grads=None):
    if mode == "magnitude":
        score = channel_scores(space, sd)
    elif mode in ("taylor", "diff-pruning"):
        score = channel_scores(space, sd, grads, multivariable=(mode == "taylor"))
    else:
        score = rng.rand(space.width)
    if space.heads:                                  # importance summed over the heads per in-head position
        d = space.width // space.heads
        keep_d = target // space.heads
        pos = np.sort(np.argsort(-score.reshape(space.heads, d).sum(axis=0), kind="stable")[:keep_d])
        return np.concatenate([h * d + pos for h in range(space.heads)])
    if space.gn:
        cpg = space.width // space.gn
        keep_per = target // space.gn
        keep = []
        for g in range(space.gn):
            idx = np.arange(g * cpg, (g + 1) * cpg)
            order = idx[np.argsort(-score[idx], kind="stable")]
            keep += sorted(order[:keep_per].tolist())
        return np.array(keep)
    return np.sort(np.argsort(-score, kind="stable")[:target])


def taylor_gradients(model, scheduler, clean_images, noise, mode, thr, log=print):
    """Accumulated parameter gradients of the denoising loss over the timesteps t = 0, 1, 2, ... with the SAME batch and
    noise at every step (reference prune.py:358-378).  `diff-pruning` stops once the loss has fallen below thr x the
    largest loss seen so far (Taylor expansion over the pruned timesteps only, L_t / L_max > thr)."""
    import torch.nn.functional as F
    model.zero_grad()
    model.eval()
    loss_max, used = 0.0, 0
    for step_k in range(scheduler.config.num_train_timesteps):
        t = torch.full((clean_images.shape[0],), step_k, device=clean_images.device, dtype=torch.long)
        out = model(scheduler.add_noise(clean_images, noise, t), t).sample
        loss = F.mse_loss(out, noise)
        loss.backward()                                       # gradients accumulate across the timesteps
        used += 1
        if mode == "diff-pruning":
            lv = float(loss.detach())
            loss_max = max(loss_max, lv)
            if lv < loss_max * thr:
                break
    log(f"Accumulated gradients over {used} timesteps for pruning")
    grads = {n: (p.grad.detach().cpu().contiguous() if p.grad is not None else torch.zeros_like(p).cpu())
             for n, p in model.named_parameters()}
    model.zero_grad()
    return grads


def prune_state_dict(cfg, sd, ratio, mode="magnitude", seed=42, grads=None):
    """-> (new unet_config, new state_dict).  Widths of the new config: pruned_width(w, ratio, groups) per stage."""
    G = cfg.get("norm_num_groups", 32)
    boc = list(cfg["block_out_channels"])
    new_boc = [pruned_width(w, ratio, G) for w in boc]
    rng = np.random.RandomState(seed)
    spaces = build_spaces(cfg, ratio)
    plan = {}                                   # param -> {dim: [(offset, width, keep idx)]}
    for sp in spaces:
        sp.keep = select_channels(sp, sd, sp.target, mode, rng, grads)
        for name, dim, off in sp.members:
            plan.setdefault(name, {}).setdefault(dim, []).append((off, sp.width, sp.keep))
    out = {}
    for name, t in sd.items():
        for dim, parts in plan.get(name, {}).items():
            parts = sorted(parts, key=lambda p: p[0])
            assert sum(p[1] for p in parts) == t.shape[dim], (name, dim, t.shape, [(p[0], p[1]) for p in parts])
            idx = torch.as_tensor(np.concatenate([off + keep for off, _, keep in parts]))
            t = t.index_select(dim, idx)
        out[name] = t.contiguous()
    new_cfg = dict(cfg, block_out_channels=new_boc)
    if cfg.get("attention_head_dim") is not None:    # heads stay, the head dim shrinks: not expressible as channels // head_dim
        hd = cfg["attention_head_dim"]
        new_cfg["attention_layout"] = [[w // hd, pruned_head_dim(w, hd, ratio)] for w in boc]
    return new_cfg, out


def main(args, backend=None):
    if backend is None:
        import gad as backend
    from src.utils import get_max_steps
    steps = args.trained_steps if args.trained_steps is not None else get_max_steps(args.load)
    if steps is None:
        raise ValueError(f"No trained checkpoints found at {args.load}")
    ck = torch.load(os.path.join(args.load, f"ckpt_steps_{steps:0>8}.pt"), map_location="cpu", weights_only=False)
    from src.diffusion_utils import dataset_config
    cfg = dict(ck.get("unet_config") or dataset_config(args.dataset)["unet_config"])
    sd = {k: v.detach().cpu().contiguous() for k, v in ck["unet"].items()}
    base = sum(v.numel() for v in sd.values())
    mode = args.pruner if args.pruner in ("magnitude", "taylor", "diff-pruning") else "random"
    grads = None
    if mode in ("taylor", "diff-pruning"):
        # one batch of the training set and one noise draw, as the reference's `clean_images` / `noise` (prune.py:240-251)
        from src.datasets import create_dataset
        from src.diffusion_utils import dataset_config as _dc
        full_cfg = _dc(args.dataset)
        bs = args.batch_size or full_cfg["batch_size"]
        ds = create_dataset(dataset_name=args.dataset, train=True)
        order = torch.randperm(len(ds), generator=torch.Generator().manual_seed(args.opt_seed))[:bs].tolist()
        dev = torch.device(args.device)
        clean = torch.stack([torch.as_tensor(ds[i][0]) for i in order]).float().to(dev)
        noise = torch.randn(clean.shape, generator=torch.Generator().manual_seed(args.opt_seed + 1)).to(dev)
        dense = getattr(backend, cfg["_class_name"])(**cfg)
        dense.load_state_dict(sd)
        dense.to(dev)
        keys = ("beta_start", "beta_end", "beta_schedule", "num_train_timesteps", "trained_betas")
        sched = backend.DDPMScheduler(**{k: v for k, v in full_cfg["scheduler_config"].items() if k in keys})
        grads = taylor_gradients(dense, sched, clean, noise, mode, args.thr)
        del dense
    new_cfg, new_sd = prune_state_dict(cfg, sd, args.pruning_ratio, mode, args.opt_seed, grads)
    model = getattr(backend, new_cfg["_class_name"])(**new_cfg)
    if args.pruner == "reinit":
        new_sd = {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    model.load_state_dict(new_sd)                                   # strict: every shape must agree
    print("#Params: {:.4f} M => {:.4f} M".format(base / 1e6, sum(v.numel() for v in new_sd.values()) / 1e6))
    tag = f"pruner={args.pruner}_pruning_ratio={args.pruning_ratio}_threshold={args.thr}"
    outdir = os.path.join(args.outdir, args.dataset, "pruned", "models", tag)
    os.makedirs(outdir, exist_ok=True)
    torch.save({"unet": new_sd, "unet_config": new_cfg}, os.path.join(outdir, f"ckpt_steps_{0:0>8}.pt"))
    print(f"Checkpoint saved at {outdir}")
    return outdir


if __name__ == "__main__":
    main(parse_args())
