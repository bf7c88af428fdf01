"""GPU parity of the Stable-Diffusion pieces: LayerNorm, GEGLU, cross-attention with Tk=77,
UNet2DConditionModel forward and LoRA-only backward vs the CPU oracle (fp64/fp32 torch)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def close(got, want, rtol=2e-4, atol=2e-4):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    assert got.shape == want.shape
    assert torch.allclose(got, want, rtol=rtol, atol=atol * max(1.0, want.abs().max().item())), \
        f"max err {(got - want).abs().max().item():.3e}"


@pytest.mark.parametrize("rows,C", [(7, 320), (300, 640), (77 * 3, 1280), (5, 96)])
def test_layernorm_fwd_bwd(rows, C):
    from gad import ops
    x = (rnd(rows, C, seed=1) * 1.7 + 0.4).double().requires_grad_(True)
    g, b = (rnd(C, seed=2) * 0.2 + 1).double().requires_grad_(True), (rnd(C, seed=3) * 0.1).double().requires_grad_(True)
    y = F.layer_norm(x, (C,), g, b, 1e-5)
    dy = rnd(rows, C, seed=4)
    y.backward(dy.double())
    gx, gg, gb = (t.detach().float().to(dev).requires_grad_(True) for t in (x, g, b))
    out = ops.layer_norm(gx, gg, gb, 1e-5)
    out.backward(dy.to(dev))
    close(out, y, atol=2e-5)
    close(gx.grad, x.grad, atol=5e-5)
    close(gg.grad, g.grad, atol=2e-5 * math.sqrt(rows))
    close(gb.grad, b.grad, atol=2e-5 * math.sqrt(rows))


def test_geglu_fwd_bwd():
    from gad import ops
    h = rnd(50, 2 * 128, seed=1).double().requires_grad_(True)
    a, gate = h.chunk(2, dim=-1)
    y = a * F.gelu(gate)
    dy = rnd(50, 128, seed=2)
    y.backward(dy.double())
    gh = h.detach().float().to(dev).requires_grad_(True)
    out = ops.geglu(gh)
    out.backward(dy.to(dev))
    close(out, y, atol=1e-6)
    close(gh.grad, h.grad, atol=1e-6)


def test_cross_attention_ragged_context_length():
    from gad import ops
    B, Tq, Tk, heads, d = 2, 64, 77, 4, 40
    C = heads * d
    q, k, v = rnd(B, Tq, C, seed=1, scale=0.5), rnd(B, Tk, C, seed=2, scale=0.5), rnd(B, Tk, C, seed=3)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))

    def split(t, T):
        return t.view(B, T, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(split(qd, Tq), split(kd, Tk), split(vd, Tk)).transpose(1, 2).reshape(B, Tq, C)
    do = rnd(B, Tq, C, seed=4)
    o.backward(do.double())
    gq, gk, gv = (t.to(dev).requires_grad_(True) for t in (q, k, v))
    out = ops.attention_core(gq, gk, gv, heads)
    out.backward(do.to(dev))
    close(out, o, atol=3e-5)
    close(gq.grad, qd.grad, atol=5e-5)
    close(gk.grad, kd.grad, atol=5e-5)
    close(gv.grad, vd.grad, atol=5e-5)


SMALL = dict(block_out_channels=(64, 128, 128, 128), attention_head_dim=4, cross_attention_dim=96, sample_size=16)


def _pair(lora_rank=None):
    import gad
    from oracle.diffusers_ref import LoRALinearLayer as RL
    from oracle.sd_unet_ref import CrossAttention as RCA
    from oracle.sd_unet_ref import UNet2DConditionModel as R
    torch.manual_seed(0)
    ref = R(**SMALL)
    net = gad.UNet2DConditionModel(**SMALL)
    net.load_state_dict(ref.state_dict())
    net.to(dev)
    if lora_rank:
        ranks = {}
        i = 0
        for name, m in ref.named_modules():
            if isinstance(m, RCA):
                for proj, lin in (("to_q", m.to_q), ("to_k", m.to_k), ("to_v", m.to_v), ("to_out", m.to_out[0])):
                    r = lora_rank - (i % 3) * 2          # ragged ranks as after prune_lora.py
                    i += 1
                    layer = RL(lin.in_features, lin.out_features, rank=r)
                    with torch.no_grad():
                        layer.up.weight.copy_(rnd(lin.out_features, r, seed=100 + i, scale=0.05))
                    lin.set_lora_layer(layer)
                    ranks[f"{name}.{proj}"] = (r, layer)
        for p in ref.parameters():
            p.requires_grad_(False)
        net.inject_lora(rank=lora_rank, ranks={k: v[0] for k, v in ranks.items()})
        for name, attn in net.attention_modules().items():
            base = name[: -len(".processor")]
            for proj, lin in (("to_q", attn.to_q), ("to_k", attn.to_k), ("to_v", attn.to_v), ("to_out", attn.to_out[0])):
                src = ranks[f"{base}.{proj}"][1]
                with torch.no_grad():
                    lin.lora_layer.down.weight.copy_(src.down.weight)
                    lin.lora_layer.up.weight.copy_(src.up.weight)
                for p in src.parameters():
                    p.requires_grad_(True)
    return ref, net


def test_sd_unet_forward_matches_oracle():
    ref, net = _pair()
    x, ctx, t = rnd(2, 4, 16, 16, seed=1), rnd(2, 77, 96, seed=2), torch.tensor([5, 800])
    close(net(x.to(dev), t.to(dev), ctx.to(dev)).sample, ref(x, t, ctx).sample, atol=1e-4)


def test_sd_lora_training_step_grads(tmp_path):
    from gad import ops
    ref, net = _pair(lora_rank=8)
    x, ctx, t = rnd(2, 4, 16, 16, seed=1), rnd(2, 77, 96, seed=2), torch.tensor([5, 800])
    noise = rnd(2, 4, 16, 16, seed=3)
    want = ref(x, t, ctx).sample
    F.mse_loss(want, noise).backward()
    got = net(x.to(dev), t.to(dev), ctx.to(dev)).sample
    loss, d = ops.mse_fwd_bwd_raw(got.contiguous(), noise.to(dev))
    got.backward(d)
    close(got, want, atol=1e-4)
    rg = {n: p.grad for n, p in ref.named_parameters() if p.grad is not None}
    gg = {n: p.grad for n, p in net.named_parameters() if p.grad is not None}
    assert len(rg) == len(gg) == 32 * 4 * 2 and set(rg) == set(gg)          # only LoRA down/up get gradients
    typical = sorted(v.norm().item() for v in rg.values())[len(rg) // 2]
    for n in rg:
        err = (gg[n].cpu().double() - rg[n].double()).norm().item() / (rg[n].norm().item() + 1e-3 * typical)
        assert err < 3e-3, (n, err)
    # safetensors round trip with ragged ranks
    net.save_attn_procs(str(tmp_path))
    import gad
    net2 = gad.UNet2DConditionModel(**SMALL)
    net2.load_state_dict({k: v for k, v in ref.state_dict().items() if "lora" not in k})
    net2.to(dev)
    net2.load_attn_procs(str(tmp_path))
    close(net2(x.to(dev), t.to(dev), ctx.to(dev)).sample, got, atol=1e-6)


def test_sd_lora_entry_point(tmp_path):
    import json
    import pandas as pd
    from safetensors.torch import load_file
    from text_to_image import train_text_to_image_lora as T
    data = tmp_path / "artbench"
    T.synthetic_cache(str(data / "latent_cache.pt"), n=300, n_artists=20, res=128, ctx_dim=96)
    pd.DataFrame({"artist": sorted({f"artist_{i:03d}" for i in range(20)})}).to_csv(data / "post_impressionism_artists.csv", index=False)
    over = json.dumps(dict(block_out_channels=(64, 128, 128, 128), attention_head_dim=4, cross_attention_dim=96, sample_size=16))
    common = ["--train_data_dir", str(data), "--output_dir", str(tmp_path / "out"), "--cls_key", "style", "--cls",
              "post_impressionism", "--train_batch_size", "8", "--unet_overrides", over, "--seed", "42",
              "--lr_scheduler", "cosine", "--learning_rate", "3e-4", "--adam_weight_decay", "1e-6"]
    a = T.parse_args(common + ["--method", "retrain", "--rank", "8", "--max_train_steps", "4", "--removal_dist", "shapley",
                               "--removal_unit", "artist", "--removal_seed", "0"])
    assert T.main(a)
    mdir = tmp_path / "out" / "artbench_post_impressionism" / "retrain" / "models" / "artist_shapley" / "shapley_seed=0"
    sd = load_file(str(mdir / "pytorch_lora_weights.safetensors"))
    assert len(sd) == 32 * 4 * 2
    k = "mid_block.attentions.0.transformer_blocks.0.attn2.processor.to_k_lora.down.weight"   # unet.save_attn_procs: no `unet.` prefix
    assert sd[k].shape == (8, 96) and sd[k.replace("down", "up")].abs().sum() > 0        # up started at 0: it trained
    t = pd.read_csv(mdir / "time.csv")
    assert list(t.columns) == ["step", "time", "gpu"] and len(t) == 4
    ridx = pd.read_csv(mdir / "removal_idx.csv")
    assert set(ridx.columns) == {"idx", "remaining"} and len(ridx) == 20
    assert T.main(a) is False                                                          # skip-if-done
    # min-SNR weighting + offset noise + bf16 operands (reference flags --snr_gamma/--noise_offset/--mixed_precision)
    import gad
    try:
        b = T.parse_args(common + ["--method", "retrain", "--rank", "4", "--max_train_steps", "2", "--snr_gamma", "5.0",
                                   "--noise_offset", "0.1", "--mixed_precision", "fp16"])
        assert T.main(b) and gad.ops.OPERAND_PRECISION[0] == 2      # fp16 -> half-precision activations (gad/half.py)
    finally:
        gad.set_operand_precision("no")


def test_trainer_loss_weights_and_min_snr():
    """loss = mean_b w_b mse_b (train_text_to_image_lora.py:1276-1298): w = 1 is the unweighted step bit for bit,
    w = 2 doubles loss and gradient; the weights follow min(SNR, gamma) / SNR."""
    import gad
    from gad.schedulers import min_snr_weights
    sch = gad.DDPMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000)
    ts = torch.tensor([0, 10, 500, 999])
    a = sch.alphas_cumprod[ts].double()
    snr = a / (1 - a)
    assert torch.allclose(min_snr_weights(sch.alphas_cumprod, ts, 5.0).double(), torch.clamp(snr, max=5.0) / snr, rtol=1e-5)
    x, ctx, t = rnd(4, 4, 16, 16, seed=1).to(dev), rnd(4, 77, 96, seed=2).to(dev), ts.to(dev)
    noise = rnd(4, 4, 16, 16, seed=3).to(dev)
    res = []
    for w in (None, torch.ones(4, device=dev), 2 * torch.ones(4, device=dev)):
        _, net = _pair(lora_rank=8)
        params = [p for n, p in net.named_parameters() if "lora_layer" in n]
        tr = gad.FusedTrainer(net, sch, None, lr=0.0, max_grad_norm=None, params=params)
        loss = tr.step(x, noise, t, ctx, loss_weights=w)
        res.append((loss.item(), tr.gflat.clone()))
    assert res[0][0] == pytest.approx(res[1][0], rel=1e-6) and torch.equal(res[0][1], res[1][1])
    assert res[2][0] == pytest.approx(2 * res[0][0], rel=1e-6) and torch.allclose(res[2][1], 2 * res[0][1], rtol=1e-6, atol=0)


def test_guided_latent_sampling_matches_oracle_loop():
    import gad
    from oracle import diffusers_ref as R
    ref, net = _pair()
    cond, uncond = rnd(2, 77, 96, seed=5), rnd(1, 77, 96, seed=6).expand(2, -1, -1).contiguous()
    kw = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False,
              steps_offset=1)
    pipe = gad.StableDiffusionLatentPipeline(net, gad.DDIMScheduler(**kw))
    got = pipe(cond.to(dev), uncond.to(dev), num_inference_steps=4, guidance_scale=7.5,
               generator=torch.Generator().manual_seed(11), height=128, width=128).latents
    sch = R.DDIMScheduler(**kw)
    sch.set_timesteps(4)
    assert sch.timesteps.tolist() == [751, 501, 251, 1]
    x = torch.randn((2, 4, 16, 16), generator=torch.Generator().manual_seed(11))
    with torch.no_grad():
        for t in sch.timesteps:
            e = ref(torch.cat([x, x]), t, torch.cat([uncond, cond])).sample
            eu, ec = e.chunk(2)
            x = sch.step(eu + 7.5 * (ec - eu), t, x).prev_sample
    close(got, x, atol=2e-4)
    loss = gad.sd_simple_loss(net, gad.DDPMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear"),
                              got[:1], cond[:1].to(dev), torch.tensor([751, 501, 251, 1]), n_noises=2,
                              generator=torch.Generator(device=dev).manual_seed(0))
    assert loss > 0 and loss == loss


def test_sd_simple_loss_matches_oracle_loop():
    """compute_model_behaviors.py:391-417 restated with the oracle U-Net and scheduler on the host: the batch is the
    sampler's whole timestep list, one fresh noise per repeat from `noise_generator`, F.mse_loss(mean) averaged over
    n_noises.  Same CPU generator on both sides (host draw, then moved).  Tolerance 1e-4 relative (fp32 U-Net)."""
    import gad
    from oracle import diffusers_ref as R
    ref, net = _pair(lora_rank=8)
    kw = dict(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000)
    z0, cond = rnd(1, 4, 16, 16, seed=21), rnd(1, 77, 96, seed=22)
    ts = torch.tensor([901, 801, 701, 601, 501, 401, 301, 201, 101, 1])
    got = gad.sd_simple_loss(net, gad.DDPMScheduler(**kw), z0.to(dev), cond.to(dev), ts, n_noises=3,
                             generator=torch.Generator().manual_seed(5))
    sch = R.DDPMScheduler(**kw)
    g = torch.Generator().manual_seed(5)
    want = 0.0
    with torch.no_grad():
        for _ in range(3):
            lat = z0.expand(len(ts), -1, -1, -1)
            noise = torch.randn(lat.size(), generator=g)
            preds = ref(sch.add_noise(lat, noise, ts), ts, cond.expand(len(ts), -1, -1)).sample
            want += F.mse_loss(preds, noise, reduction="mean")
    want = (want / 3).item()
    assert got == pytest.approx(want, rel=1e-4), (got, want)


def test_sd_model_behaviours_entry_point(tmp_path):
    """train two LoRAs (full data = reference, one Shapley coalition) with the trainer entry point, then
    compute_model_behaviors.py: db row grammar of the reference (:459-498), resume from its checkpoint, duplicate guard;
    a model compared with itself has nrmse 0 / similarity 1."""
    import json
    import pandas as pd
    from text_to_image import compute_model_behaviors as M
    from text_to_image import train_text_to_image_lora as T
    data = tmp_path / "artbench"
    T.synthetic_cache(str(data / "latent_cache.pt"), n=120, n_artists=10, res=128, ctx_dim=96)
    pd.DataFrame({"artist": sorted({f"artist_{i:03d}" for i in range(10)})}).to_csv(data / "post_impressionism_artists.csv", index=False)
    over = json.dumps(dict(block_out_channels=(64, 128, 128, 128), attention_head_dim=4, cross_attention_dim=96, sample_size=16))
    common = ["--train_data_dir", str(data), "--output_dir", str(tmp_path / "out"), "--cls_key", "style", "--cls",
              "post_impressionism", "--train_batch_size", "8", "--unet_overrides", over, "--seed", "42", "--rank", "4",
              "--method", "retrain", "--max_train_steps", "3", "--learning_rate", "1e-3"]
    assert T.main(T.parse_args(common))
    assert T.main(T.parse_args(common + ["--removal_dist", "shapley", "--removal_unit", "artist", "--removal_seed", "1"]))
    root = tmp_path / "out" / "artbench_post_impressionism" / "retrain" / "models"
    full, coal = root / "full", root / "artist_shapley" / "shapley_seed=1"
    db = str(tmp_path / "behaviours.jsonl")
    base = ["--reference_lora_dir", str(full), "--db", db, "--num_images", "3", "--resolution", "128", "--seed", "42",
            "--unet_overrides", over, "--num_inference_steps", "4", "--n_noises", "2", "--no_duplicate"]
    ck = str(tmp_path / "ck.pt")
    assert M.main(M.parse_args(base + ["--lora_dir", str(coal), "--exp_name", "retrain_artist_shapley_seed_1", "--ckpt_path", ck,
                                       "--ckpt_freq", "2"]))
    assert M.main(M.parse_args(base + ["--lora_dir", str(full), "--exp_name", "retrain_full"]))
    rows = [json.loads(l) for l in open(db)]
    assert len(rows) == 2
    r = rows[0]
    ridx = pd.read_csv(coal / "removal_idx.csv")
    assert r["remaining_idx"] == ridx["idx"][ridx["remaining"]].to_list() and r["removal_idx"] == ridx["idx"][~ridx["remaining"]].to_list()
    for i in range(3):
        for b in M.BEHAVIOURS:
            assert f"generated_image_{i}_{b}" in r and f"generated_image_{i}_{b}_time" in r
        assert math.isfinite(r[f"generated_image_{i}_simple_loss"]) and r[f"generated_image_{i}_simple_loss"] > 0
    for k in ("aesthetic_score_0.5", "aesthetic_score_0.9", "clip_prompt_score_avg", "aesthetic_score_avg", "exp_name", "seed"):
        assert k in r
    # configs 4 / 5 through the one-coalition-per-GPU scheduler: train on the coalition, then the behaviours, one row each
    from gad.coalition import run_sharded
    from gad.cycles import SDLoRACycle
    db2 = str(tmp_path / "sharded.jsonl")
    cyc = SDLoRACycle(dev, common, [a_ for a_ in base if a_ not in ("--db", db)], n_groups=10)
    recs = run_sharded(cyc, [1, 2], db_path=db2, verbose=True)                # seed 1 finds its trained LoRA (cancelled), seed 2 trains
    srows = [json.loads(l) for l in open(db2)]
    assert [r_["exp_name"] for r_ in srows] == ["retrain_artist_shapley_seed_1", "retrain_artist_shapley_seed_2"]
    assert srows[0]["remaining_idx"] == r["remaining_idx"]
    assert srows[0]["aesthetic_score_0.9"] == pytest.approx(r["aesthetic_score_0.9"], rel=1e-6)
    assert recs[1].extra[2] == pytest.approx(srows[1]["aesthetic_score_0.9"]) and len(recs[1].extra) == 8
    assert rows[1]["remaining_idx"] is None                                   # full-data LoRA dir has no removal_idx.csv
    for i in range(3):                                                         # same weights, same seed -> identical samples
        assert rows[1][f"generated_image_{i}_nrmse"] == 0.0
        assert abs(rows[1][f"generated_image_{i}_clip_similarity"] - 1.0) < 1e-5
    assert any(r[f"generated_image_{i}_nrmse"] > 0 for i in range(3))          # the coalition model differs
    # duplicate guard (:168-189) and resume from the checkpoint written after 2 of 3 images
    assert M.main(M.parse_args(base + ["--lora_dir", str(full), "--exp_name", "retrain_full"])) is False
    assert os.path.exists(ck)
    db2 = str(tmp_path / "behaviours2.jsonl")
    resumed = [a if a != db else db2 for a in base]
    assert M.main(M.parse_args(resumed + ["--lora_dir", str(coal), "--exp_name", "retrain_artist_shapley_seed_1", "--ckpt_path", ck,
                                          "--ckpt_freq", "2"]))
    r2 = json.loads(open(db2).readline())
    for i in range(3):
        assert r2[f"generated_image_{i}_nrmse"] == pytest.approx(r[f"generated_image_{i}_nrmse"], rel=1e-6)
        assert r2[f"generated_image_{i}_simple_loss"] == pytest.approx(r[f"generated_image_{i}_simple_loss"], rel=1e-5)
