"""ArtBench metadata on the consumer side.

The reference writes `{data_dir}/metadata.csv` (columns file_name, caption, artist, style, filename) together with
`{style}_artists.csv` / `{style}_filenames.csv` (text_to_image/artbench/create_metadata.py:77-95,113-114) and reads
them back through the HF `imagefolder` loader, filtering on `--cls_key style --cls post_impressionism`
(train_text_to_image_lora.py:911-934,1011-1024).  Here the pixels never reach the trainer (the frozen VAE / CLIP
encoders are hub-fetched and off the hot path): the tables are joined with precomputed latents and prompt
embeddings into the `latent_cache.pt` the MI355X trainer keeps resident in HBM.
"""
import os

import pandas as pd
import torch

COLUMNS = ["file_name", "caption", "artist", "style", "filename"]


def read_metadata(data_dir: str, cls_key: str = None, cls: str = None) -> pd.DataFrame:
    df = pd.read_csv(os.path.join(data_dir, "metadata.csv"))
    missing = [c for c in COLUMNS if c not in df.columns]
    if missing:
        raise KeyError(f"{data_dir}/metadata.csv lacks columns {missing}")
    if cls_key is not None and cls is not None:
        df = df[df[cls_key] == cls]
    return df.reset_index(drop=True)


def unit_table(data_dir: str, cls: str, unit: str) -> pd.DataFrame:
    """`{cls}_{unit}s.csv`: one row per removal unit, sorted (create_metadata.py:86-95); the coalition samplers
    index its rows (train_text_to_image_lora.py:944-947)."""
    return pd.read_csv(os.path.join(data_dir, f"{cls}_{unit}s.csv" if cls else f"{unit}s.csv"))


def assemble_latent_cache(data_dir: str, latents: dict, text_emb, cls_key: str = "style", cls: str = None,
                          out: str = None) -> dict:
    """Join metadata.csv with `latents` ({file_name: [4,h,w] tensor, already x vae.config.scaling_factor}) and
    `text_emb` (one [77,D] tensor shared by all rows, or {caption: [77,D]}) in metadata row order."""
    df = read_metadata(data_dir, cls_key if cls else None, cls)
    absent = [f for f in df["file_name"] if f not in latents]
    if absent:
        raise KeyError(f"{len(absent)} rows of metadata.csv have no latent, e.g. {absent[:3]}")
    cache = {"latents": torch.stack([latents[f].float() for f in df["file_name"]])}
    if isinstance(text_emb, dict):
        cache["text_emb"] = torch.stack([text_emb[c].float() for c in df["caption"]])
    else:
        cache["text_emb"] = text_emb.float().reshape(1, *text_emb.shape[-2:])
    for c in ("artist", "filename", "style", "caption"):
        cache[c] = df[c].tolist()
    if out is None:
        out = os.path.join(data_dir, "latent_cache.pt")
    torch.save(cache, out)
    return cache
