"""Demographic-diversity score arithmetic (reference diversity_score.py:122-171): Ward clustering of the
reference embeddings, nearest-mean-cluster assignment of the generated ones, entropy of the proportions.
The embeddings come from the caller (BLIP-VQA vision pooler in the reference, fetched from the hub :89-90)."""
import numpy as np
from scipy.cluster.hierarchy import fcluster, ward
from scipy.spatial.distance import squareform


def diversity_from_embeddings(emb_ref, emb_gen, num_cluster):
    emb_ref, emb_gen = np.asarray(emb_ref), np.asarray(emb_gen)
    sim = emb_ref @ emb_ref.T
    top = np.max(sim)
    dist = top - sim
    np.fill_diagonal(dist, 0)
    labels = fcluster(ward(squareform(dist, checks=False)), num_cluster, criterion="maxclust")
    d_gen = top - emb_gen @ emb_ref.T
    # mean distance of every generated sample to every reference cluster, in one contraction
    member = np.stack([(labels == i).astype(np.float64) for i in range(1, num_cluster + 1)], axis=1)
    counts = member.sum(axis=0)
    with np.errstate(invalid="ignore", divide="ignore"):
        mean_d = (d_gen @ member) / counts          # empty cluster -> nan (np.mean of empty), never the argmin
    assigned = np.nanargmin(mean_d, axis=1) + 1 if np.isfinite(mean_d).any() else np.ones(len(emb_gen), int)
    cluster_count = np.array([(assigned == i).sum() for i in range(1, num_cluster + 1)], dtype=np.float64)
    prop = cluster_count / len(assigned)
    entropy = float(-np.sum(prop * np.log2(prop + np.finfo(float).eps)))
    return entropy, cluster_count.tolist(), prop.tolist(), labels, assigned
