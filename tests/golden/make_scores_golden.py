"""Golden vectors for the score-tail arithmetic by RUNNING THE REFERENCE'S OWN FUNCTIONS on CPU (build container only):
  src/attributions/global_scores/precision_recall.py   ManifoldBuilder(features=...) / calc_pr   (:54-72,195-295)
  src/attributions/global_scores/inception_score.py    eval_is                                   (:15-76)
Placeholders stand in only for what those functions do not compute: `src.constants`, `src.datasets` (imports), the VGG
weights (features are passed in), and `pytorch_fid.inception._inception_v3`, which becomes a seeded linear classifier on
8x8 images so that eval_is's own softmax / KL / exp arithmetic runs unchanged.  `ManifoldBuilder.op_device` is only
assigned on the extract-features path of the reference (:164), so it is pointed at "cpu" for the features= path.
Run:  python tests/golden/make_scores_golden.py   ->  tests/golden/scores.npz
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def tiny_classifier():
    g = torch.Generator().manual_seed(99)
    lin = torch.nn.Linear(3 * 8 * 8, 1000)
    with torch.no_grad():
        lin.weight.copy_(torch.randn(1000, 192, generator=g) * 0.05)
        lin.bias.copy_(torch.randn(1000, generator=g) * 0.1)
    return torch.nn.Sequential(torch.nn.Flatten(), lin)


def main():
    def ph(name, **a):
        m = types.ModuleType(name)
        m.__dict__.update(a)
        sys.modules[name] = m
        return m

    class _A:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return self

        def __getattr__(self, k):
            return _A()

    sys.path.insert(0, REF)
    ph("src.constants", DATASET_DIR="/tmp/_ds", OUTDIR="/tmp/_out", LOGDIR="/tmp/_log", PRETRAINEDMODEL_DIR="/tmp/_pm",
       MAX_NUM_SAMPLE_IMAGES_TO_SAVE=64)
    ph("src.datasets", ImageDataset=_A, create_dataset=_A)
    ph("pytorch_fid")
    ph("pytorch_fid.inception", _inception_v3=lambda **k: tiny_classifier())
    from src.attributions.global_scores import inception_score as IS
    from src.attributions.global_scores import precision_recall as PR

    g = torch.Generator().manual_seed(0)
    ref = torch.randn(300, 64, generator=g)
    gen = torch.randn(200, 64, generator=g) * 1.1 + 0.1
    PR.ManifoldBuilder.op_device = "cpu"
    m_ref = PR.ManifoldBuilder(features=ref, nhood_size=3, row_batch_size=128, col_batch_size=100).manifold
    m_gen = PR.ManifoldBuilder(features=gen, nhood_size=3, row_batch_size=128, col_batch_size=100).manifold
    precision, recall = PR.calc_pr(m_gen, m_ref, 128, 100, "cpu")

    images = torch.rand(40, 3, 8, 8, generator=g)
    is1 = IS.eval_is(images, batch_size=16, resize=False, splits=1)
    is4 = IS.eval_is(images, batch_size=16, resize=False, splits=4)
    np.savez_compressed(os.path.join(OUT, "scores.npz"), pr_ref=ref.numpy(), pr_gen=gen.numpy(),
                        kth_ref=m_ref.kth.float().numpy(), kth_gen=m_gen.kth.float().numpy(),
                        precision=np.array(precision), recall=np.array(recall),
                        is_images=images.numpy(), is_splits1=np.array(is1), is_splits4=np.array(is4))
    print("precision", precision, "recall", recall, "IS", is1, is4)


if __name__ == "__main__":
    main()
