"""Child process of tests/test_sharding_gloo.py: one gloo rank of run_sharded over a stub engine.
argv: db_path n_seeds mode [rendezvous timeout s]      mode: ok | raise_once:<seed> | raise_always:<seed> | die:<seed> | slow:<seed>:<seconds>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from gad.coalition import CoalitionRecord, run_sharded  # noqa: E402


class StubEngine:
    n_groups, device = 20, torch.device("cpu")

    def __init__(self, mode):
        self.kind, _, seed = mode.partition(":")
        seed, _, self.secs = seed.partition(":")
        self.bad = int(seed) if seed else -1
        self.calls = {}

    def run_coalition(self, seed, verbose=False):
        self.calls[seed] = self.calls.get(seed, 0) + 1
        if seed == self.bad:
            if self.kind == "slow":
                import time
                time.sleep(float(self.secs))                   # a straggler: alive, late
            if self.kind == "die":
                os._exit(17)                                   # a GPU fault / OOM kill: no exception, no cleanup
            if self.kind == "raise_always" or (self.kind == "raise_once" and self.calls[seed] == 1):
                raise RuntimeError(f"synthetic failure of coalition {seed}")
        return CoalitionRecord(seed, 100 - seed, seed, 10.0 + 0.5 * seed, 0.1, 1.0, 2.0, 3, [seed % 20])

    def jsonl_row(self, rec):
        return dict(removal_seed=rec.removal_seed, fid_value=rec.fid_value, n_remaining=rec.n_remaining)


if __name__ == "__main__":
    db, n, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
    tmo = float(sys.argv[4]) if len(sys.argv) > 4 else 60
    run_sharded(StubEngine(mode), list(range(n)), db_path=db, retries=1, rendezvous_timeout_s=tmo)
    if world > 1:
        dist.destroy_process_group()
