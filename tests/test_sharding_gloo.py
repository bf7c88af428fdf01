"""N > 1 path on CPU: world-size-2 gloo processes shard the coalition seeds, all_gather the fixed-size records
and rank 0 writes the merged jsonl (the engine's GPU work is replaced by a deterministic stub; the
collective/merge/idempotence logic under test is the product's)."""
import json
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gad.coalition import CoalitionRecord, gather_records, shard_seeds


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _StubEngine:
    """Same surface run_sharded uses; run_coalition is a cheap deterministic function of the seed."""
    n_groups = 20
    device = torch.device("cpu")

    def run_coalition(self, seed, verbose=False):
        from src.datasets import remove_data_by_shapley
        labels = [i // 5 for i in range(100)]
        rem, rmv = remove_data_by_shapley([(None, l) for l in labels], seed=seed, by_class=True)
        return CoalitionRecord(seed, len(rem), len(rmv), 10.0 + seed * 0.5, 0.1, 1.0, 2.0, 3,
                               sorted(set(labels[i] for i in rem)))

    def jsonl_row(self, rec):
        return dict(removal_seed=rec.removal_seed, fid_value=rec.fid_value, n_remaining=rec.n_remaining,
                    remaining_classes=rec.remaining_classes)


def _worker(rank, world, port, db):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gad.coalition import run_sharded
    recs = run_sharded(_StubEngine(), list(range(7)), db_path=db)
    assert [r.removal_seed for r in recs] == [0, 1, 3, 4, 5, 6]       # every rank sees every new record, seed order
    dist.barrier()
    dist.destroy_process_group()


def test_shard_seeds_partition():
    seeds = list(range(11))
    parts = [shard_seeds(seeds, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == seeds and parts[1] == [1, 5, 9]


def test_record_pack_roundtrip():
    r = CoalitionRecord(5, 6500, 3500, 12.25, 0.03, 70.5, 161.0, 1000, [0, 3, 19], 1.75, 0.5, 0.25)
    v = r.pack(20)
    assert v.dtype == torch.float64 and v.numel() == CoalitionRecord.NSCALAR + 20
    assert CoalitionRecord.unpack(v) == r


@pytest.mark.timeout(180)
def test_two_rank_gloo_all_gather_and_merge(tmp_path):
    db = str(tmp_path / "db.jsonl")
    with open(db, "w") as f:                                           # seed 2 is already done: must be skipped
        f.write(json.dumps({"removal_seed": 2, "fid_value": -1.0}) + "\n")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, db), nprocs=2, join=True)
    rows = [json.loads(l) for l in open(db)]
    assert rows[0]["fid_value"] == -1.0
    new = rows[1:]
    assert [r["removal_seed"] for r in new] == [0, 1, 3, 4, 5, 6]
    assert all(abs(r["fid_value"] - (10.0 + 0.5 * r["removal_seed"])) < 1e-12 for r in new)
