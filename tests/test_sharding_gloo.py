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


# ---- durability / failure handling of the scheduler (reference: per-job `open(db, "a+")` rows,
# ---- unlearn.job `--requeue`, setup_unlearn_commands.py:133-154 re-entry) -------------------------------------
import subprocess  # noqa: E402
import time  # noqa: E402

from gad import launch  # noqa: E402
from gad.coalition import finished_seeds  # noqa: E402

WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_shard_worker.py")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rows(path):
    return [json.loads(l) for l in open(path)] if os.path.exists(path) else []


@pytest.mark.timeout(120)
def test_failed_coalition_is_recorded_and_retried(tmp_path):
    db = str(tmp_path / "db.jsonl")
    rc = subprocess.run([sys.executable, WORKER, db, "6", "raise_once:3"]).returncode
    assert rc == 0
    assert sorted(r["removal_seed"] for r in _rows(db)) == [0, 1, 2, 3, 4, 5]          # retried in a fresh cycle
    fails = _rows(db + ".failed")
    assert len(fails) == 1 and fails[0]["removal_seed"] == 3 and "synthetic failure" in fails[0]["error"]
    assert not launch.os.path.exists(db + ".rank0")                                    # shard consolidated away


def test_db_under_a_directory_that_does_not_exist_yet(tmp_path):
    """`--db out/run7/db.jsonl` on a fresh machine: the scheduler creates the directory for the db, its shards and tombstones
    instead of failing after the first coalition has been computed."""
    db = str(tmp_path / "fresh" / "nested" / "db.jsonl")
    rc = subprocess.run([sys.executable, WORKER, db, "4", ""]).returncode
    assert rc == 0
    assert sorted(r["removal_seed"] for r in _rows(db)) == [0, 1, 2, 3]


@pytest.mark.timeout(120)
def test_persistently_failing_coalition_does_not_lose_the_others(tmp_path):
    db = str(tmp_path / "db.jsonl")
    rc = subprocess.run([sys.executable, WORKER, db, "6", "raise_always:2"]).returncode
    assert rc == 0
    assert sorted(r["removal_seed"] for r in _rows(db)) == [0, 1, 3, 4, 5]
    assert len(_rows(db + ".failed")) == 2                                              # first attempt + one retry
    # re-entry runs only the missing seed
    rc = subprocess.run([sys.executable, WORKER, db, "6", "ok"]).returncode
    assert rc == 0 and sorted(r["removal_seed"] for r in _rows(db)) == [0, 1, 2, 3, 4, 5]


@pytest.mark.timeout(180)
def test_rank_dying_mid_run_keeps_finished_rows_and_requeue_completes(tmp_path):
    """rank 1 is killed (os._exit) while running seed 5: its earlier seeds are already durable in its shard, rank 0
    does not hang in the all_gather (tombstone), merges every shard, and a second entry finishes seed 5 and 7."""
    db = str(tmp_path / "db.jsonl")
    codes = launch.spawn_workers([sys.executable, WORKER, db, "8", "die:5"], 2, db_path=db)
    assert codes[0] == 0 and codes[1] == 17
    assert sorted(r["removal_seed"] for r in _rows(db)) == [0, 1, 2, 3, 4, 6]          # 1, 3 came from the dead rank's shard
    assert finished_seeds(db) == {0, 1, 2, 3, 4, 6}
    codes = launch.spawn_workers([sys.executable, WORKER, db, "8", "ok"], 2, db_path=db)   # the requeued entry
    assert codes == [0, 0]
    assert sorted(r["removal_seed"] for r in _rows(db)) == list(range(8))
    assert len({r["removal_seed"] for r in _rows(db)}) == 8                            # no duplicates


@pytest.mark.timeout(180)
def test_rank0_dying_does_not_hang_or_spin_the_survivor(tmp_path):
    """rank 0 hosts the process group's store: when it is killed the survivor's rendezvous sees the store fail (or the
    tombstone), skips the collective and exits cleanly with its rows durable in its own shard; the requeued entry
    merges that shard and finishes rank 0's seeds."""
    db = str(tmp_path / "db.jsonl")
    t0 = time.time()
    codes = launch.spawn_workers([sys.executable, WORKER, db, "8", "die:4"], 2, db_path=db)
    assert codes[0] == 17 and codes[1] == 0 and time.time() - t0 < 60
    assert finished_seeds(db) == {0, 2, 1, 3, 5, 7}                                    # nobody merged: rows sit in the shards
    codes = launch.spawn_workers([sys.executable, WORKER, db, "8", "ok"], 2, db_path=db)
    assert codes == [0, 0]
    assert sorted(r["removal_seed"] for r in _rows(db)) == list(range(8))


def test_merge_keeps_the_shard_of_a_rank_that_may_still_be_alive(tmp_path):
    """A rank that missed the rendezvous by timeout (no tombstone) may still be appending: its rows are copied into the
    db, its shard stays; a consumed shard is renamed before it is read."""
    from gad.coalition import merge_shards
    db = str(tmp_path / "db.jsonl")
    for r, seeds in ((0, [0, 2]), (1, [1, 3])):
        with open(f"{db}.rank{r}", "w") as f:
            for s in seeds:
                f.write(json.dumps({"removal_seed": s, "fid_value": float(s), "device": f"cuda:{r}"}) + "\n")
    gathered = [{"removal_seed": 1, "fid_value": 1.0, "device": "cuda:0"}]             # rebuilt by rank 0: must not win
    assert merge_shards(db, gathered, keep_ranks=[1]) == [0, 1, 2, 3]
    assert os.path.exists(f"{db}.rank1") and not os.path.exists(f"{db}.rank0")
    assert {r["removal_seed"]: r["device"] for r in _rows(db)} == {0: "cuda:0", 1: "cuda:1", 2: "cuda:0", 3: "cuda:1"}
    assert merge_shards(db, keep_ranks=[]) == [] and not os.path.exists(f"{db}.rank1")   # later entry: nothing new, consumed


@pytest.mark.timeout(120)
def test_survivors_are_stopped_after_the_grace_period(tmp_path):
    """spawn_workers(grace_s=...): once a rank has died the others get a bounded time, then terminate() / kill()."""
    prog = "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(3)\ntime.sleep(600)\n"
    t0 = time.time()
    codes = launch.spawn_workers([sys.executable, "-c", prog], 2, grace_s=1.0)
    assert codes[1] == 3 and codes[0] < 0 and time.time() - t0 < 60


@pytest.mark.timeout(180)
def test_bench_launcher_starts_n_ranks_before_any_gpu_call(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent spawns 2 ranks; n_gpus comes from the process group."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["GAD_STUB_DB"] = str(tmp_path / "stub.jsonl")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--stub", "--steps", "3", "--warmup", "0"],
                       env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0, 1] and line["stub"] is True
    assert line["records_gathered"] == [0, 1, 2, 3, 4, 5]
    assert sorted(x["removal_seed"] for x in _rows(env["GAD_STUB_DB"])) == [0, 1, 2, 3, 4, 5]


def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--stub"], env=env,
                       capture_output=True, text=True)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr
