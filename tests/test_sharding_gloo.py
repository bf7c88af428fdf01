"""N > 1 path on CPU: gloo processes at world size 2 and 8 (the north star's node: one coalition per GPU on 8 GPUs) shard the
coalition seeds, all_gather the fixed-size records and rank 0 writes the merged jsonl (the engine's GPU work is replaced by a
deterministic stub; the collective / rendezvous / merge / idempotence / launcher logic under test is the product's)."""
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
    n = 3 * world + 1
    recs = run_sharded(_StubEngine(), list(range(n)), db_path=db)
    assert [r.removal_seed for r in recs] == [s for s in range(n) if s != 2]       # every rank sees every new record, seed order
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


WORLDS = [2, 8]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", WORLDS)
def test_gloo_all_gather_and_merge(tmp_path, world):
    db = str(tmp_path / "db.jsonl")
    with open(db, "w") as f:                                           # seed 2 is already done: must be skipped
        f.write(json.dumps({"removal_seed": 2, "fid_value": -1.0}) + "\n")
    port = _free_port()
    mp.spawn(_worker, args=(world, port, db), nprocs=world, join=True)
    rows = [json.loads(l) for l in open(db)]
    assert rows[0]["fid_value"] == -1.0
    new = rows[1:]
    assert [r["removal_seed"] for r in new] == [s for s in range(3 * world + 1) if s != 2]
    assert all(abs(r["fid_value"] - (10.0 + 0.5 * r["removal_seed"])) < 1e-12 for r in new)
    assert not [f for f in os.listdir(tmp_path) if ".rank" in f]       # every shard consolidated away


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


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", WORLDS)
def test_rank_dying_mid_run_keeps_finished_rows_and_requeue_completes(tmp_path, world):
    """A non-zero rank is killed (os._exit) while running its second seed: its first is already durable in its shard,
    rank 0 does not hang in the all_gather (tombstone), merges every shard, and a second entry finishes the rest."""
    db = str(tmp_path / "db.jsonl")
    n, victim = 4 * world, world - 1
    die_at = victim + world                                                            # the victim's second seed
    codes = launch.spawn_workers([sys.executable, WORKER, db, str(n), f"die:{die_at}"], world, db_path=db)
    assert codes[victim] == 17 and all(c == 0 for r, c in enumerate(codes) if r != victim)
    want = sorted(s for s in range(n) if not (s % world == victim and s >= die_at))
    assert sorted(r["removal_seed"] for r in _rows(db)) == want                        # incl. the dead rank's first seed, from its shard
    assert finished_seeds(db) == set(want)
    codes = launch.spawn_workers([sys.executable, WORKER, db, str(n), "ok"], world, db_path=db)   # the requeued entry
    assert codes == [0] * world
    assert sorted(r["removal_seed"] for r in _rows(db)) == list(range(n))
    assert len({r["removal_seed"] for r in _rows(db)}) == n                            # no duplicates


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", WORLDS)
def test_rank0_dying_does_not_hang_or_spin_the_survivor(tmp_path, world):
    """rank 0 hosts the process group's store: when it is killed the survivors' rendezvous sees the store fail (or the
    tombstone), skips the collective and exits cleanly with their rows durable in their own shards; the requeued entry
    merges those shards and finishes rank 0's seeds."""
    db = str(tmp_path / "db.jsonl")
    n = 4 * world
    t0 = time.time()
    codes = launch.spawn_workers([sys.executable, WORKER, db, str(n), f"die:{2 * world}"], world, db_path=db)
    assert codes[0] == 17 and all(c == 0 for c in codes[1:]) and time.time() - t0 < 90
    assert finished_seeds(db) == {s for s in range(n) if not (s % world == 0 and s >= 2 * world)}   # nobody merged: rows sit in the shards
    codes = launch.spawn_workers([sys.executable, WORKER, db, str(n), "ok"], world, db_path=db)
    assert codes == [0] * world
    assert sorted(r["removal_seed"] for r in _rows(db)) == list(range(n))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", WORLDS)
def test_late_rank0_still_gathers(tmp_path, world):
    """Rank 0 is the SLOW one (its first coalition takes far longer than the rendezvous timeout): the other ranks have no
    deadline of their own, wait for its decision, and the collective runs - nobody gives up and leaves rank 0 alone in
    the all_gather (a rank that left after 2 x timeout kept its arrival key, and a late rank 0 published
    "gather" to peers that were gone)."""
    db = str(tmp_path / "db.jsonl")
    n = 2 * world
    t0 = time.time()
    codes = launch.spawn_workers([sys.executable, WORKER, db, str(n), "slow:0:4", "1"], world, db_path=db)
    assert codes == [0] * world and time.time() - t0 < 90
    assert sorted(r["removal_seed"] for r in _rows(db)) == list(range(n))
    assert not [f for f in os.listdir(tmp_path) if ".rank" in f]


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", WORLDS)
def test_straggler_shard_survives_and_the_launcher_consolidates_it(tmp_path, world):
    """A non-zero rank is alive but later than rank 0's rendezvous timeout: rank 0 publishes skip:<rank>, every rank skips
    the collective, rank 0 merges what is there and KEEPS the straggler's shard (it may still be appending); the
    straggler finishes into its shard; `gad.launch` ends by merging leftover shards on the host so the db is complete."""
    from gad.coalition import merge_shards
    db = str(tmp_path / "db.jsonl")
    n, slow = 2 * world, world - 1
    codes = launch.spawn_workers([sys.executable, WORKER, db, str(n), f"slow:{slow}:5", "1"], world, db_path=db)
    assert codes == [0] * world
    assert finished_seeds(db) == set(range(n))                                         # nothing lost ...
    in_db = {r["removal_seed"] for r in _rows(db)}
    assert in_db >= {s for s in range(n) if s % world != slow}                         # ... the punctual ranks' rows are merged
    merge_shards(db)                                                                   # what gad.launch.main does at its end
    assert sorted(r["removal_seed"] for r in _rows(db)) == list(range(n))
    assert not [f for f in os.listdir(tmp_path) if ".rank" in f]


def test_run_sharded_uses_the_pipelined_engine_and_retries_its_failures(tmp_path):
    """An engine with `in_flight == 2` is driven through `run_pipelined` (rows appended as coalitions finish); a coalition it
    reports as failed lands in <db>.failed and is retried through `run_coalition`."""
    from gad.coalition import run_sharded

    class Eng(_StubEngine):
        in_flight = 2
        calls = []

        def run_pipelined(self, seeds, on_record=None, on_error=None, verbose=False, n_train=2):
            out = []
            for s_ in seeds:
                self.calls.append(("pipe", s_))
                if s_ == 3:
                    on_error(s_, RuntimeError("synthetic failure in flight"))
                    continue
                out.append(_StubEngine.run_coalition(self, s_))
                on_record(out[-1])
            return out

        def run_coalition(self, seed, verbose=False):
            self.calls.append(("seq", seed))
            return _StubEngine.run_coalition(self, seed)
    db = str(tmp_path / "db.jsonl")
    eng = Eng()
    recs = run_sharded(eng, list(range(5)), db_path=db)
    assert sorted(r.removal_seed for r in recs) == [0, 1, 2, 3, 4]
    assert eng.calls == [("pipe", 0), ("pipe", 1), ("pipe", 2), ("pipe", 3), ("pipe", 4), ("seq", 3)]
    assert sorted(r["removal_seed"] for r in _rows(db)) == [0, 1, 2, 3, 4]
    fails = _rows(db + ".failed")
    assert len(fails) == 1 and fails[0]["removal_seed"] == 3 and "in flight" in fails[0]["error"]


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


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", WORLDS)
def test_bench_launcher_starts_n_ranks_before_any_gpu_call(tmp_path, world):
    """`python bench.py --gpus N` with no WORLD_SIZE: the parent spawns N ranks; n_gpus / ranks_seen come from the process
    group's collective, not from the flag."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["GAD_STUB_DB"] = str(tmp_path / "stub.jsonl")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(world), "--stub", "--steps", "3", "--warmup", "0"],
                       env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == world and line["ranks_seen"] == list(range(world)) and line["stub"] is True
    assert line["records_gathered"] == list(range(3 * world))
    assert sorted(x["removal_seed"] for x in _rows(env["GAD_STUB_DB"])) == list(range(3 * world))


def _noise_proc(q):
    import time as _t
    t0 = _t.time()
    for c in range(320):                                               # one coalition's sampler noise: 10 240 x 3 x 32 x 32, reference batches of 32
        torch.randn((32, 3, 32, 32), generator=torch.Generator().manual_seed(c), dtype=torch.float32)
    q.put(_t.time() - t0)


@pytest.mark.timeout(300)
def test_eight_ranks_host_noise_does_not_serialise():
    """Every rank draws its coalition's initial sampler noise from per-batch CPU generators (bit-identical to the
    reference's `torch.Generator().manual_seed(counter)`, src/diffusion_utils.py:336-341 -> coalition.FusedSampler.initial_noise):
    eight ranks doing so at once must not queue behind each other - the draw is ~0.2 s against ~100 s of GPU work per
    coalition, and eight concurrent draws finish in well under eight sequential ones on this host."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    _noise_proc(q)
    alone = q.get()
    procs = [ctx.Process(target=_noise_proc, args=(q,)) for _ in range(8)]
    t0 = time.time()
    for p_ in procs:
        p_.start()
    together = [q.get(timeout=120) for _ in procs]
    for p_ in procs:
        p_.join()
    wall = time.time() - t0
    print(f"sampler noise of one coalition: {alone:.3f} s alone; 8 ranks at once: slowest {max(together):.3f} s (process wall {wall:.1f} s incl. interpreter start)")
    assert max(together) < 5.0 and max(together) < 0.05 * 97.0          # < 5 % of a coalition's ~97 s even in the worst case


def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--stub"], env=env,
                       capture_output=True, text=True)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr
