"""Node-local launcher of the one-coalition-per-GPU job: the replacement of the reference's SLURM job array
(text_to_image/experiments/setup_unlearn_commands.py:160-214 writes one command per coalition,
unlearn.job:15,17 runs them with `--requeue` and `--open-mode=append`).

    python -m gad.launch --gpus 8 --dataset cifar100 --seeds 0:64 --db /path/db.jsonl [--gd_steps 1000 ...]

The parent process NEVER initialises the GPU (no HIP call, no torch.cuda call): it starts one child per GPU with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, waits, and forwards the exit codes.  A child that
dies (GPU fault, OOM kill) gets a tombstone next to the db so that the surviving ranks skip the final collective
instead of hanging in it (`gad.coalition.run_sharded`); finished coalitions are already durable in the per-rank
shards, and the launch is re-entered (`--requeue R`) for the seeds that are still missing.
"""
from __future__ import annotations

import argparse
import os
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def tombstone(db_path: str, rank: int) -> str:
    return f"{db_path}.rank{rank}.dead"


def clear_tombstones(db_path: Optional[str], world: int):
    if db_path:
        for r in range(world):
            try:
                os.remove(tombstone(db_path, r))
            except FileNotFoundError:
                pass


def spawn_workers(cmd: Sequence[str], nprocs: int, extra_env: Optional[Dict[str, str]] = None,
                  db_path: Optional[str] = None, poll_s: float = 0.2, grace_s: Optional[float] = None) -> List[int]:
    """Start `cmd` nprocs times as fresh child processes (rank r: RANK = LOCAL_RANK = r), wait for all of them and
    return their exit codes.  Children inherit stdout / stderr, so rank 0's JSON line is the parent's.  A child that
    exits non-zero while others are still running is marked dead for them (tombstone).  The survivors keep working -
    their coalitions are hours of useful work and the final rendezvous does not wait for a tombstoned rank - for at
    most `grace_s` seconds after the first death (default: GAD_SURVIVOR_GRACE, unset = unbounded); then they are
    terminated, killed if they ignore that, and reported with their (negative) signal codes, so a requeue entry
    starts fresh children instead of waiting on a hung rank."""
    if grace_s is None and os.environ.get("GAD_SURVIVOR_GRACE"):
        grace_s = float(os.environ["GAD_SURVIVOR_GRACE"])
    env0 = dict(os.environ)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC: RCCL across processes needs it on this pool
    env0.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(nprocs),
                LOCAL_WORLD_SIZE=str(nprocs))
    if extra_env:
        env0.update(extra_env)
    clear_tombstones(db_path, nprocs)
    procs = []
    for r in range(nprocs):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(list(cmd), env=env))
    codes: List[Optional[int]] = [None] * nprocs
    first_death, stage = None, 0
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                rc = p.poll()
                if rc is not None:
                    codes[r] = rc
                    if rc != 0:
                        first_death = first_death or time.time()
                        if db_path:
                            with open(tombstone(db_path, r), "w") as f:
                                f.write(f"exit code {rc}\n")
        if grace_s is not None and first_death is not None:
            late = time.time() - first_death - grace_s
            if late > 0 and stage == 0 or late > 10 and stage == 1:
                for r, p in enumerate(procs):
                    if codes[r] is None:
                        (p.terminate if stage == 0 else p.kill)()
                stage += 1
        time.sleep(poll_s)
    return [int(c) for c in codes]


def parse_seeds(spec: str) -> List[int]:
    """"0:64" -> 0..63, "3,5,9" -> [3, 5, 9]"""
    out: List[int] = []
    for part in spec.split(","):
        if ":" in part:
            a, b = part.split(":")
            out += list(range(int(a), int(b)))
        elif part:
            out.append(int(part))
    return out


def _parse(argv=None):
    ap = argparse.ArgumentParser(description="one coalition per GPU, N GPUs of one node")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dataset", default="cifar100")
    ap.add_argument("--seeds", default="0:8", help="removal seeds: a:b or a,b,c")
    ap.add_argument("--db", required=True)
    ap.add_argument("--gd_steps", type=int, default=None)
    ap.add_argument("--n_samples", type=int, default=10240)
    ap.add_argument("--batch_size", type=int, default=32, help="reference sampling batch (per-batch generator seeds)")
    ap.add_argument("--num_inference_steps", type=int, default=100)
    ap.add_argument("--opt_seed", type=int, default=42)
    ap.add_argument("--mixed_precision", default="no", choices=["no", "fp16", "bf16"])
    ap.add_argument("--in_flight", type=int, default=3, choices=[1, 2, 3],
                    help="cifar cycle: coalitions in flight per GPU - k > 1: the sampling phase of one beside the training phases of the "
                         "next k - 1, each on its own HIP stream (CoalitionEngine.run_pipelined); 1: one after the other")
    ap.add_argument("--requeue", type=int, default=1, help="re-entries of the whole launch while seeds are missing")
    ap.add_argument("--retries", type=int, default=1, help="in-process retries of a coalition that raised")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--cycle", default="cifar", choices=["cifar", "celeba", "sd"],
                    help="cifar: the in-process CIFAR-family engine; celeba / sd: the kept entry points of configs 3 / 4-5 run "
                         "in-process per coalition (gad.cycles)")
    ap.add_argument("--cycle_args", default="[]", help="celeba: JSON list of shared unlearn.py flags; sd: JSON object "
                    "{\"train\": [...], \"behaviours\": [...], \"n_groups\": 258}")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def worker(a) -> int:
    """One rank: bind cuda:LOCAL_RANK, join the process group (backend "nccl" IS RCCL on ROCm), run this rank's share."""
    import torch
    import torch.distributed as dist

    import gad
    from gad.coalition import CoalitionEngine, run_sharded

    world, rank, local = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
    if os.environ.get("GAD_SHARE_GPU0"):            # rehearsal of the N-rank path on a one-GPU box (with --backend gloo)
        local = 0
    dev = torch.device(f"cuda:{local}")
    torch.cuda.set_device(dev)
    if world > 1:
        from datetime import timedelta
        # bounded: a rank that died before / during the rendezvous must not leave the others blocked in it forever
        tmo = timedelta(seconds=float(os.environ.get("GAD_INIT_TIMEOUT", 600)))
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group("gloo", timeout=tmo)
    gad.set_operand_precision(a.mixed_precision)
    if a.cycle == "cifar":
        engine = CoalitionEngine(a.dataset, device=dev, gd_steps=a.gd_steps, n_samples=a.n_samples,
                                 sample_batch=a.batch_size, num_inference_steps=a.num_inference_steps, opt_seed=a.opt_seed)
        engine.in_flight = a.in_flight
    else:
        import json
        from gad import cycles
        spec = json.loads(a.cycle_args)
        engine = (cycles.CelebaCycle(dev, spec) if a.cycle == "celeba"
                  else cycles.SDLoRACycle(dev, spec["train"], spec["behaviours"], int(spec.get("n_groups", 258))))
    recs = run_sharded(engine, parse_seeds(a.seeds), db_path=a.db, verbose=True, retries=a.retries)
    if rank == 0:
        print(f"[gad.launch] {len(recs)} coalition records consolidated into {a.db}", flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main(argv=None) -> int:
    a = _parse(argv)
    if a.worker:
        return worker(a)
    from gad.coalition import finished_seeds            # host-only import path: no GPU call
    seeds = set(parse_seeds(a.seeds))
    args = [x for x in (argv if argv is not None else sys.argv[1:])]
    cmd = [sys.executable, "-m", "gad.launch", "--worker"] + args
    pkg_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {"PYTHONPATH": pkg_root + os.pathsep + os.environ.get("PYTHONPATH", "")}
    codes = [0]
    for entry in range(a.requeue + 1):
        if os.path.dirname(a.db):
            os.makedirs(os.path.dirname(a.db), exist_ok=True)    # shards and tombstones live next to the db
        missing = sorted(seeds - finished_seeds(a.db))
        if not missing:
            break
        print(f"[gad.launch] entry {entry}: {len(missing)} coalitions to run on {a.gpus} GPUs", flush=True)
        codes = spawn_workers(cmd, a.gpus, env, db_path=a.db)
        if any(codes):
            print(f"[gad.launch] exit codes {codes}", flush=True)
    clear_tombstones(a.db, a.gpus)
    # Rows may still sit only in rank shards - rank 0 died after the work was done, or the ranks skipped the collective and
    # rank 0 never merged: consolidate them here, on the host (no GPU, no process group), so that what lds.py reads is the db.
    from gad.coalition import _read_rows, _shard_paths, merge_shards
    if _shard_paths(a.db):
        moved = merge_shards(a.db)
        print(f"[gad.launch] merged {len(moved)} row(s) left in rank shards into {a.db}", flush=True)
    in_db = {int(r["removal_seed"]) for r in _read_rows(a.db)} if os.path.exists(a.db) else set()
    missing = sorted(seeds - in_db)
    if missing:
        print(f"[gad.launch] still missing from {a.db} after {a.requeue} requeues: {missing}", flush=True)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
