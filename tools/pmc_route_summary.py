"""profiles/pmc_summary_<workload>.json for a secondary workload: HBM bytes per launch of its dominant kernel FAMILY from two separate
rocprofv3 --pmc passes (FETCH_SIZE doubled per the gfx950 note, WRITE_SIZE exact).  The family = the kernels one bracket of bench.py's
profiler spans (e.g. attn_delta + attn_bwd1 + attn_dq_reduce); bytes per launch = all their bytes / launches of the first-named kernel.
usage: pmc_route_summary.py <dir with pmc_FETCH_SIZE/ pmc_WRITE_SIZE/> <workload> <dominant key or ''> <kernel substring> [more substrings]"""
import csv, glob, json, sys
from collections import defaultdict

d, workload, key, kernels = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]


def per_kernel(counter):
    per = defaultdict(lambda: defaultdict(float))
    for f in glob.glob(f"{d}/pmc_{counter}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for k in kernels:
                if k in r["Kernel_Name"]:
                    per[k][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
                    break
    return {k: (len(v), sum(v.values())) for k, v in per.items()}


fe, wr = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
n = fe.get(kernels[0], (0, 0.0))[0]
total = sum(2 * fe.get(k, (0, 0.0))[1] + wr.get(k, (0, 0.0))[1] for k in kernels) * 1024
out = {"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two SEPARATE passes, tools/prof_round.sh) -- python3 bench.py --workload {workload} "
                 "--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-train-rate ; tools/pmc_route_summary.py",
       "dominant_kernel": " + ".join(kernels), "family_launches_per_pass": n,
       "per_kernel": {k: {"launches": fe.get(k, (0, 0))[0], "FETCH_SIZE_KiB_total": fe.get(k, (0, 0.0))[1], "WRITE_SIZE_KiB_total": wr.get(k, (0, 0.0))[1]} for k in kernels},
       "dominant_kernel_hbm_bytes_per_launch": total / n if n else None,
       "note": "FETCH_SIZE doubled (gfx950 tallies the 128-B requests of 16-B/lane streams at 64 B), WRITE_SIZE exact; fabric-side counters include Infinity-Cache hits"}
if key:
    out["dominant_key"] = key
print(json.dumps(out, indent=1))
