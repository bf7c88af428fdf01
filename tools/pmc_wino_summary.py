"""profiles/pmc_summary.json from the per-counter files of tools/prof_round.sh ... pmc (round 4 layout): HBM bytes per launch of every
kernel of the Winograd F(4x4,3x3) route over the cifar20 launch mix - the one-launch product kernel wino4_fused2_kernel (the bench's
dominant kernel), the input transform, GroupNorm writing V, and the three-launch fallback's kernels.  usage: pmc_wino_summary.py <dir> <tag>"""
import json, os, sys
d, tag = sys.argv[1], sys.argv[2]


def ld(c, k):
    p = f"{d}/{tag}_pmc_{c}_{k}.json"
    return json.load(open(p)) if os.path.exists(p) else None


parts = {}
for k in ("wino4_fused2", "wino_input", "gn_wino4", "wino_gemm", "wino_output"):
    f, w = ld("FETCH_SIZE", k), ld("WRITE_SIZE", k)
    if not f or not w or not f["launches"]:
        continue
    # FETCH_SIZE doubled (gfx950 tallies the 128-B requests of 16-B/lane streams at 64 B: MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; KiB
    parts[k] = {"launches_per_pass": f["launches"], "avg_FETCH_SIZE_KiB": f["avg"], "avg_WRITE_SIZE_KiB": w["avg"],
                "hbm_bytes_per_launch": (2 * f["avg"] + w["avg"]) * 1024}
out = {
 "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE --output-format csv (two SEPARATE passes, tools/prof_round.sh via "
           "tools/evidence_round.sh) -- python3 bench.py --workload cifar20 --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing "
           "--no-train-rate --in-flight 1 ; tools/pmc_summarize.py, tools/pmc_wino_summary.py",
 "dominant_key": "conv_fwd_wino4_32_1_4",
 "dominant_kernel": "wino4_fused2_kernel<2, 3> (all 36 F(4x4) products + the output transform of a 3x3 convolution in one launch): every launch "
                    "of the run - sampler forward at B=1024 (V written by gn_wino4_kernel), training forward at B=128 and the data gradients "
                    "on the rotated weights (V written by wino4_input_kernel)",
 "traffic_is": "products_stage",
 **parts,
 "dominant_kernel_hbm_bytes_per_launch": parts.get("wino4_fused2", {}).get("hbm_bytes_per_launch"),
 "note": "FETCH_SIZE doubled (gfx950 tallies the 128-B requests of 16-B/lane streams at 64 B), WRITE_SIZE exact; fabric-side counters include "
         "Infinity-Cache hits.  wino4_fused2_kernel reads V (2.25 x the input) and U, writes y and reads the residual: against |V| + |U| + |y| "
         "(+ |residual|) algorithmic for that stage; V itself is written once by gn_wino4_kernel (x in, V out) or wino4_input_kernel.  Round 3's "
         "three-launch route for comparison: profiles/r03_pmc_summary.json (4.7 x the convolution's algorithmic bytes)",
}
print(json.dumps(out, indent=1))
