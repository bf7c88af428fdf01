"""profiles/pmc_summary.json from the per-counter files of tools/prof_round.sh ... pmc: HBM bytes per convolution of the Winograd
F(4x4,3x3) route (wino4_input_kernel + the 36 batched products gemm_kernel<..., 4, 1> + wino4_output_kernel) over the cifar20 launch
mix.  usage: pmc_wino_summary.py <dir> <tag>"""
import json, sys
d, tag = sys.argv[1], sys.argv[2]
ld = lambda c, k: json.load(open(f"{d}/{tag}_pmc_{c}_{k}.json"))
parts = {}
total = 0.0
for k in ("wino_input", "wino_gemm", "wino_output"):
    f, w = ld("FETCH_SIZE", k), ld("WRITE_SIZE", k)
    # FETCH_SIZE doubled (gfx950 tallies the 128-B requests of 16-B/lane streams at 64 B: MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; KiB
    b = (2 * f["avg"] + w["avg"]) * 1024
    parts[k] = {"launches_per_pass": f["launches"], "avg_FETCH_SIZE_KiB": f["avg"], "avg_WRITE_SIZE_KiB": w["avg"], "hbm_bytes_per_launch": b}
    total += b
out = {
 "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE --output-format csv (two SEPARATE passes, tools/prof_round.sh via "
           "tools/evidence_round.sh) -- python3 bench.py --workload cifar20 --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing "
           "--no-train-rate ; tools/pmc_summarize.py, tools/pmc_wino_summary.py",
 "dominant_kernel": "wino4_input_kernel + wino4_gemm_kernel<64, 128> (six-position products; small launches: 36 batched products on "
                    "gemm_kernel<20, 22, 64, 64, 4, 1>) + wino4_output_kernel (Winograd F(4x4,3x3) = one 3x3 convolution): every launch of the run - sampler forward at B=1024, training forward at B=128 and the data "
                    "gradients on the rotated weights",
 **parts,
 "dominant_kernel_hbm_bytes_per_launch": total,
 "note": "FETCH_SIZE doubled (gfx950 tallies the 128-B requests of 16-B/lane streams at 64 B), WRITE_SIZE exact; fabric-side counters include "
         "Infinity-Cache hits.  The route moves the transformed input V (2.25 x the input) and the 36 product panels M (2.25 x the output) out "
         "and back in: |x| + 2.25|x| written, 2.25|x| (+ U) read and 2.25|y| written, 2.25|y| read + |y| written, against |x| + |w| + |y| "
         "algorithmic.  The direct kernels before the Winograd routes: profiles/r03_direct_pmc_summary.json",
}
print(json.dumps(out, indent=1))
