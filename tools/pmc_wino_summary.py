"""profiles/pmc_summary.json from the four per-counter files of tools/prof_round.sh ... pmc: HBM bytes per launch of the Winograd pair
(wino_input_kernel + wino_gemm_kernel<64, 128>) over the cifar20 launch mix.  usage: pmc_wino_summary.py <dir> <tag>"""
import json, sys
d, tag = sys.argv[1], sys.argv[2]
ld = lambda c, k: json.load(open(f"{d}/{tag}_pmc_{c}_{k}.json"))
gf, gw, xf, xw = ld("FETCH_SIZE", "wino_gemm"), ld("WRITE_SIZE", "wino_gemm"), ld("FETCH_SIZE", "wino_input"), ld("WRITE_SIZE", "wino_input")
# FETCH_SIZE doubled (gfx950 tallies the 128-B requests of 16-B/lane streams at 64 B: MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact; KiB
gemm = (2 * gf["avg"] + gw["avg"]) * 1024
inp_all = (2 * xf["avg"] + xw["avg"]) * 1024
# wino_input_kernel also precedes the 128 x 64 GEMM instances: scale its per-launch average to the launches that pair with <64, 128>
out = {
 "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE --output-format csv (two SEPARATE passes, tools/prof_round.sh via "
           "tools/evidence_round.sh) -- python3 bench.py --workload cifar20 --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing "
           "--no-train-rate ; tools/pmc_summarize.py, tools/pmc_wino_summary.py",
 "dominant_kernel": "wino_input_kernel + wino_gemm_kernel<64, 128> (Winograd F(2x2,3x3) pair = one 3x3 convolution): every launch of the "
                    "run - sampler forward at B=1024, training forward at B=128 and the data gradients on the rotated weights",
 "wino_gemm": {"launches_per_pass": gf["launches"], "avg_FETCH_SIZE_KiB": gf["avg"], "avg_WRITE_SIZE_KiB": gw["avg"], "hbm_bytes_per_launch": gemm},
 "wino_input": {"launches_per_pass": xf["launches"], "avg_FETCH_SIZE_KiB": xf["avg"], "avg_WRITE_SIZE_KiB": xw["avg"], "hbm_bytes_per_launch": inp_all},
 "dominant_kernel_hbm_bytes_per_launch": gemm + inp_all,
 "note": "FETCH_SIZE doubled (gfx950 tallies the 128-B requests of 16-B/lane streams at 64 B), WRITE_SIZE exact; fabric-side counters include "
         "Infinity-Cache hits.  The pair moves the transformed input V (4 x the input, 16 values per 2x2 tile) out and back in: "
         "|x| + 4|x| written + 4|x| read (once: the second channel block of a tile row hits L2) + |y| against |x| + |w| + |y| algorithmic.  "
         "The direct kernels it replaced: profiles/r03_direct_pmc_summary.json",
}
print(json.dumps(out, indent=1))
