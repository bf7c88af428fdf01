"""Print a bench.py JSON line as a per-kernel-instance table sorted by time per step."""
import json
import sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], d["unit"], "ms/step", d["ms_per_step"], "path_mfma_frac", d.get("path_mfma_frac"))
r = d.get("roofline", {})
print("dominant:", r.get("kernel"), "frac", r.get("frac"), "share", r.get("share_of_step_time"))
rows = sorted(((k, v["launches"] / d["steps"], v["avg_us"], v["launches"] * v["avg_us"] / d["steps"] / 1000, v["tflops"])
               for k, v in d["contraction_kernels"]["by_instance"].items()), key=lambda x: -x[3])
for x in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-44s n/step %6.1f avg %8.1f us  ms/step %7.2f  TF/s %7.1f" % x)
print("sum of bracketed launches, ms/step:", sum(x[3] for x in rows))
