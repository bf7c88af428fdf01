"""Sum a rocprofv3 --kernel-trace --stats output directory's kernel_stats csv files: calls / total ms per kernel name prefix.
usage: python tools/count_kernels.py <dir> [name substring ...]"""
import csv
import glob
import sys

rows = {}
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        d = rows.setdefault(name, [0, 0.0])
        d[0] += int(r["Calls"])
        d[1] += float(r["TotalDurationNs"]) / 1e6
tot = sum(v[1] for v in rows.values())
for name, (n, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    if len(sys.argv) > 2 and not any(s in name for s in sys.argv[2:]):
        continue
    print(f"{name[:100]:100s} {n:7d} calls {ms:10.2f} ms {100 * ms / tot:6.2f} %")
print(f"total {tot:.2f} ms")
