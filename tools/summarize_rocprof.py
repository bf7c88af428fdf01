"""Condense a rocprofv3 --kernel-trace --stats CSV into profiles/<name>_kernel_stats.csv (top kernels)."""
import csv, re, sys
src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "percent"])
    for r in rows[:40]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
        n = re.sub(r"^void ", "", n)[:110]
        w.writerow([n, r["Calls"], f"{float(r['TotalDurationNs'])/1e6:.3f}", f"{float(r['AverageNs'])/1e3:.2f}",
                    f"{float(r['MinNs'])/1e3:.2f}", f"{float(r['MaxNs'])/1e3:.2f}", f"{float(r['TotalDurationNs'])/tot*100:.2f}"])
    w.writerow(["TOTAL", sum(int(r["Calls"]) for r in rows), f"{tot/1e6:.3f}", "", "", "", "100.00"])
print("wrote", dst)
