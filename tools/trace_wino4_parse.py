"""Parse the kernel trace of tools/trace_wino4.py: per shape, the durations of the route's three kernels (last of three runs).
usage: trace_wino4_parse.py <kernel_trace.csv>"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if ("wino4" in r["Kernel_Name"] or ", 4, 1>" in r["Kernel_Name"]) and "weights" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
shapes = [(1024, 32, 128, 128), (1024, 32, 256, 128), (1024, 16, 256, 256), (1024, 16, 512, 256), (128, 32, 128, 128), (16, 64, 320, 320), (32, 64, 224, 224)]
i = 0
for B, H, Cin, Cout in shapes:
    last = rows[i + 6:i + 9]
    i += 9
    d, name = {}, ""
    for r in last:
        k = "in" if "input" in r["Kernel_Name"] else ("out" if "output" in r["Kernel_Name"] else "gemm")
        d[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if k == "gemm":
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:44]
    ex = 2.0 * B * H * H * Cout * 9 * Cin * 0.25
    g = d["gemm"]
    print(f"B{B} {H}x{H} {Cin}->{Cout}: input {d['in']:.0f} us, products {g:.0f} us = {ex / g / 1e6:.1f} TF/s executed ({ex / g / 1e6 / 157.3:.2f} of peak), "
          f"output {d['out']:.0f} us | {name}")
