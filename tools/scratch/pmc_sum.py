import csv, glob, sys, collections
d = sys.argv[1]
per = collections.defaultdict(float)
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"][:70], r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
for k in sorted(per, key=lambda k: int(k[1])):
    if "wino" in k[0] or "conv3x3" in k[0]: print(k[1], k[0][:64], k[2], f"{per[k]:.4g}")
