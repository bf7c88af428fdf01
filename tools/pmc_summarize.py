"""Average a PMC counter over the launches of one kernel from a rocprofv3 `--pmc X --kernel-trace` run.
usage: pmc_summarize.py <dir with *_counter_collection.csv> <kernel name substring[|substring...]> <counter> -> JSON on stdout.
Counter rows are summed per dispatch (one row per XCD / instance) and then averaged over dispatches."""
import csv
import glob
import json
import sys
from collections import defaultdict

d, kern, counter = sys.argv[1], sys.argv[2], sys.argv[3]
files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
per = defaultdict(float)
for f in files:
    for r in csv.DictReader(open(f)):
        if any(k in r["Kernel_Name"] for k in kern.split("|")) and r["Counter_Name"] == counter:
            per[(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
vals = list(per.values())
print(json.dumps({"kernel": kern, "counter": counter, "launches": len(vals),
                  "avg": sum(vals) / max(1, len(vals)), "min": min(vals) if vals else None, "max": max(vals) if vals else None}))
