# LDS bank-conflict counters of the half path's contraction kernels (run on the GPU box): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per kernel
# over tools/ab_hgemm_zero.py's launches.   usage: bash tools/pmc_lds_conflicts.sh r04
set -e
export TMPDIR=/tmp
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/pmcl
cd $R/tools
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmcl -- python3 ab_hgemm_zero.py > /dev/null 2> $O/${TAG}_pmc_lds.err
python3 - <<PY > $O/${TAG}_pmc_lds_conflicts.txt
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float))
for f in glob.glob("$O/pmcl/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "hgemm" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
print("LDS bank conflicts of the half path's contraction kernels (tools/ab_hgemm_zero.py launches; SQ_LDS_BANK_CONFLICT = extra LDS cycles, SQ_LDS_IDX_ACTIVE = all LDS-array cycles)")
for k, v in sorted(acc.items()):
    c, a = v.get("SQ_LDS_BANK_CONFLICT", 0.0), v.get("SQ_LDS_IDX_ACTIVE", 0.0)
    print(f"{k[:110]:110s} conflict cycles {c:14.0f}  active {a:14.0f}  ratio {c / a if a else 0:.4f}")
PY
rm -rf $O/pmcl
cat $O/${TAG}_pmc_lds_conflicts.txt
