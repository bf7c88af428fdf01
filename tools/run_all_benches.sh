set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out
for wl in cifar20-pruned sd256 sd512; do
  st=20; [ $wl != cifar20-pruned ] && st=5
  timeout -k 10 280 python3 bench.py --workload $wl --steps $st --warmup 2 > $O/r2_bench_${wl}_5.json 2> $O/r2_bench_${wl}_5.err
  echo "done $wl"
done
for wl in cifar20 sd256 sd512; do
  st=20; [ $wl != cifar20 ] && st=5
  timeout -k 10 200 python3 bench.py --workload $wl --precision bf16 --steps $st --warmup 2 --no-cpu-baseline > $O/r2_bench_bf16_${wl}_5.json 2> $O/r2_bench_bf16_${wl}_5.err
  echo "done bf16 $wl"
done
python3 - <<'P'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2_bench_*_5.json")):
    t=open(f).read(); j=json.loads(t[t.index('{"metric'):])
    print(f, j["value"], j["unit"], round(j["ms_per_step"],2), "frac", round(j["roofline"]["frac"],3), "path", round(j.get("path_mfma_frac",0),3))
P
