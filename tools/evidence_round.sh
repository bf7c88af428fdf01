# Everything profiles/<tag>_* is made from, in one GPU-box call:  bash tools/evidence_round.sh r03
#   <tag>_bench_default.json                 python bench.py (headline + secondary workloads + cpu_baseline)
#   <tag>_bench_<workload>_kernel_stats.csv  rocprofv3 --kernel-trace --stats of python bench.py --workload W  (+ _under_rocprof.json)
#   <tag>_pmc_{FETCH,WRITE}_SIZE_w{32,16}.json  separate --pmc passes over the cifar20 launch mix
#   <tag>_train_step_kernel_stats.csv        rocprofv3 of tools/prof_train.py
#   <tag>_bench_bf16_<workload>.json         bf16-operand mode lines (not the headline)
#   <tag>_ab_winograd{,_wgrad}.txt           tools/ab_winograd.py, tools/ab_winograd_wgrad.py
#   <tag>_attention.txt                      tools/bench_attention.py
#   <tag>_full_coalition.txt                 two real end-to-end coalitions
# Two halves (a gpurun call is limited to 20 minutes):  bash tools/evidence_round.sh r04 a   then   ... r04 b
set -e
TAG=${1:-r04}
HALF=${2:-ab}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
if [[ $HALF == *a* ]]; then
python3 bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err
echo "done default bench"
bash tools/prof_round.sh $TAG "cifar20 cifar20-pruned sd256 sd512 celeba celeba-pruned" pmc
fi
if [[ $HALF != *b* ]]; then exit 0; fi
rm -rf $O/prof_train
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python3 tools/prof_train.py 20 > $O/${TAG}_train_step.txt 2>&1
python3 tools/summarize_rocprof.py $(find $O/prof_train -name "*kernel_stats.csv" | head -1) $O/${TAG}_train_step_kernel_stats.csv
rm -rf $O/prof_train
for wl in cifar20 sd256 sd512; do
  st=20; [ $wl != cifar20 ] && st=5
  timeout -k 10 200 python3 bench.py --workload $wl --precision bf16 --steps $st --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_bf16_${wl}.json 2> $O/${TAG}_bench_bf16_${wl}.err
  echo "done bf16 $wl"
done
timeout -k 10 300 python3 tools/ab_winograd.py > $O/${TAG}_ab_winograd.txt 2>&1
timeout -k 10 300 python3 tools/ab_winograd_wgrad.py > $O/${TAG}_ab_winograd_wgrad.txt 2>&1
timeout -k 10 600 python3 tools/bench_attention.py 5 > $O/${TAG}_attention.txt 2>&1
echo "done attention"
# complete coalitions end to end: four with two in flight per GPU (train(0) | sample(0) + train(1) | ... | sample(3): the first training and the
# last sampling phase run alone), then two strictly one after the other
timeout -k 10 700 python3 bench.py --full-coalition --steps 4 --warmup 0 --no-cpu-baseline --no-kernel-timing > $O/${TAG}_full_coalition.txt 2>&1
echo "done full coalition (two in flight)"
timeout -k 10 400 python3 bench.py --full-coalition --steps 2 --warmup 0 --no-cpu-baseline --no-kernel-timing --in-flight 1 > $O/${TAG}_full_coalition_one_stream.txt 2>&1
echo "done full coalition (one stream)"
