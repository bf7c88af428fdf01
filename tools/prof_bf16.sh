set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
for wl in sd256; do
  steps=20; [ "$wl" = sd256 ] && steps=5
  rm -rf $O/prof_bf16_$wl
  (cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16_$wl -- python3 bench.py --workload $wl --precision bf16 --steps $steps --warmup 2 --no-cpu-baseline --no-train-rate > $O/r2_prof_bench_bf16_$wl.json 2> $O/r2_prof_bench_bf16_$wl.err)
  f=$(find $O/prof_bf16_$wl -name "*kernel_stats.csv" | head -1)
  python3 $R/tools/summarize_rocprof.py $f $O/r02_bench_bf16_${wl}_kernel_stats.csv
  rm -rf $O/prof_bf16_$wl
done
