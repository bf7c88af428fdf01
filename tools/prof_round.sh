set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
for wl in cifar20 cifar20-pruned sd256 sd512; do
  steps=20; [ "$wl" = sd256 ] && steps=5; [ "$wl" = sd512 ] && steps=5
  rm -rf $O/prof_$wl
  (cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -- python3 bench.py --workload $wl --steps $steps --warmup 2 --no-cpu-baseline --no-train-rate > $O/r2_prof_bench_$wl.json 2> $O/r2_prof_bench_$wl.err)
  f=$(find $O/prof_$wl -name "*kernel_stats.csv" | head -1)
  python3 $R/tools/summarize_rocprof.py $f $O/r02_bench_${wl}_kernel_stats.csv
  rm -rf $O/prof_$wl
  echo "done $wl"
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_$c
  (cd $R && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-train-rate > /dev/null 2> $O/r2_pmc_$c.err)
  python3 $R/tools/pmc_summarize.py $O/pmc_$c "conv3x3_patch_f32_kernel<32, 1, false, 128" $c > $O/r02_pmc_${c}_w32.json
  python3 $R/tools/pmc_summarize.py $O/pmc_$c "conv3x3_patch_f32_kernel<16, 1, false, 128" $c > $O/r02_pmc_${c}_w16.json
  rm -rf $O/pmc_$c
  echo "done $c"
done
cat $O/r02_pmc_*_w32.json
