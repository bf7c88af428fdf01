# rocprofv3 --kernel-trace --stats of `python bench.py --workload W` for each W, summarised into gpurun_out/<tag>_bench_<W>_kernel_stats.csv
# (+ the JSON line printed under the profiler).  usage (on the GPU box): bash tools/prof_round.sh r03 "cifar20 cifar20-pruned sd512 celeba celeba-pruned" [pmc]
set -e
export TMPDIR=/tmp
TAG=${1:-r03}
WLS=${2:-"cifar20 cifar20-pruned sd256 sd512 celeba celeba-pruned"}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
for wl in $WLS; do
  steps=20; case $wl in sd256|sd512|celeba|celeba-pruned) steps=5;; esac
  rm -rf $O/prof_$wl
  # (CIFAR workloads run two coalitions in flight on two streams: --no-one-stream-pass keeps the profile to the timed region)
  (cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -- python3 bench.py --workload $wl --steps $steps --warmup 2 --no-cpu-baseline --no-train-rate --no-one-stream-pass > $O/${TAG}_bench_${wl}_under_rocprof.json 2> $O/${TAG}_prof_bench_$wl.err)
  f=$(find $O/prof_$wl -name "*kernel_stats.csv" | head -1)
  python3 $R/tools/summarize_rocprof.py $f $O/${TAG}_bench_${wl}_kernel_stats.csv
  rm -rf $O/prof_$wl
  case $wl in cifar20|cifar20-pruned)       # the same on ONE stream: per-kernel durations without the other stream's share of the chip
    (cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$wl -- python3 bench.py --workload $wl --steps $steps --warmup 2 --no-cpu-baseline --no-train-rate --in-flight 1 > $O/${TAG}_bench_${wl}_one_stream_under_rocprof.json 2> $O/${TAG}_prof_bench_${wl}_one_stream.err)
    f=$(find $O/prof_$wl -name "*kernel_stats.csv" | head -1)
    python3 $R/tools/summarize_rocprof.py $f $O/${TAG}_bench_${wl}_one_stream_kernel_stats.csv
    rm -rf $O/prof_$wl;;
  esac
  echo "done $wl"
done
if [ "$3" = pmc ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/pmc_$c
    (cd $R && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 bench.py --workload cifar20 --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-train-rate --in-flight 1 > /dev/null 2> $O/${TAG}_pmc_$c.err)
    python3 $R/tools/pmc_summarize.py $O/pmc_$c "wino4_fused2_kernel" $c > $O/${TAG}_pmc_${c}_wino4_fused2.json
    python3 $R/tools/pmc_summarize.py $O/pmc_$c "gn_wino4_kernel" $c > $O/${TAG}_pmc_${c}_gn_wino4.json
    python3 $R/tools/pmc_summarize.py $O/pmc_$c "wino4_gemm_kernel|, 4, 1>(" $c > $O/${TAG}_pmc_${c}_wino_gemm.json
    python3 $R/tools/pmc_summarize.py $O/pmc_$c "wino4_input_kernel" $c > $O/${TAG}_pmc_${c}_wino_input.json
    python3 $R/tools/pmc_summarize.py $O/pmc_$c "wino4_output_kernel" $c > $O/${TAG}_pmc_${c}_wino_output.json
    python3 $R/tools/pmc_summarize.py $O/pmc_$c "wgrad3x3_patch_f32_kernel<32, 128>" $c > $O/${TAG}_pmc_${c}_wgrad32.json
    rm -rf $O/pmc_$c
    echo "done $c"
  done
  python3 $R/tools/pmc_wino_summary.py $O $TAG > $O/${TAG}_pmc_summary.json
  cat $O/${TAG}_pmc_summary.json
  # the secondary workloads' dominant families: sd512 (single-pass attention backward at T = 4096, d = 40) and celeba (three-launch F(4x4))
  for spec in "sd512|attn_bwd1_f32_kernel<40, 4" "celeba|wino4_output_kernel|wino4_gemm_kernel|, 4, 1>("; do
    wl=${spec%%|*}; rest=${spec#*|}
    rm -rf $O/pmcw; mkdir -p $O/pmcw
    for c in FETCH_SIZE WRITE_SIZE; do
      (cd $R && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmcw/pmc_$c -- python3 bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-train-rate > /dev/null 2> $O/${TAG}_pmc_${c}_$wl.err)
    done
    IFS='|' read -ra ks <<< "$rest"
    python3 $R/tools/pmc_route_summary.py $O/pmcw $wl "" "${ks[@]}" > $O/${TAG}_pmc_summary_$wl.json
    rm -rf $O/pmcw
    echo "done pmc $wl"
  done
fi
