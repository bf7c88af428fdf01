# Evidence for the half-precision activation path (run on the GPU box): bench lines, rocprofv3 kernel stats, HBM counters of the
# dominant convolution kernel, the engine's A/B sweeps.   usage: bash tools/evidence_half.sh r04
set -e
export TMPDIR=/tmp
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
python3 bench.py --workload sd512 --precision bf16 --steps 10 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_sd512_bf16.json 2> $O/${TAG}_bench_sd512_bf16.err
python3 bench.py --workload sd256 --precision bf16 --steps 10 --warmup 2 --no-cpu-baseline > $O/${TAG}_bench_sd256_bf16.json 2> $O/${TAG}_bench_sd256_bf16.err
echo "bench lines done"
rm -rf $O/prof_h
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_h -- python3 bench.py --workload sd512 --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/${TAG}_bench_sd512_bf16_under_rocprof.json 2> $O/${TAG}_prof_sd512_bf16.err
python3 tools/summarize_rocprof.py $(find $O/prof_h -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_sd512_bf16_kernel_stats.csv
rm -rf $O/prof_h
echo "kernel stats done"
rm -rf $O/pmch; mkdir -p $O/pmch
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmch/pmc_$c -- python3 bench.py --workload sd512 --precision bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $O/${TAG}_pmc_${c}_sd512_bf16.err
  echo "pmc $c done"
done
python3 tools/pmc_route_summary.py $O/pmch "sd512 --precision bf16" "" "hgemm_kernel<4, 2, 2, 5, 32, 4, true, 1, true>" > $O/${TAG}_pmc_summary_sd512_bf16_conv.json
# (the dominant family of the sd512 bf16 line: bench.py's bracket spans delta + dQ + dK/dV; Tk = 77 launches of the same dQ kernel are in the sum)
python3 tools/pmc_route_summary.py $O/pmch "sd512 --precision bf16" "attn_bwd_d40_4096_4096_2" "attn_bwd_dkv_bf16_kernel<40, true, 2>" "attn_bwd_dq_bf16_kernel<40, true, 2>" "attn_delta_h_kernel<40>" > $O/${TAG}_pmc_summary_sd512_bf16_attn_bwd.json
rm -rf $O/pmch
cd tools
python3 ab_hgemm.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_ab_hgemm.txt
python3 ab_hgemm_zero.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_hgemm_zero_dma.txt
echo "all done"
