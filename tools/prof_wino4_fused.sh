# counter passes over tools/prof_wino4_fused.py (on the GPU box): bash tools/prof_wino4_fused.sh <tag>
set -e
export TMPDIR=/tmp
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
i=0
for pmc in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $O/pw_$i
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $O/pw_$i -- python3 tools/prof_wino4_fused.py > /dev/null 2> $O/${TAG}_pw_$i.err || { echo "pass $i failed"; tail -3 $O/${TAG}_pw_$i.err; continue; }
  python3 tools/prof_wino4_fused.py sum $O/pw_$i > $O/${TAG}_wino4_fused_pmc_$i.txt
  rm -rf $O/pw_$i
  echo "pass $i done"
done
rm -rf $O/pw_t
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pw_t -- python3 tools/prof_wino4_fused.py > /dev/null 2> $O/${TAG}_pw_t.err
f=$(find $O/pw_t -name "*kernel_stats.csv" | head -1)
python3 tools/summarize_rocprof.py $f $O/${TAG}_wino4_fused_kernel_stats.csv
rm -rf $O/pw_t
cat $O/${TAG}_wino4_fused_pmc_*.txt
cat $O/${TAG}_wino4_fused_kernel_stats.csv
