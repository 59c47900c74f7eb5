# where do the wave-cycles of the hot kernels go?  SQ issue / wait / instruction-mix / LDS counters, each group in its own rocprofv3 --pmc pass
# usage (GPU box, repo root): bash tools/gpu/pmc_sq.sh r04     -> gpurun_out/<tag>_pmc_sq.json
tag=${1:-r04}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
args=""
for mode in fp32 bf16 bf16x3; do
  dirs=""
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM" \
             "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_BRANCH"; do
    i=$((i+1)); d=$R/gpurun_out/pmcsq_${mode}_$i; rm -rf $d
    rocprofv3 --kernel-trace --pmc $grp -d $d -- python $R/bench.py --precision $mode --steps 1 --warmup 1 --no-cpu-baseline --no-infer --no-bf16 --no-x3 --no-variants > /dev/null 2>&1
    dirs="$dirs,$d"
  done
  args="$args $mode=${dirs#,}"
done
cd $R && python tools/pmc_sq.py gpurun_out/${tag}_pmc_sq.json $args 2>&1 | tail -12
rm -rf gpurun_out/pmcsq_*
