# HBM-side traffic per kernel launch (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes) and MFMA utilisation, for bench.py's step in both precisions.
# usage (on the GPU box, from the repo root): bash tools/gpu/pmc_traffic.sh r02     -> gpurun_out/<tag>_pmc_traffic[_bf16].json
#        PMC_EXTRA="--net lstm3" PMC_SFX=_lstm3 [PMC_MODES="bf16"] bash tools/gpu/pmc_traffic.sh r05   -> ..._pmc_traffic[_bf16]_lstm3.json, ..._pmc_mfma_util_lstm3.json
#        (another workload: extra bench.py arguments + the table suffix bench.py looks for: _<net>, _c4, _c5shape)
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in ${PMC_MODES:-fp32 bf16 bf16x3}; do
  sfx=""; [ $mode != fp32 ] && sfx="_$mode"
  sfx="$sfx$PMC_SFX"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_${mode}_$ctr
    rocprofv3 --kernel-trace --pmc $ctr -d $R/gpurun_out/pmc_${mode}_$ctr -- python $R/bench.py --precision $mode --steps 1 --warmup 1 --no-cpu-baseline --no-infer --no-bf16 --no-x3 --no-variants $PMC_EXTRA > /dev/null 2>&1
  done
  python $R/tools/pmc_summary.py $R/gpurun_out/pmc_${mode}_FETCH_SIZE $R/gpurun_out/pmc_${mode}_WRITE_SIZE $R/gpurun_out/${tag}_pmc_traffic$sfx.json
  rm -rf $R/gpurun_out/pmc_${mode}_FETCH_SIZE $R/gpurun_out/pmc_${mode}_WRITE_SIZE
  rm -rf $R/gpurun_out/pmc_${mode}_mfma
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_${mode}_mfma -- python $R/bench.py --precision $mode --steps 1 --warmup 1 --no-cpu-baseline --no-infer --no-bf16 --no-x3 --no-variants $PMC_EXTRA > /dev/null 2>&1
done
cd $R && python tools/pmc_mfma.py gpurun_out/pmc_fp32_mfma gpurun_out/pmc_bf16_mfma gpurun_out/pmc_bf16x3_mfma gpurun_out/${tag}_pmc_mfma_util$PMC_SFX.json 2>&1 | tail -18
rm -rf gpurun_out/pmc_fp32_mfma gpurun_out/pmc_bf16_mfma gpurun_out/pmc_bf16x3_mfma
