# round 6, second GPU call: same-box A/B of the branch-free stage cursor in the bf16 kernel-row weight gradient (abl_tmp/lib_r05.so = the round-5
# library) -- micro-benchmark (tools/wgbench.py), the bf16 / bf16x3 steps, counters of the new loop, and the LDS-DMA staging form again
tag=${1:-r06b}
R=$GRAFT_REPO_ROOT
: > gpurun_out/${tag}_ab.log
for rep in 1 2; do
  KB_LIB=abl_tmp/lib_r05.so python tools/wgbench.py r05 2>&1 | grep wgrad_bf16 | tee -a gpurun_out/${tag}_ab.log
  python tools/wgbench.py new 2>&1 | grep wgrad_bf16 | tee -a gpurun_out/${tag}_ab.log
done
WG_FLAGS=4096 WG_SHAPES='L0 5x5,L1 5x5' python tools/wgbench.py new-dma 2>&1 | grep wgrad_bf16 | tee -a gpurun_out/${tag}_ab.log
WG_FLAGS=32 WG_SHAPES='L0 5x5,L1 5x5' python tools/wgbench.py new-prb32 2>&1 | grep wgrad_bf16 | tee -a gpurun_out/${tag}_ab.log
for rep in 1 2; do for lib in abl_tmp/lib_r05.so lstm-unet_amd/csrc/liblstmunet_hip.so; do for prec in bf16 bf16x3; do
  python bench.py --lib $lib --precision $prec --steps 8 --warmup 3 --no-cpu-baseline --no-variants --no-infer 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('step $prec $lib', d['ms_per_step'], d['build_id'], [(r['kernel'][:28], r['ms_per_step'], r['frac']) for r in d['roofline']['all_mfma_kernels'][:4]])" | tee -a gpurun_out/${tag}_ab.log
done; done; done
export WG_SHAPES='L0 5x5,L1 5x5'
args=""
for v in r05 new; do
  lib=$R/lstm-unet_amd/csrc/liblstmunet_hip.so; [ $v = r05 ] && lib=$R/abl_tmp/lib_r05.so
  dirs=""; i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAVE_CYCLES"; do
    i=$((i+1)); d=$R/gpurun_out/pmcabl_${v}_$i; rm -rf $d
    (cd /tmp && export TMPDIR=/tmp && KB_LIB=$lib rocprofv3 --kernel-trace --pmc $grp -d $d -- python $R/tools/wgbench.py pmc > /dev/null 2>&1)
    dirs="$dirs,$d"
  done
  args="$args $v=${dirs#,}"
done
python tools/pmc_abl.py gpurun_out/${tag}_ab.log wgrad_row_bf16_kernel $args
rm -rf gpurun_out/pmcabl_*
