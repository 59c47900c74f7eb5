# round 6, first GPU call: (1) the driver's bench command on the start-of-round tree, (2) RCCL with ONE rank (forced collectives: test + bench
# line), (3) the vendor library's bf16 GEMM rate / clock on this box, (4) the diagnosis the round-5 verdict asks for: ablation builds of the
# bf16 kernel-row weight gradient (abl_tmp/libwg<bits>.so, cross-compiled by tools/build_wg_abl.py) timed by tools/wgbench.py, each with its
# shader clock, MFMA-busy share and SQ wait / instruction counters (two rocprofv3 --pmc passes per variant, --pmc alone with --kernel-trace)
tag=${1:-r06a}
R=$GRAFT_REPO_ROOT
python bench.py > gpurun_out/${tag}_f32_bench_line.json 2> gpurun_out/${tag}_bench.err; tail -1 gpurun_out/${tag}_bench.err
timeout 900 python -m pytest tests/test_dp_equivalence.py -q -m gpu -s -k forced > gpurun_out/${tag}_rccl_forced_test.log 2>&1; tail -4 gpurun_out/${tag}_rccl_forced_test.log
timeout 600 python bench.py --force-collectives --sync-bn --steps 5 --warmup 2 --no-variants --no-cpu-baseline --no-infer --no-bf16 --no-x3 > gpurun_out/${tag}_forced_rccl_bench_line.json 2> gpurun_out/${tag}_forced_rccl.err; tail -2 gpurun_out/${tag}_forced_rccl.err
timeout 600 python bench.py --precision bf16 --force-collectives --sync-bn --steps 8 --warmup 3 --no-variants --no-cpu-baseline --no-infer > gpurun_out/${tag}_forced_rccl_bf16_bench_line.json 2>> gpurun_out/${tag}_forced_rccl.err
python tools/gemm_ref.py 2>&1 | grep gemm_ref | tee gpurun_out/${tag}_gemm_ref.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_gemm -- python $R/tools/gemm_ref.py 5 > /dev/null 2>&1
cd $R
python - <<PY | tee -a gpurun_out/${tag}_gemm_ref.log
import sys; sys.path.insert(0, 'tools')
import pmc_mfma
for k, v in list(pmc_mfma.summarize('gpurun_out/pmc_gemm').items())[:12]:
    print('gemm_ref kernel %-90s util %5.1f %%  clock %.2f GHz  %9.1f us x %d' % (k[:90], 100 * v['mfma_utilisation'], v['shader_clock_ghz'], v['avg_duration_us'], v['launches']))
PY
rm -rf gpurun_out/pmc_gemm
export WG_SHAPES='L0 5x5,L1 5x5'
: > gpurun_out/${tag}_wgrad_bf16_ablation.txt
python tools/wgbench.py product 2>&1 | grep wgrad_bf16 | tee -a gpurun_out/${tag}_wgrad_bf16_ablation.txt
for bits in 1 2 3 4 8 16 17 18 19 31; do
  KB_LIB=abl_tmp/libwg$bits.so python tools/wgbench.py abl$bits 2>&1 | grep wgrad_bf16 | tee -a gpurun_out/${tag}_wgrad_bf16_ablation.txt
done
args=""
for v in product 1 2 3 16 18 19 31; do
  lib=""; [ $v != product ] && lib=$R/abl_tmp/libwg$v.so
  dirs=""
  i=0
  for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_DATA_FIFO_FULL SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAVE_CYCLES"; do
    i=$((i+1)); d=$R/gpurun_out/pmcabl_${v}_$i; rm -rf $d
    (cd /tmp && KB_LIB=$lib rocprofv3 --kernel-trace --pmc $grp -d $d -- python $R/tools/wgbench.py pmc > /dev/null 2>&1)
    dirs="$dirs,$d"
  done
  args="$args $v=${dirs#,}"
done
python tools/pmc_abl.py gpurun_out/${tag}_wgrad_bf16_ablation.txt wgrad_row_bf16_kernel $args
rm -rf gpurun_out/pmcabl_*
