# round 5, eleventh GPU call: lu_split3 without fma contraction (the split image is the split of the STORED fp32 value); the test files behind the one
# that stopped the suite; config-4 in precision 'bf16x3' on the memory-lean route (832 x 992, T = 16, B = 2)
tag=${1:-r05k}
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_kernels.py tests/test_engine.py -q -x -m gpu -k "split or bf16x3" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_oracle.py tests/test_postprocess.py tests/test_tf_bundle.py tests/test_tf_pin_tool.py tests/test_tf_pinned.py -q -x -m gpu 2>&1 | tail -3
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$1', d['ms_per_step'], d['value'], d['step_tflops_achieved_per_gpu'], d['peak_hbm_gb'], d['allocator'], [(c['kernel'][:30], c['frac'], c['ms_per_step'], c['launches_per_step']) for c in r['all_mfma_kernels'][:4]], [(c['kernel'][:22], c['ms_per_step']) for c in r['hbm_kernels'][:3]])"; }
C4="--hw 832 992 --batch 2 --unroll 16 --steps 3 --warmup 2 --no-bf16 --no-x3 --no-infer --no-cpu-baseline --no-variants"
timeout 900 python bench.py $C4 --precision bf16x3 2>gpurun_out/${tag}_c4_x3.err | tee gpurun_out/${tag}_x3_c4_bench_line.json | line "c4 x3  "
tail -3 gpurun_out/${tag}_c4_x3.err
timeout 900 python bench.py $C4 2>/dev/null | tee gpurun_out/${tag}_f32_c4_bench_line.json | line "c4 fp32"
