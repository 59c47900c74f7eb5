# round 5, third GPU call: (1) kernel + engine tests of the two new kernel forms (chunk-aligned fp32 K split, third-generation bf16
# halo loop), (2) same-box A/B of generation 3 vs generation 2 (LU_CONV_F_LOOP_GEN2 = 32768) on the bf16 step of the three nets,
# (3) same-box A/B of the chunk-aligned K split vs the counted loop (LU_CONV_F_SPLIT_TAPS = 16384) on the fp32 step,
# (4) config-4 with two warm-up steps (allocator pool grown before the timed region)
tag=${1:-r05c}
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_kernels.py -q -x -m gpu -k "conv or lstm" > gpurun_out/${tag}_kernel_tests.log 2>&1; tail -3 gpurun_out/${tag}_kernel_tests.log
timeout 1500 python -m pytest tests/test_engine.py tests/test_dp_equivalence.py tests/test_drivers.py tests/test_bench_contract.py -q -x -s -m gpu > gpurun_out/${tag}_engine_tests.log 2>&1; tail -3 gpurun_out/${tag}_engine_tests.log; grep -E "dp8|dp2 vs|train_step_parity c1" gpurun_out/${tag}_engine_tests.log | head
bash tools/gpu/r04_ab_flags.sh ${tag}_gen3_vs_gen2 32768 0 params lstm3 default5
for i in 1 2; do for fl in 0 16384; do
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-infer --no-variants --no-bf16 --conv-flags $fl 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fp32 conv-flags=$fl', d['ms_per_step'], d['step_tflops_achieved_per_gpu'], [(c['kernel'][:34], c['frac'], c['ms_per_step']) for c in d['roofline']['all_mfma_kernels'][:8]])"
done; done 2>&1 | tee gpurun_out/${tag}_ksplit_ab.log
python bench.py --hw 832 992 --batch 2 --unroll 16 --steps 3 --warmup 2 --no-bf16 --no-infer --no-cpu-baseline --no-variants > gpurun_out/${tag}_f32_c4_bench_line.json 2> gpurun_out/${tag}_c4.err
python -c "
import json; d=json.load(open('gpurun_out/${tag}_f32_c4_bench_line.json')); print('c4 warmup 2, 3 steps:', d['ms_per_step'], d['step_tflops_achieved_per_gpu'], d['peak_hbm_gb'], d.get('allocator'))"
