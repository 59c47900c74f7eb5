# A/B of the all-taps 3x3 bf16 weight gradient (flags 1024 = 4 fat waves, 2048 = 8 waves) against the kernel-row form
tag=${1:-r04w}
python -m pytest tests/test_kernels.py -q -m gpu -x -k "all_taps_form or half_blocks or mfma_variant" 2>&1 | tail -2
for net in ${NETS:-params lstm3}; do
for i in 1 2; do
for wf in 0 1024 2048; do
python bench.py --net $net --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-infer --no-variants --no-wgrad-overlap --wgrad-flags $wf 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$net wgrad-flags=$wf', d['ms_per_step'], d['step_tflops_achieved_per_gpu'], [(c['kernel'][:32], c['frac'], c['ms_per_step']) for c in d['roofline']['all_mfma_kernels'] if 'wgrad_row_bf16_kernel<3' in c['kernel'] or 'LSTM' in c['kernel']])"
done
done
done 2>&1 | tee gpurun_out/${tag}.log
