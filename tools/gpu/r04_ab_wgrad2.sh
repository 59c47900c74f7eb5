tag=${1:-r04w3}
for net in params lstm3; do
for args in "--wgrad-flags 2048" "" "--wgrad-rounds 3" "--wgrad-rounds 8" "--wgrad-flags 2048" ""; do
python bench.py --net $net --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-infer --no-variants --no-wgrad-overlap $args 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$net [$args]', d['ms_per_step'], d['step_tflops_achieved_per_gpu'], [(c['kernel'][:26], c['frac'], c['ms_per_step']) for c in d['roofline']['all_mfma_kernels'] if 'wgrad_row_bf16_kernel' in c['kernel']], [(c['kernel'][:22], c['ms_per_step']) for c in d['roofline']['hbm_kernels'] if 'reduce' in c['kernel']])"
done
done 2>&1 | tee gpurun_out/${tag}.log
