cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
{
for i in 1 2; do
for wf in 0 256; do
for net in params lstm3; do
python bench.py --net $net --steps 4 --warmup 2 --no-cpu-baseline --no-infer --no-variants --no-bf16 --wgrad-flags $wf 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$net wgrad-flags=$wf', d['ms_per_step'], d['step_tflops_achieved_per_gpu'], [(c['kernel'][:30], c['frac'], c['ms_per_step']) for c in d['roofline']['all_mfma_kernels'] if 'wgrad_row' in c['kernel']])"
done
done
done
} 2>&1 | tee gpurun_out/r04o_kp32.log
