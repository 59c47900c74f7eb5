cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels.py -m gpu -x -q -k "conv or convlstm or bf16" 2>&1 | tail -3 | tee gpurun_out/r04n_ktests.log
{
for i in 1 2; do
for lib in tools/ab/liblstmunet_old.so lstm-unet_amd/csrc/liblstmunet_hip.so; do
for net in params lstm3; do
python bench.py --net $net --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-infer --no-variants --no-wgrad-overlap --lib $lib 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$net $lib', d['ms_per_step'], d['step_tflops_achieved_per_gpu'], [(c['kernel'][:40], c['frac'], c['ms_per_step']) for c in d['roofline']['all_mfma_kernels'][:7]])"
done
done
done
} 2>&1 | tee gpurun_out/r04n_ab.log
