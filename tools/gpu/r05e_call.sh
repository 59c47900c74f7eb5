# round 5, fifth GPU call: same-box A/B of LU_WGRAD_F_HALF_BLOCK (65536: 4-wave / 64-channel blocks of the bf16 5x5 weight gradient, two
# independent blocks per CU) on the bf16 step of the Params and all-5x5 nets; then the instruction / wait counters of the bf16 step
tag=${1:-r05e}
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels.py -q -x -m gpu -k "realigned or wgrad_bf16" 2>&1 | tail -2
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['ms_per_step'], d['step_tflops_achieved_per_gpu'], [(c['kernel'][:34], c['frac'], c['ms_per_step']) for c in d['roofline']['all_mfma_kernels'][:3]])"; }
for net in params default5; do
B="python bench.py --net $net --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-infer --no-variants --no-wgrad-overlap"
for i in 1 2 3; do
$B 2>/dev/null | line "$net base      "
$B --wgrad-flags 65536 2>/dev/null | line "$net half-block"
done; done 2>&1 | tee gpurun_out/${tag}_ab.log
