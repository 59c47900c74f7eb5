# round 5, fourth GPU call: A/Bs on the bf16 step (same box, alternating): (1) LU_WGRAD_F_XREALIGN (re-aligned x-fragment reads instead
# of funnel shifts / register moves in the 5x5 weight gradient), (2) look-ahead depth of the third-generation halo loop (builds with
# -DLU_G3_LA=2 / 5 under abl_tmp/ against the product's 3)
tag=${1:-r05d}
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels.py -q -x -m gpu -k "realigned or wgrad_bf16" 2>&1 | tail -2
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1', d['ms_per_step'], d['step_tflops_achieved_per_gpu'], [(c['kernel'][:34], c['frac'], c['ms_per_step']) for c in d['roofline']['all_mfma_kernels'][:3]])"; }
B="python bench.py --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-infer --no-variants --no-wgrad-overlap"
for i in 1 2 3; do
$B 2>/dev/null | line "base       "
$B --wgrad-flags 32768 2>/dev/null | line "xrealign   "
$B --lib abl_tmp/liblstmunet_g3la2.so 2>/dev/null | line "g3 LA=2    "
$B --lib abl_tmp/liblstmunet_g3la5.so 2>/dev/null | line "g3 LA=5    "
done 2>&1 | tee gpurun_out/${tag}_ab.log
