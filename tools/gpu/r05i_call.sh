# round 5, ninth GPU call: precision 'bf16x3' with the fused gate backward; the full-size oracle tests of the mode (fp32 tolerances); same-box
# step times; streaming inference of the mode
tag=${1:-r05i}
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_kernels.py tests/test_engine.py -q -x -m gpu -k "split or bf16x3" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -x -m gpu -k "bf16x3" -s 2>&1 | grep -v "^$\|amdgpu" | tail -60 > gpurun_out/${tag}_fullsize_x3.log; tail -25 gpurun_out/${tag}_fullsize_x3.log
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$1', d['ms_per_step'], d['value'], d['step_tflops_achieved_per_gpu'], d['peak_hbm_gb'], d.get('inference'), [(c['kernel'][:30], c['frac'], c['ms_per_step'], c['launches_per_step']) for c in r['all_mfma_kernels'][:3]], [(c['kernel'][:22], c['ms_per_step']) for c in r['hbm_kernels'][:3]])"; }
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-bf16 --no-x3"
for i in 1 2; do
$B --precision bf16x3 2>gpurun_out/${tag}_err1.log | line "x3   "
$B 2>/dev/null | line "fp32 "
done 2>&1 | tee gpurun_out/${tag}_ab.log
tail -2 gpurun_out/${tag}_err1.log
