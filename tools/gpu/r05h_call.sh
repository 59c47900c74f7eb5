# round 5, eighth GPU call: precision 'bf16x3' -- weight gradients with the terms as frames (two launches instead of six, bias gradient riding):
# tests on the MI355X, accuracy against the fp32 engine, same-box A/B against the six-launch form and against the fp32 step
tag=${1:-r05h}
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_kernels.py tests/test_engine.py -q -x -m gpu -k "split6 or bf16x3" -s 2>&1 | grep "split6\|passed\|failed\|rror" | tail -12
timeout 600 python tools/x3_compare.py > gpurun_out/${tag}_x3_compare.json 2> gpurun_out/${tag}_x3_compare.err; python - <<PY
import json
d=json.load(open('gpurun_out/${tag}_x3_compare.json'))
for k in ('bf16x3_vs_fp32','bf16_vs_fp32'):
    for r in d[k]: print(k, r['window'], 'logits %.2e loss-rel %.2e argmax-disagree %.2e worst-grad-L2 %.2e %s median %.2e' % (r['logits_max_abs_diff_over_max'], r['loss_rel_diff'], r['argmax_disagree_fraction'], r['worst_grad_l2_rel'][0], r['worst_grad_l2_rel'][1], r['median_grad_l2_rel']))
PY
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$1', d['ms_per_step'], d['value'], d['step_tflops_achieved_per_gpu'], d['peak_hbm_gb'], [(c['kernel'][:30], c['frac'], c['ms_per_step'], c['launches_per_step']) for c in r['all_mfma_kernels'][:3]], [(c['kernel'][:14], c['ms_per_step']) for c in r['hbm_kernels'][:2]])"; }
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-infer --no-variants --no-bf16 --no-x3"
for i in 1 2; do
$B --precision bf16x3 2>gpurun_out/${tag}_err1.log | line "x3 two launches "
$B --precision bf16x3 --ab-x3-wgrad6 2>/dev/null | line "x3 six launches "
$B 2>/dev/null | line "fp32            "
done 2>&1 | tee gpurun_out/${tag}_ab.log
tail -2 gpurun_out/${tag}_err1.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_x3 -- python $R/bench.py --precision bf16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-variants > /dev/null 2>&1
cd $R && python tools/prof_summary.py gpurun_out/${tag}_prof_x3 gpurun_out/${tag}_x3_kernel_stats 60 | head -16; rm -rf gpurun_out/${tag}_prof_x3
