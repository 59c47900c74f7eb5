# round 5, seventh GPU call: precision 'bf16x3' -- the gate epilogue writes the split image of h (LU_CONV_F_H16_SPLIT): tests, same-box A/B
# against the split6 pass per step, weight gradients on the side stream, then a rocprofv3 kernel trace of the bf16x3 step
tag=${1:-r05g}
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_kernels.py tests/test_engine.py -q -x -m gpu -k "split6 or bf16x3" 2>&1 | tail -2
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
print('$1', d['ms_per_step'], d['value'], d['step_tflops_achieved_per_gpu'], d['peak_hbm_gb'], [(c['kernel'][:30], c['frac'], c['ms_per_step']) for c in r['all_mfma_kernels'][:3]], [(c['kernel'][:14], c['ms_per_step']) for c in r['hbm_kernels'][:2]])"; }
B="python bench.py --precision bf16x3 --steps 6 --warmup 2 --no-cpu-baseline --no-infer --no-variants"
for i in 1 2; do
$B 2>gpurun_out/${tag}_err1.log | line "x3 fused-split          "
$B --ab-x3-split-pass 2>/dev/null | line "x3 split pass           "
$B --wgrad-overlap 2>/dev/null | line "x3 fused + wgrad overlap"
done 2>&1 | tee gpurun_out/${tag}_ab.log
tail -2 gpurun_out/${tag}_err1.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_x3 -- python $R/bench.py --precision bf16x3 --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-variants > /dev/null 2>&1
cd $R && python tools/prof_summary.py gpurun_out/${tag}_prof_x3 gpurun_out/${tag}_x3_kernel_stats 60 | head -40; rm -rf gpurun_out/${tag}_prof_x3
