# A/B of lu_conv_desc / lu_wgrad_desc flag bits on the bf16 step, alternating runs inside one call
# usage: bash tools/gpu/r04_ab_flags.sh <tag> <conv-flags> [wgrad-flags] [nets...]
tag=${1:-r04ab}; cf=${2:-8192}; wf=${3:-0}; shift 3
nets=${@:-params lstm3}
for net in $nets; do
for i in 1 2; do
for fl in 0 1; do
if [ $fl = 1 ]; then a="--conv-flags $cf --wgrad-flags $wf"; else a=""; fi
python bench.py --net $net --precision bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-infer --no-variants --no-wgrad-overlap $a 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$net flags=$fl', d['ms_per_step'], d['step_tflops_achieved_per_gpu'], [(c['kernel'][:40], c['frac'], c['ms_per_step']) for c in d['roofline']['all_mfma_kernels'][:6]])"
done
done
done 2>&1 | tee gpurun_out/${tag}.log
