# same-box A/B of two builds of the kernel library: tools/ab/liblstmunet_old.so (built from the previous commit's sources) against
# the in-tree one.  usage: bash tools/gpu/r04_ab_lib.sh <tag> [precision] [rounds]
tag=${1:-r04lib}; prec=${2:-fp32}; rounds=${3:-2}
mkdir -p gpurun_out
{
for i in $(seq $rounds); do
for lib in tools/ab/liblstmunet_old.so lstm-unet_amd/csrc/liblstmunet_hip.so; do
python bench.py --precision $prec --steps 4 --warmup 2 --no-cpu-baseline --no-infer --no-variants --no-bf16 --lib $lib 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', d['build_id'], d['ms_per_step'], d['step_tflops_achieved_per_gpu'], [(c['kernel'][:34], c['frac'], c['ms_per_step']) for c in d['roofline']['all_mfma_kernels'][:9]])"
done
done
} 2>&1 | tee gpurun_out/${tag}.log
