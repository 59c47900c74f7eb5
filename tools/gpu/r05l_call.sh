# round 5, twelfth GPU call: precision 'bf16x3' at config-4 with split convolutions on every level (fp32 weight gradients where W % 32 != 0);
# the ragged-geometry gradient test of the mode; config-2 / config-5-shape step times of the final kernels
tag=${1:-r05l}
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -x -m gpu -k "ragged and bf16x3" -s 2>&1 | grep -v "^$\|amdgpu" | tail -14 | cut -c1-260
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$1', d['ms_per_step'], d['value'], d['step_tflops_achieved_per_gpu'], d['peak_hbm_gb'], d['allocator'], [(c['kernel'][:30], c['frac'], c['ms_per_step'], c['launches_per_step']) for c in r['all_mfma_kernels'][:5]], [(c['kernel'][:22], c['ms_per_step']) for c in r['hbm_kernels'][:3]])"; }
C4="--hw 832 992 --batch 2 --unroll 16 --steps 3 --warmup 2 --no-bf16 --no-x3 --no-infer --no-cpu-baseline --no-variants"
timeout 900 python bench.py $C4 --precision bf16x3 2>gpurun_out/${tag}_c4_x3.err | tee gpurun_out/${tag}_x3_c4_bench_line.json | line "c4 x3  "
tail -3 gpurun_out/${tag}_c4_x3.err
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-variants --no-bf16 --no-x3 --no-infer"
$B --precision bf16x3 2>/dev/null | line "c2 x3  "
$B --precision bf16x3 --size 512 --batch 2 2>/dev/null | line "512 x3 "
$B --size 512 --batch 2 2>/dev/null | line "512 f32"
