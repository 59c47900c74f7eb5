# round 5: timing probe for the fp16-piece split: the bf16 kernels with every v_mfma_f32_32x32x16_bf16 replaced by
# v_mfma_f32_32x32x16_f16 on the same bits (abl_tmp/liblstmunet_f16probe.so, tools/build_f16_probe.py; results are garbage) next to
# the product library, alternating, bf16 mode and bf16x3 mode
cd $GRAFT_REPO_ROOT
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$1', d['ms_per_step'], [(c['kernel'][:34], c['frac'], c['ms_per_step']) for c in r['all_mfma_kernels'][:3]])"; }
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --no-bf16 --no-x3 --no-infer"
for i in 1 2; do
$B --precision bf16 2>/dev/null | line "bf16 mode, bf16 MFMA  "
$B --precision bf16 --lib abl_tmp/liblstmunet_f16probe.so 2>/dev/null | line "bf16 mode, fp16 MFMA  "
$B --precision bf16x3 2>/dev/null | line "bf16x3 mode, bf16 MFMA"
$B --precision bf16x3 --lib abl_tmp/liblstmunet_f16probe.so 2>/dev/null | line "bf16x3 mode, fp16 MFMA"
done 2>&1 | tee gpurun_out/r05p_f16_probe.log
