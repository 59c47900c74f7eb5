# round 5: do the kernel-form choices of bf16 mode still hold at bf16x3's six-fold reduction depth?  existing flags only, same box, alternating
cd $GRAFT_REPO_ROOT
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('$1', d['ms_per_step'], [(c['kernel'][:26], c['frac'], c['ms_per_step']) for c in r['all_mfma_kernels'][:3]])"; }
B="python bench.py --precision bf16x3 --steps 5 --warmup 2 --no-cpu-baseline --no-variants --no-infer"
$B 2>/dev/null | line "default                          "
$B --conv-flags 8192 2>/dev/null | line "conv: half blocks everywhere     "
$B --conv-flags 2 2>/dev/null | line "conv: 16x32 patches              "
$B --conv-flags 1 2>/dev/null | line "conv: 8x32 patches               "
$B --conv-flags 32768 2>/dev/null | line "conv: second loop generation     "
$B 2>/dev/null | line "default                          "
$B --wgrad-flags 4096 2>/dev/null | line "wgrad: LDS-DMA staging           "
$B --wgrad-flags 32768 2>/dev/null | line "wgrad: re-aligned x reads        "
$B --wgrad-flags 65536 2>/dev/null | line "wgrad: half blocks               "
$B --wgrad-flags 32 2>/dev/null | line "wgrad: 32-pixel stages           "
$B 2>/dev/null | line "default                          "
