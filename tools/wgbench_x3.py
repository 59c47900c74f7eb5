"""Micro-benchmark of the precision-'bf16x3' weight gradients at the config-2 ConvLSTM shapes: the piece-aware kernel (LU_WGRAD_F_PIECES3, round 6)
against the terms-as-frames form of round 5 (two launches), HIP events.  KB_LIB=<path> selects another build of the library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd'))
import torch
from lu_native import ops
if os.environ.get('WG_ROUNDS'):      # rounds of blocks the pixel axis is cut into (calls.BF16_ROW_ROUNDS; default 5)
    from lu_native import calls as _calls
    _calls.BF16_ROW_ROUNDS = int(os.environ['WG_ROUNDS'])
if os.environ.get('KB_LIB'):
    ops.LIB_PATH = os.path.abspath(os.environ['KB_LIB'])
dev = torch.device('cuda', 0)
T, B = 8, 4
for name, hw, C, F, k in [('L0 rec 5x5', 256, 128, 128, 5), ('L1 rec 5x5', 128, 256, 256, 5), ('L1 inp 5x5', 128, 128, 256, 5), ('L2 rec 5x5', 64, 256, 256, 5),
                          ('L3 rec 5x5', 32, 512, 512, 5), ('conv 3x3 128->256', 128, 128, 64, 3)]:
    N = 4 * F
    x = torch.randn(T * B, hw, hw, C, device=dev) * 0.5
    dy = torch.randn(T * B, hw, hw, N, device=dev) * 0.5
    x6, dy6 = ops.split6(x, order=0), ops.split6(dy, order=1)
    del x, dy
    dw = torch.empty(k, k, C, N, device=dev)
    fl = 6 * 2.0 * k * k * C * N * hw * hw * B * T
    out = {}
    for form in ('pieces', 'terms'):
        def fn():
            if form == 'pieces':
                ops.conv2d_wgrad(x6, dy6, dw, 1, bf16=True, terms=(0, 6), pieces=True)
            else:
                ops.conv2d_wgrad(x6, dy6, dw, 1, bf16=True, terms=(0, 3))
                ops.conv2d_wgrad(x6, dy6, dw, 1, beta=1.0, bf16=True, terms=(3, 3))
        fn(); torch.cuda.synchronize()
        out[form] = dw.clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        print('wgrad_x3 %-18s %-7s %8.3f ms  %7.1f executed bf16 TFLOP/s = %.3f of 2.5 PF (incl. slab reduce)' % (name, form, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500), flush=True)
    print('   pieces vs terms: max |diff| / max |dw| = %.3e' % float((out['pieces'] - out['terms']).abs().max() / out['terms'].abs().max()), flush=True)
    del x6, dy6
