"""Micro-benchmark of the bf16 weight gradients at the config-2 ConvLSTM shapes (bf16 tape operands), HIP events.
KB_LIB=<path> selects another build of the library (ablation builds of tools/gpu/r04_wg_ablate.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd'))
import torch
from lu_native import ops
if os.environ.get('KB_LIB'):
    ops.LIB_PATH = os.path.abspath(os.environ['KB_LIB'])
ops.WGRAD_FLAGS |= int(os.environ.get('WG_FLAGS', '0'))
dev = torch.device('cuda', 0)
tag = sys.argv[1] if len(sys.argv) > 1 else ''
T, B = 8, 4
SHAPES = [('L0 5x5', 256, 128, 5), ('L1 5x5', 128, 256, 5), ('L0 3x3', 256, 128, 3), ('L1 3x3', 128, 256, 3), ('D0.conv1 3x3', 128, 32, 3)]
if os.environ.get('WG_SHAPES'):      # e.g. WG_SHAPES='L0 5x5,L1 5x5'
    SHAPES = [s_ for s_ in SHAPES if s_[0] in os.environ['WG_SHAPES'].split(',')]
for name, hw, F, k in SHAPES:
    N = 4 * F if 'conv' not in name else 128
    C = F if 'conv' not in name else 128
    x = (torch.randn(T * B, hw, hw, C, device=dev) * 0.5).to(torch.bfloat16)
    dy = (torch.randn(T * B, hw, hw, N, device=dev) * 0.5).to(torch.bfloat16)
    dw = torch.empty(k, k, C, N, device=dev)
    fl = 2.0 * k * k * C * N * hw * hw * B * T
    fn = lambda: ops.conv2d_wgrad(x, dy, dw, 1, bf16=True)
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    print('%-8s wgrad_bf16 %-14s %8.3f ms  %7.1f TFLOP/s (incl. slab reduce)' % (tag, name, ms, fl / ms / 1e9), flush=True)
    del x, dy
