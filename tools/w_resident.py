"""Does the WEIGHT stream cost the fp32 halo kernel time?  The same launch (config-2 L1 recurrent input gradient: 4 x 128 x 128, 1024 -> 256,
5x5) with (a) its real 26 MB kernel, (b) one [C, N] matrix for all 25 taps (tap stride 0: 1 MB, L2-resident), (c) one 16-row chunk for
everything (row stride real, 16 KB: L1-resident is not expressible, so (b) only) -- identical instruction streams, only the addresses differ.
usage: [KB_LIB=...] python tools/w_resident.py [tag]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd'))
import torch
from lu_native import ops
if os.environ.get('KB_LIB'):
    ops.LIB_PATH = os.path.abspath(os.environ['KB_LIB'])
dev = torch.device('cuda', 0)
tag = sys.argv[1] if len(sys.argv) > 1 else ''
k, hw, B = 5, 128, 4


def timeit(fn, flops, name, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print('%-8s %-44s %8.3f ms  %6.1f TFLOP/s' % (tag, name, ms, flops / ms / 1e9), flush=True)


for C, N in ((1024, 256), (256, 256), (256, 1024)):
    x = torch.randn(B, hw, hw, C, device=dev)
    w = torch.randn(k, k, C, N, device=dev) * 0.02
    w1 = (torch.randn(C, N, device=dev) * 0.02)[None, None].expand(k, k, C, N)
    x1 = torch.randn(1, hw, hw, C, device=dev).expand(B, hw, hw, C)
    out = torch.empty(B, hw, hw, N, device=dev)
    fl = 2.0 * k * k * C * N * hw * hw * B
    for _ in range(2):
        timeit(lambda: ops.conv_raw([(x, w)], B, hw, hw, hw, hw, k, 1, 1, 2, 2, N, None, out), fl, 'C %d N %d real weights' % (C, N))
        timeit(lambda: ops.conv_raw([(x, w1)], B, hw, hw, hw, hw, k, 1, 1, 2, 2, N, None, out), fl, 'C %d N %d one matrix for all taps' % (C, N))
        timeit(lambda: ops.conv_raw([(x1, w1)], B, hw, hw, hw, hw, k, 1, 1, 2, 2, N, None, out), fl, 'C %d N %d ... and one frame for all' % (C, N))
