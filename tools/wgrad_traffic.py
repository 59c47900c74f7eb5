"""One launch of the bf16 kernel-row weight gradient per (shape, split count) -- run under
`rocprofv3 --kernel-trace --pmc FETCH_SIZE` and read the per-dispatch values with tools/pmc_dispatch.py (A/B tool).
usage: python tools/wgrad_traffic.py"""
import sys
sys.path.insert(0, '/root/repo/lstm-unet_amd')
import torch
from lu_native import ops, calls

dev = 'cuda'
SHAPES = [('L0rec', 32, 256, 128, 512), ('L1rec', 32, 128, 256, 1024), ('L1ker', 32, 128, 128, 1024)]
orig = calls.wgrad_splits_bf16_row
for name, fr, hw, C, N in SHAPES:
    x = (torch.randn(fr, hw, hw, C, device=dev) * 0.5).to(torch.bfloat16)
    dy = (torch.randn(fr, hw, hw, N, device=dev) * 0.5).to(torch.bfloat16)
    dw = torch.empty(5, 5, C, N, device=dev)
    base = orig(fr * hw * hw, 5, C, N, 128)
    db = torch.zeros(N, device=dev)
    for s, bias in ((base, None), (base, db), (base * 2, db)):
        calls.wgrad_splits_bf16_row = lambda *a, s=s, **k: s
        ops.conv2d_wgrad(x, dy, dw, 1, bf16=True, dbias=bias)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.conv2d_wgrad(x, dy, dw, 1, bf16=True, dbias=bias)
        e1.record()
        torch.cuda.synchronize()
        alg = (x.numel() + dy.numel()) * 2 + dw.numel() * 4
        print('%s splits %3d dbias %s  %.3f ms  algorithmic %.0f MB' % (name, s, bias is not None, e0.elapsed_time(e1), alg / 1e6), flush=True)
    del x, dy, dw
calls.wgrad_splits_bf16_row = orig
