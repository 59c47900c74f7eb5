"""Per-tile fixed cost of the bf16 5x5 halo kernel: the same tiles (256^2, 4 frames, N = 128: 512 tiles of 16 x 32 pixels = two
rounds of one tile per CU) at several channel counts -- time per launch = rounds * (stages * a + b).  usage: python tools/tile_fit.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd'))
import torch
from lu_native import ops
if os.environ.get('KB_LIB'):      # e.g. an ablation build (python -m lu_native.build --ablation 8: no epilogue)
    ops.LIB_PATH = os.path.abspath(os.environ['KB_LIB'])
dev = torch.device('cuda', 0)
k, hw, B, N = 5, 256, 4, 128
F32 = os.environ.get('TF_PREC', 'bf16') == 'fp32'      # fp32: conv_halo_kernel<5,BIAS>, 8 x 32 tiles, two per CU
res = []
FRAMES = tuple(int(v) for v in os.environ.get('TF_FRAMES', '4,2').split(','))
for frames in FRAMES:
    for C in ((64, 128, 256, 512) if F32 else (128, 256, 512, 1024)):
        x = torch.randn(frames, hw, hw, C, device=dev)
        w = torch.randn(k, k, C, N, device=dev) * 0.02
        if not F32:
            x, w = x.to(torch.bfloat16), ops.pack_bf16(w)
        out = torch.empty(frames, hw, hw, N, device=dev)
        fn = lambda: ops.conv_raw([(x, w)], frames, hw, hw, hw, hw, k, 1, 1, 2, 2, N, None, out)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        reps = 5 if F32 else 20
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        us = 1e3 * a.elapsed_time(b) / reps
        stages = 25 * C // (16 if F32 else 32)
        tiles = frames * (hw // (8 if F32 else 16)) * (hw // 32) / (2.0 if F32 else 1.0)      # (fp32: two tiles per CU at a time)
        print('frames %d C %4d: %8.1f us  tiles %d (%.1f per CU), %d stages per tile -> %.1f us per tile-round' % (
            frames, C, us, tiles, tiles / 256.0, stages, us / (tiles / 256.0)), flush=True)
        res.append((frames, C, us, stages, tiles / 256.0))
for frames in FRAMES:
    pts = [(s, us / r) for f, c, us, s, r in res if f == frames]
    (s0, t0), (s1, t1) = pts[0], pts[-1]
    a_ = (t1 - t0) / (s1 - s0)
    print('frames %d: a = %.4f us per stage, b = %.1f us per tile (from the smallest and the largest C); mid points predicted %s measured %s' % (
        frames, a_, t0 - a_ * s0, [round(a_ * s + t0 - a_ * s0, 1) for s, _ in pts[1:-1]], [round(t, 1) for _, t in pts[1:-1]]))
