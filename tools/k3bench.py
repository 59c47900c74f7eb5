"""Micro-benchmark of the bf16 3x3 halo convolutions at the config-2 layer shapes (32 frames), product or ablation library.
usage: [KB_LIB=gpurun_out/abl/liblstmunet_ablN.so] python tools/k3bench.py [tag]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd'))
import torch
from lu_native import ops
if os.environ.get('KB_LIB'):
    ops.LIB_PATH = os.path.abspath(os.environ['KB_LIB'])
dev = torch.device('cuda', 0)
ops.CONV_FLAGS |= int(os.environ.get('KB_FLAGS', '0'))      # e.g. 2 = LU_CONV_F_PATCH16
tag = sys.argv[1] if len(sys.argv) > 1 else 'product'
frames = 32
shapes = [(128, 128, 128), (64, 256, 256), (32, 512, 512), (64, 512, 128), (128, 256, 64), (256, 64, 32), (256, 32, 32)]
if os.environ.get('KB_SHAPES'):
    shapes = [shapes[int(i)] for i in os.environ['KB_SHAPES'].split(',')]
for hw, cin, n in shapes:
    for b16 in (True, False):
        x = torch.randn(frames, hw, hw, cin, device=dev)
        if b16:
            x = x.to(torch.bfloat16)
        w = ops.pack_bf16(torch.randn(3, 3, cin, n, device=dev) * 0.05)
        bias = torch.randn(n, device=dev)
        out = torch.empty(frames, hw, hw, n, device=dev)
        fn = lambda: ops.conv_raw([(x, w)], frames, hw, hw, hw, hw, 3, 1, 1, 1, 1, n, bias, out)
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        fl = 2.0 * 9 * cin * n * hw * hw * frames
        print('%-8s conv3 %3d^2 %3d->%3d src=%s %8.1f us %7.1f TFLOP/s  (%.0f MB in+out -> %.2f TB/s)' % (
            tag, hw, cin, n, 'bf16' if b16 else 'fp32', 1e3 * ms, fl / ms / 1e9,
            (x.numel() * x.element_size() + out.numel() * 4) / 1e6, (x.numel() * x.element_size() + out.numel() * 4) / ms / 1e9), flush=True)
