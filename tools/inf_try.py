"""Streaming-inference frame rate (B = 1, 256x256) under the K-split knobs.  usage: python tools/inf_try.py   (GPU, A/B tool)"""
import sys, time
sys.path.insert(0, '/root/repo/lstm-unet_amd')
import torch
import Networks, Params
from lu_native import ops, calls
net = Params.CTCParams.net_kernel_params
for prec in ('fp32', 'bf16'):
    for fmt, cap, minit in ((None, 16, 24), (None, 32, 12), (None, 48, 8), (160, 16, 24), (0, 16, 24)):
        ops.FUSED_MIN_TILES = fmt
        calls.SPLIT_CAP, calls.SPLIT_MIN_IT = cap, minit
        m = Networks.ULSTMnet2D(net, 'NCHW', True, seed=0, precision=prec)
        frames = [torch.randn(1, 1, 1, 256, 256, device='cuda') for _ in range(4)]
        for i in range(3): m(frames[i % 4], training=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20): m(frames[i % 4], training=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(prec, 'FUSED_MIN_TILES', fmt, 'cap', cap, 'min_it', minit, '%.2f ms/frame  %.1f fps' % (dt * 1e3, 1 / dt), flush=True)
        del m
from lu_native.graph import GraphedFrame
ops.FUSED_MIN_TILES = None
calls.SPLIT_CAP, calls.SPLIT_MIN_IT = 32, 12
for prec in ('fp32', 'bf16'):
    m = Networks.ULSTMnet2D(net, 'NCHW', True, seed=0, precision=prec)
    frames = [torch.randn(1, 1, 1, 256, 256, device='cuda') for _ in range(4)]
    g = GraphedFrame(m, frames[0])
    for i in range(3): g(frames[i % 4])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): g(frames[i % 4])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(prec, 'hipGraph replay %.2f ms/frame  %.1f fps' % (dt * 1e3, 1 / dt), flush=True)
