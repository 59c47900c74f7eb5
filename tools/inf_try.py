import sys, time
sys.path.insert(0, '/root/repo/lstm-unet_amd')
import torch
import Networks, Params
from lu_native import ops
net = Params.CTCParams.net_kernel_params
for prec in ('fp32', 'bf16'):
    for fmt in (160, 100, 60, 0):
        ops.FUSED_MIN_TILES = fmt
        m = Networks.ULSTMnet2D(net, 'NCHW', True, seed=0, precision=prec)
        frames = [torch.randn(1, 1, 1, 256, 256, device='cuda') for _ in range(4)]
        for i in range(3): m(frames[i % 4], training=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20): m(frames[i % 4], training=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(prec, 'FUSED_MIN_TILES', fmt, '%.2f ms/frame  %.1f fps' % (dt * 1e3, 1 / dt), flush=True)
        del m
