import sys, time
sys.path.insert(0, '/root/repo/lstm-unet_amd')
import torch, numpy as np
import Networks, Params
from lu_native.graph import GraphedFrame
net = Params.CTCParams.net_kernel_params
for prec in ('fp32', 'bf16'):
    torch.manual_seed(0)
    frames = [torch.randn(1, 1, 1, 256, 256) for _ in range(6)]
    m1 = Networks.ULSTMnet2D(net, 'NCHW', True, seed=0, precision=prec)
    outs = []
    for f in frames:
        outs.append(m1(f, training=False)[1].clone())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): m1(frames[i % 6], training=False)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 20
    m2 = Networks.ULSTMnet2D(net, 'NCHW', True, seed=0, precision=prec)
    g = GraphedFrame(m2, frames[0])
    g.reset_states()
    err = 0.0
    for f, o in zip(frames, outs):
        sm = g(f)[1]
        err = max(err, float((sm - o).abs().max()))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): g(frames[i % 6])
    torch.cuda.synchronize(); tg = (time.perf_counter() - t0) / 20
    print(prec, 'eager %.2f ms/frame (%.1f fps)   graph %.2f ms/frame (%.1f fps)   max |softmax diff| %.2e' % (te*1e3, 1/te, tg*1e3, 1/tg, err))
