"""Streaming inference with post-processing: frame rate over repeated runs for pipeline depths 2 / 3 (A/B, GPU)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import Params, Networks, Inference2D
from DataHandeling import SyntheticSequence2D
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dev = torch.device('cuda', 0)
m = Networks.ULSTMnet2D(Params.CTCParams.net_kernel_params, 'NCHW', True, seed=0, precision=prec)
frames = [torch.randn(1, 1, 1, 256, 256, device=dev) for _ in range(4)]
prov = SyntheticSequence2D(image_crop_size=(256, 256), unroll_len=1, batch_size=1, data_format='NCHW', seed=7, rank=0)
seg = prov.get_batch()[1][0, 0, 0]
seg = np.where(seg < 0, 0, seg).astype(np.int64)
fake = torch.from_numpy(np.eye(3, dtype=np.float32)[seg].transpose(2, 0, 1) * 0.9 + 0.03).to(dev).contiguous()
for i in range(5):
    m(frames[i % 4], training=False)
for depth in (2, 3, 2, 3, 4):
    pipe = Inference2D.PostPipeline(2, 10, 10 ** 6, depth=depth)
    for i in range(depth):
        pipe.push(-1 - i, fake)
    pipe.flush(); torch.cuda.synchronize()
    rates = []
    for rep in range(8):
        n = 60
        t0 = time.perf_counter()
        for i in range(n):
            _, sm = m(frames[i % 4], training=False)
            for _ in pipe.push(i, fake):
                pass
        for _ in pipe.flush():
            pass
        torch.cuda.synchronize()
        rates.append(n / (time.perf_counter() - t0))
    print('%s depth %d: %s  min %.0f' % (prec, depth, ' '.join('%.0f' % r for r in rates), min(rates)), flush=True)
