"""Host-side profile of the streaming-inference frame (B = 1, T = 1, 256x256 + reflect pad): where the launch thread's time goes.
usage: python tools/inf_hostprof.py [fp32|bf16] [frames]"""
import cProfile, pstats, sys, os, time, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd')); sys.path.insert(0, ROOT)
import torch
import Params, Networks
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device('cuda', 0)
m = Networks.ULSTMnet2D(Params.CTCParams.net_kernel_params, 'NCHW', True, seed=0, precision=prec)
frames = [torch.randn(1, 1, 1, 256, 256, device=dev) for _ in range(4)]
for i in range(5):
    m(frames[i % 4], training=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    m(frames[i % 4], training=False)
t_host = time.perf_counter() - t0          # launch thread only (no sync)
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('%s: %d frames, host launch time %.3f ms/frame, wall %.3f ms/frame (%.1f frames/s)' % (prec, n, 1e3 * t_host / n, 1e3 * t_all / n, n / t_all))
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    m(frames[i % 4], training=False)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(28)
print(s.getvalue()[:6000])

# ---- the same with the post-processing pipeline (Inference2D.PostPipeline) ----
import numpy as np
import Inference2D
from DataHandeling import SyntheticSequence2D
prov = SyntheticSequence2D(image_crop_size=(256, 256), unroll_len=1, batch_size=1, data_format='NCHW', seed=7, rank=0)
seg = prov.get_batch()[1][0, 0, 0]
seg = np.where(seg < 0, 0, seg).astype(np.int64)
fake = torch.from_numpy(np.eye(3, dtype=np.float32)[seg].transpose(2, 0, 1) * 0.9 + 0.03).to(dev).contiguous()
pipe = Inference2D.PostPipeline(2, 10, 10 ** 6)
pipe.push(-2, fake); pipe.push(-1, fake); pipe.flush()
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    m(frames[i % 4], training=False)
    pipe.push(i, fake)
t_host = time.perf_counter() - t0
pipe.flush(); torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('%s + post: host %.3f ms/frame, wall %.3f ms/frame (%.1f frames/s)' % (prec, 1e3 * t_host / n, 1e3 * t_all / n, n / t_all))
pr = cProfile.Profile()
pr.enable()
for i in range(n):
    pipe.push(i, fake)
pr.disable()
pipe.flush(); torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14)
print(s.getvalue()[:3500])
