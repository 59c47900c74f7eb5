"""A/B of the streaming-inference + post-processing pipeline on the GPU box: who issues the device -> host copy, and with what
the forward shares the chip.  usage: python tools/post_ab.py [fp32|bf16]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd')); sys.path.insert(0, ROOT)
import numpy as np, torch
import Networks, Params, Inference2D
from DataHandeling import SyntheticSequence2D
from lu_native import post
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
dev = torch.device('cuda', 0)
net = Params.CTCParams.net_kernel_params
m = Networks.ULSTMnet2D(net, 'NCHW', True, seed=0, precision=prec)
H = W = 256
frames = [torch.randn(1, 1, 1, H, W, device=dev) for _ in range(4)]
prov = SyntheticSequence2D(image_crop_size=(H, W), unroll_len=1, batch_size=1, data_format='NCHW', seed=7, rank=0)
seg = prov.get_batch()[1][0, 0, 0]
seg = np.where(seg < 0, 0, seg).astype(np.int64)
fake = torch.from_numpy(np.eye(3, dtype=np.float32)[seg].transpose(2, 0, 1) * 0.9 + 0.03).to(dev).contiguous()
def run(n, pipe):
    for i in range(3):
        m(frames[i % 4], training=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        m(frames[i % 4], training=False)
        if pipe is not None:
            for _ in pipe.push(i, fake):
                pass
    if pipe is not None:
        for _ in pipe.flush():
            pass
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)
print(prec, 'forward only', round(run(40, None), 1))
for lib_copy in (True, False, True, False):
    pipe = Inference2D.PostPipeline(2, 10, 10 ** 6)
    pipe.push(-1, fake); pipe.flush()
    for p in pipe._procs:
        p.library_copy = lib_copy
    print(prec, 'library_copy', lib_copy, round(run(40, pipe), 1))
# post alone, synchronous
pp = post.PostProcessor()
pp(fake, 2, 10, 10 ** 6)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(40):
    pp(fake, 2, 10, 10 ** 6)
print('post alone, synchronous: %.3f ms/frame' % ((time.perf_counter() - t0) / 40 * 1e3))
# a full Fluo-C2DL-MSC-size frame with hundreds of objects (BASELINE config-4 frame size): device-driven route and, for a map
# with nested objects, the exact sequential replay
from oracle import postprocess_oracle as po      # (tool: the synthetic softmax generator lives beside the oracle)
import json
big = {}
for nested in (False, True):
    sm = torch.from_numpy(po.synthetic_softmax(832, 992, seed=1 + nested, n_cells=400, nested=nested, noise=0.25, rmax=16)).to(dev)
    pp = post.PostProcessor()
    lab = pp(sm, 2, 10, 5000)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10):
        lab = pp(sm, 2, 10, 5000)
    dt = (time.perf_counter() - t0) / 10 * 1e3
    big['nested' if nested else 'plain'] = {'ms_per_frame': round(dt, 3), 'objects': int(lab.max()), 'sequential_replays': pp.fallbacks,
                                            'frames': 11}
    print('832x992, %d objects, nested=%s: %.3f ms/frame synchronous, sequential replays %d of 11' % (lab.max(), nested, dt, pp.fallbacks))
if len(sys.argv) > 2:
    json.dump({'what': 'GPU post-processing of one 832x992 frame (softmax on the device -> uint16 label map on the host), synchronous',
               'cases': big}, open(sys.argv[2], 'w'), indent=1)
