"""Post-processing chain alone on a synthetic softmax (for rocprofv3 --kernel-trace): usage python tools/post_prof.py [H W]"""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/lstm-unet_amd')
import numpy as np, torch
import Inference2D
from DataHandeling import SyntheticSequence2D
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 256)
prov = SyntheticSequence2D(image_crop_size=(H, W), unroll_len=1, batch_size=1, data_format='NCHW', seed=7, rank=0)
seg = prov.get_batch()[1][0, 0, 0]
seg = np.where(seg < 0, 0, seg).astype(np.int64)
fake = torch.from_numpy(np.eye(3, dtype=np.float32)[seg].transpose(2, 0, 1) * 0.9 + 0.03).cuda().contiguous()
for _ in range(3):
    lab = Inference2D.postprocess(fake, 2, 10, 10 ** 6)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    lab = Inference2D.postprocess(fake, 2, 10, 10 ** 6)
torch.cuda.synchronize()
print('postprocess alone: %.3f ms/frame, %d objects' % ((time.perf_counter() - t0) / 20 * 1e3, lab.max()))
