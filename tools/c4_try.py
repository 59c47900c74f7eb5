"""A/B of dispatch policies on a training step of a given shape.  usage: python tools/c4_try.py H W T B [precision]  (GPU)"""
import sys, time
sys.path.insert(0, '/root/repo')
sys.path.insert(0, '/root/repo/lstm-unet_amd')
import torch
import Params, train2D
import bench
from lu_native import ops, calls, dp as dpmod

H, W, T, B = [int(v) for v in sys.argv[1:5]]
prec = sys.argv[5] if len(sys.argv) > 5 else 'fp32'
dev = torch.device('cuda', 0)
net = Params.CTCParams.net_kernel_params
batches = bench.synthetic_batches(2, B, T, H, W, 0, dev)
orig_splits = calls.conv_splits


def old_splits(frames, Hout, Wout, N, k, channels, halo=True):
    tiles = -(-(frames * Hout * Wout) // 256) * -(-N // 128)
    n_it = k * k * -(-channels // 16)
    if tiles >= 384 or n_it < 64:
        return 1
    return int(max(1, min(512 // tiles, n_it // 32, 16)))


for label, overlap in (('no side stream', False), ('side stream', True)):
    fmt, splits = None, orig_splits
    ops.FUSED_MIN_TILES = fmt
    calls.conv_splits = splits
    tr = train2D.Trainer(Params.CTCParams.net_model, net, 'NCHW', Params.CTCParams.class_weights, Params.CTCParams.learning_rate,
                         dp=dpmod.DataParallel() if hasattr(dpmod, 'DataParallel') else None, seed=0, precision=prec)
    tr.engine.overlap_wgrad = overlap

    def step(i):
        img, seg, keep = batches[i % 2]
        tr.train_step(img, seg, want_outputs=True)
        tr.model.reset_states_per_batch(keep)
    step(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(2):
        step(1 + i)
    torch.cuda.synchronize()
    print('%-22s %.1f ms/step  peak %.1f GB' % (label, (time.perf_counter() - t0) / 2 * 1e3, torch.cuda.max_memory_allocated() / 1e9), flush=True)
    del tr
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
