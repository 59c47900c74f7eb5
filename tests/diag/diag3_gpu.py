"""GPU diagnostic 3 (round 5): is the HIP path's distance from the fp64 gradients at config-1 the distance of ANY fp32
implementation?  Runs the c1 training step of tests/test_engine.py::test_train_step_parity three ways -- fp64 oracle, torch-fp32
oracle, HIP engine -- and prints, side by side: forward noise (logits, carried h / c per level) and every gradient tensor's
max-abs / tensor-max and L2-relative error against fp64.
usage: python tests/diag/diag3_gpu.py [c1|k5-odd]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'lstm-unet_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import torch_oracle as tho
from test_engine import perturbed_params, rel_err, grad_floor, make_engine, to_tb, from_tb, GPU_CASES
from lu_native import ops

which = sys.argv[1] if len(sys.argv) > 1 else 'c1'
name, net, cin, B, T, H, W, pad = next(c for c in GPU_CASES if c[0] == which)
dev = torch.device('cuda', 0)
ops.FUSED_MIN_TILES = 0       # as the test's autouse fixture
rng = np.random.default_rng(5)
p = perturbed_params(net, cin, 7)
x = rng.standard_normal((B, T, H, W, cin)).astype(np.float32)
gt = rng.integers(-1, 3, size=(B, T, H, W)).astype(np.float32)
cw = [0.15, 0.25, 0.6]
torch.set_num_threads(16)
t64 = tho.TorchULSTM(net, cin, p, dtype=torch.float64)
l64, lg64, g64 = t64.train_step(x, gt, cw, apply=False)
t32 = tho.TorchULSTM(net, cin, p, dtype=torch.float32)
l32, lg32, g32 = t32.train_step(x, gt, cw, apply=False)
e = make_engine(net, p, cin, dev, False)
lg = e.forward(torch.from_numpy(to_tb(x)).to(dev), T, B, True)
g = torch.from_numpy(to_tb(gt[..., None])).to(dev).view(-1)
cwt = torch.tensor(cw, dtype=torch.float32, device=dev)
sums, _ = ops.wce_forward(lg.view(-1, 3), g, cwt, False)
e.backward(ops.wce_backward(lg.view(-1, 3), g, cwt, sums, 1.0).view(lg.shape))
torch.cuda.synchronize()
got = from_tb(lg.cpu().numpy(), B, T)
ref = lg64.numpy()
print('%s: loss fp64 %.8f torch32 %.8f hip %.8f' % (name, float(l64), float(l32), float(ops.wce_loss(sums).cpu()[0])))
print('forward noise vs fp64   logits: hip %.3e  torch32 %.3e   (max|logit| %.3f)' %
      (np.abs(got - ref).max(), np.abs(lg32.numpy() - ref).max(), np.abs(ref).max()))
for bi, (b64, b32, bh) in enumerate(zip(t64.states, t32.states, e.states)):
    for li, ((h64, c64), (h32, c32), (hh, ch)) in enumerate(zip(b64, b32, bh)):
        print('   level %d layer %d   h: hip %.3e torch32 %.3e (rms hip %.3e torch32 %.3e)   c: hip %.3e torch32 %.3e' % (
            bi, li, np.abs(hh.cpu().numpy() - h64.numpy()).max(), np.abs(h32.numpy() - h64.numpy()).max(),
            np.sqrt(np.mean((hh.cpu().numpy() - h64.numpy()) ** 2)), np.sqrt(np.mean((h32.numpy() - h64.numpy()) ** 2)),
            np.abs(ch.cpu().numpy() - c64.numpy()).max(), np.abs(c32.numpy() - c64.numpy()).max()))
fl = grad_floor({k: v.numpy() for k, v in g64.items()})


def l2rel(a, r, k):
    return float(np.linalg.norm(a.astype(np.float64) - r) / max(np.linalg.norm(r), fl * (3.0 if '.conv.' in k and k.endswith('.bias') else 1.0)))


rows = []
for k in g64:
    r = g64[k].numpy()
    a, b = e.G[k].cpu().numpy(), g32[k].numpy()
    rows.append((rel_err(a, r, fl), l2rel(a, r, k), rel_err(b, r, fl), l2rel(b, r, k), k))
print('%-36s %10s %10s | %10s %10s' % ('tensor', 'hip max', 'hip L2', 't32 max', 't32 L2'))
for hm, hl, tm_, tl, k in sorted(rows, reverse=True):
    print('%-36s %10.3e %10.3e | %10.3e %10.3e' % (k, hm, hl, tm_, tl))
print('worst: hip %.3e / %.3e   torch32 %.3e / %.3e;  median: hip %.3e / %.3e  torch32 %.3e / %.3e' % (
    max(r[0] for r in rows), max(r[1] for r in rows), max(r[2] for r in rows), max(r[3] for r in rows),
    sorted(r[0] for r in rows)[len(rows) // 2], sorted(r[1] for r in rows)[len(rows) // 2],
    sorted(r[2] for r in rows)[len(rows) // 2], sorted(r[3] for r in rows)[len(rows) // 2]))
