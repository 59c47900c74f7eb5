"""Calibration of the gradient tolerances: how far is an INDEPENDENT fp32 implementation -- the torch
oracle run in fp32 on the CPU -- from the fp64 oracle's gradients, tensor by tensor, on the inputs the GPU gradient tests use?
The GPU tests (tests/test_fullsize_gpu.py::test_*_every_gradient_*) run the same two oracle passes next to the HIP step and
state their tolerance as a multiple of THIS error, not of the HIP measurement.  CPU only: usable in the build container.

    python tools/grad_calibrate.py [--hw 64 64] [--T 4] [--B 2] [--seed 53] [--out profiles/r05_grad_calibration_64.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'lstm-unet_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import torch_oracle as tho      # noqa: E402  (test infrastructure: tools/ and tests/ only)


def host_params(net, in_channels=1, seed=0):
    """The engine's own initial weights (lu_native.plan: Keras-default initialisers, seeded) without a device."""
    from lu_native.plan import make_plan, param_specs, bn_stat_specs, init_tensor
    plan = make_plan(net, in_channels)
    gen = torch.Generator().manual_seed(seed)
    p = {}
    for name, shape, kind in param_specs(plan):
        p[name] = init_tensor(shape, kind, gen).numpy().copy()
    for name, shape, kind in bn_stat_specs(plan):
        p[name] = init_tensor(shape, kind, None).numpy().copy()
    return p


def grad_rows(grads, grads_ref):
    """Per-tensor (max-abs / tensor-max, L2-relative, name, tensor max) with the floors of the GPU tests."""
    gmax = max(float(np.abs(np.asarray(v)).max()) for v in grads_ref.values())
    floor = 1e-3 * gmax
    rows = []
    for k, gr in grads_ref.items():
        a = np.asarray(grads[k], dtype=np.float64)
        r_ = np.asarray(gr, dtype=np.float64)
        scale = max(float(np.abs(r_).max()), floor)
        l2s = max(float(np.linalg.norm(r_)), floor * (3.0 if '.conv.' in k and k.endswith('.bias') else 1.0))
        rows.append((float(np.abs(a - r_).max()) / scale, float(np.linalg.norm(a - r_)) / l2s, k, float(np.abs(r_).max())))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--hw', type=int, nargs=2, default=[64, 64])
    ap.add_argument('--T', type=int, default=4)
    ap.add_argument('--B', type=int, default=2)
    ap.add_argument('--seed', type=int, default=53)
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    if a.threads:
        torch.set_num_threads(a.threads)
    import Params
    import oracle_farm as of      # tests/oracle_farm.py: the input generator of the GPU gradient tests
    net = Params.CTCParams.net_kernel_params
    H, W = a.hw
    x, gt, _, _ = of.make_inputs(dict(B=a.B, T=a.T, H=H, W=W, seed=a.seed))
    cw = [0.15, 0.25, 0.6]
    p = host_params(net)
    t0 = time.time()
    l64, _, g64 = tho.TorchULSTM(net, 1, p, dtype=torch.float64).train_step(x, gt, cw, apply=False)
    t1 = time.time()
    l32, _, g32 = tho.TorchULSTM(net, 1, p, dtype=torch.float32).train_step(x, gt, cw, apply=False)
    t2 = time.time()
    rows = grad_rows({k: v.numpy() for k, v in g32.items()}, {k: v.numpy() for k, v in g64.items()})
    print('torch fp32 vs fp64 oracle, %dx%d T=%d B=%d: loss %.7f / %.7f, fp64 %.1f s, fp32 %.1f s' %
          (H, W, a.T, a.B, float(l32), float(l64), t1 - t0, t2 - t1))
    for mr, l2, k, gm in sorted(rows, reverse=True)[:12]:
        print('   %-38s max-rel %.3e  L2-rel %.3e  (tensor max %.3e)' % (k, mr, l2, gm))
    print('   worst max-rel %.3e, worst L2-rel %.3e' % (max(r[0] for r in rows), max(r[1] for r in rows)))
    if a.out:
        with open(a.out, 'w') as f:
            json.dump({'hw': [H, W], 'T': a.T, 'B': a.B, 'seed': a.seed, 'loss_fp64': float(l64), 'loss_fp32': float(l32),
                       'rows': [{'tensor': k, 'max_rel': mr, 'l2_rel': l2, 'tensor_max': gm} for mr, l2, k, gm in rows]}, f, indent=1)


if __name__ == '__main__':
    main()
