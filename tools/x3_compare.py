"""precision 'bf16x3' against precision 'fp32' on the MI355X at a production geometry: one training step from the same weights,
inputs and (zero) state -- logits, loss, every gradient tensor (L2-relative and max-abs / tensor max), then a second step from
the carried state.  The claim under test is fp32 ARITHMETIC: the two engines must differ by fp32 rounding (summation order), not
by bf16 rounding.  usage: python tools/x3_compare.py [H W B T] > gpurun_out/x3_compare.json"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd'))
import Params  # noqa: E402
from lu_native import ops  # noqa: E402
from lu_native.engine import Engine  # noqa: E402


def main():
    H, W, B, T = [int(a) for a in sys.argv[1:5]] if len(sys.argv) >= 5 else (256, 256, 4, 8)
    dev = torch.device('cuda', 0)
    net = Params.CTCParams.net_kernel_params
    rng = np.random.default_rng(3)
    cwt = torch.tensor([0.15, 0.25, 0.6], dtype=torch.float32, device=dev)
    xs = [torch.from_numpy(rng.standard_normal((T * B, H, W, 1)).astype(np.float32)).to(dev) for _ in range(2)]
    gts = [torch.from_numpy(rng.integers(-1, 3, size=(T * B * H * W,)).astype(np.float32)).to(dev) for _ in range(2)]
    res = {}
    for prec in ('fp32', 'bf16x3', 'bf16'):
        e = Engine(net, pad_image=False, precision=prec, seed=0)
        e.build(1, dev)
        steps = []
        for s in range(2):
            lg = e.forward(xs[s], T, B, True)
            sums, _ = ops.wce_forward(lg.view(-1, 3), gts[s], cwt, False)
            e.backward(ops.wce_backward(lg.view(-1, 3), gts[s], cwt, sums, 1.0).view(lg.shape))
            torch.cuda.synchronize()
            steps.append((lg.double().cpu(), float(ops.wce_loss(sums).cpu()[0]), {k: v.double().cpu() for k, v in e.G.items()}))
            e.reset_states_per_batch(np.ones(B, np.float32))
        res[prec] = steps
        del e
        torch.cuda.empty_cache()
    out = {'workload': '%dx%d B=%d T=%d, Params.py widths, random init, two training windows (the second from the carried state)' % (H, W, B, T)}
    for prec in ('bf16x3', 'bf16'):
        rows = []
        for s in range(2):
            l32, loss32, g32 = res['fp32'][s]
            l, loss, g = res[prec][s]
            fl = 1e-3 * max(float(v.abs().max()) for v in g32.values())
            per = {k: (float((g[k] - g32[k]).norm() / max(float(g32[k].norm()), fl)),
                       float((g[k] - g32[k]).abs().max() / max(float(g32[k].abs().max()), fl))) for k in g32}
            wl2 = max((v[0], k) for k, v in per.items())
            wmx = max((v[1], k) for k, v in per.items())
            rows.append({'window': s, 'logits_max_abs_diff_over_max': float((l - l32).abs().max() / l32.abs().max()),
                         'loss_fp32': loss32, 'loss': loss, 'loss_rel_diff': abs(loss - loss32) / abs(loss32),
                         'argmax_disagree_fraction': float((l.argmax(-1) != l32.argmax(-1)).double().mean()),
                         'worst_grad_l2_rel': wl2, 'worst_grad_max_rel': wmx,
                         'median_grad_l2_rel': float(np.median([v[0] for v in per.values()]))})
        out[prec + '_vs_fp32'] = rows
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
