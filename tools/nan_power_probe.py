"""Is a kernel's time DATA-dependent on this part (power-limited clocks)?  The bf16 training step on the usual synthetic clips against the
same step on all-NaN images (every MFMA operand / accumulator then holds one constant bit pattern: minimal switching).  Context: the
fp16-MFMA timing probe (tools/build_f16_probe.py) runs on garbage numerics, so its 9 % gain had to be told apart from this effect.
usage: python tools/nan_power_probe.py [precision] [lib.so]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import Params  # noqa: E402
import train2D  # noqa: E402
from lu_native import ops, build as lu_build  # noqa: E402


def main(precision='bf16', lib=None):
    if lib:
        ops.LIB_PATH = lu_build.LIB = os.path.abspath(lib)
    dev = torch.device('cuda', 0)
    B, T, H, W = 4, 8, 256, 256
    rng = np.random.default_rng(0)
    seg = torch.from_numpy(rng.integers(-1, 3, size=(B, T, 1, H, W)).astype(np.float32)).to(dev)
    for what in ('random images', 'all-NaN images', 'all-zero images', 'random images'):
        tr = train2D.Trainer(Params.CTCParams.net_model, Params.CTCParams.net_kernel_params, 'NCHW', Params.CTCParams.class_weights,
                             Params.CTCParams.learning_rate, seed=0, precision=precision)
        img = torch.randn(B, T, 1, H, W, device=dev)
        if 'NaN' in what:
            img = img * float('nan')
        elif 'zero' in what:
            img = img * 0.0
        for _ in range(2):
            tr.train_step(img, seg, want_outputs=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            tr.train_step(img, seg, want_outputs=True)
        torch.cuda.synchronize()
        print('%-8s %-18s %.2f ms / step' % (precision, what, 1e3 * (time.perf_counter() - t0) / 4), flush=True)
        del tr
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main(*sys.argv[1:3])
