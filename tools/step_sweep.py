"""Time one ConvLSTM step per network level, fused vs K-split + gate kernel, balanced vs m-tile-per-XCD block numbering
(fp32 / bf16).  usage: python tools/step_sweep.py [fp32|bf16]   (GPU; A/B tool for calls.conv_splits / ops.fused_step_applies)"""
import sys
sys.path.insert(0, '/root/repo/lstm-unet_amd')
import torch
from lu_native import ops, calls, cabi

SHAPES = {'inf L0': (1, 272, 272, 128, 1), 'inf L1': (1, 136, 136, 256, 128), 'inf L2': (1, 68, 68, 256, 256),
          'inf L3': (1, 34, 34, 512, 256), 'c5 L2': (2, 128, 128, 256, 256), 'c5 L3': (2, 64, 64, 512, 256),
          'trn L0': (4, 256, 256, 128, 1), 'trn L1': (4, 128, 128, 256, 128), 'trn L2': (4, 64, 64, 256, 256),
          'trn L3': (4, 32, 32, 512, 256)}


def timeit(fn, n=12):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main(prec):
    dev = 'cuda'
    bf = prec == 'bf16'
    for name, (fr, H, W, F, cin) in SHAPES.items():
        g = torch.Generator(device=dev).manual_seed(0)
        cpad = cin if cin % 4 == 0 else 4
        x = torch.randn(fr, H, W, cpad, device=dev, generator=g)
        if cpad != cin:
            x[..., cin:] = 0
        h = torch.randn(fr, H, W, F, device=dev, generator=g) * 0.5
        c = torch.randn(fr, H, W, F, device=dev, generator=g) * 0.5
        wk = torch.randn(5, 5, cpad, 4 * F, device=dev, generator=g) * 0.02
        wr = torch.randn(5, 5, F, 4 * F, device=dev, generator=g) * 0.02
        b = torch.zeros(4 * F, device=dev)
        ho, co = torch.empty_like(h), torch.empty_like(c)
        if bf:
            wk, wr = ops.pack_bf16(wk), ops.pack_bf16(wr)
        flops = 2.0 * 25 * (cin + F) * 4 * F * fr * H * W
        res = []
        for label, fmt, flags in (('fused', 0, 0), ('fused/nobal', 0, cabi.LU_CONV_F_NO_BALANCE), ('split', 10 ** 9, 0),
                                  ('split/nobal', 10 ** 9, cabi.LU_CONV_F_NO_BALANCE), ('split/nohalo', 10 ** 9, cabi.LU_CONV_F_NO_HALO)):
            ops.FUSED_MIN_TILES = ops.FUSED_MIN_TILES_BF16 = fmt
            ops.CONV_FLAGS = flags
            try:
                us = timeit(lambda: ops.convlstm_step(x, h, c, wk, wr, b, ho, co, None))
                res.append('%s %7.0f us %6.1f TF' % (label, us, flops / us / 1e6))
            except Exception as e:      # noqa
                res.append('%s FAILED %s' % (label, str(e)[:60]))
        s = calls.conv_splits(fr, H, W, 4 * F, 5, cpad + F)
        if name.startswith('inf') and not bf:      # split-count sweep of the K-split route
            ops.FUSED_MIN_TILES, sw = 10 ** 9, []
            ops.CONV_FLAGS = 0
            for s2 in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16):
                calls.FORCE_SPLITS = s2      # (part of the plan cache's key: calls._knobs)
                sw.append('%d:%.0f' % (s2, timeit(lambda: ops.convlstm_step(x, h, c, wk, wr, b, ho, co, None))))
            calls.FORCE_SPLITS = None
            print('   splits sweep (us):', ' '.join(sw), flush=True)
        print('%s %-7s s=%d model fused %5.0f split %5.0f | %s' % (prec, name, s, calls.fused_step_cost_us(fr, H, W, F, 5, cpad + F),
              calls.conv_cost_us(fr, H, W, 4 * F, 5, cpad + F, s), ' | '.join(res)), flush=True)
    ops.CONV_FLAGS = 0


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'fp32')
