"""Per-kernel micro-benchmark at config-2 layer shapes (HIP events on the launch stream).
usage: python tools/kbench.py [tag]   -- env vars select kernel variants (see lu_conv.hip / lu_wgrad.hip)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd'))
import torch
from lu_native import ops
if os.environ.get('KB_LIB'):           # e.g. lstm-unet_amd/csrc/liblstmunet_abl.so (python -m lu_native.build --ablation)
    ops.LIB_PATH = os.path.abspath(os.environ['KB_LIB'])

if os.environ.get('KB_CONV_FLAGS'):    # LU_CONV_F_* bits OR-ed into every conv descriptor
    ops.CONV_FLAGS = int(os.environ['KB_CONV_FLAGS'], 0)
dev = torch.device('cuda', 0)
tag = sys.argv[1] if len(sys.argv) > 1 else ''
which = os.environ.get('KB', 'fwd,dgrad,wgrad').split(',')


def timeit(fn, flops, name, reps=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print('%-8s %-34s %8.3f ms  %6.1f TFLOP/s' % (tag, name, ms, flops / ms / 1e9), flush=True)


def r(*s, scale=1.0):
    return torch.randn(*s, device=dev) * scale

levels = [('L0', 256, 1, 128), ('L1', 128, 128, 256), ('L2', 64, 256, 256), ('L3', 32, 256, 512)]
B, k = 4, 5
for name, hw, cin, F in levels:
    if 'fwd' in which:
        x, h, c = r(B, hw, hw, cin), r(B, hw, hw, F, scale=0.5), r(B, hw, hw, F)
        kx, kh, b = r(k, k, cin, 4 * F, scale=0.05), r(k, k, F, 4 * F, scale=0.02), r(4 * F)
        ho, co, g = torch.empty_like(h), torch.empty_like(c), torch.empty(B, hw, hw, 4 * F, device=dev)
        fl = 2.0 * k * k * (cin + F) * 4 * F * hw * hw * B
        timeit(lambda: ops.convlstm_step(x, h, c, kx, kh, b, ho, co, g), fl, 'lstm_step_fused ' + name)
        if 'bf16' in which:
            pk, ph = ops.pack_bf16(kx), ops.pack_bf16(kh)
            ho2, co2 = torch.empty_like(h), torch.empty_like(c)
            xp = x if cin % 4 == 0 else torch.cat([x, torch.zeros(B, hw, hw, 4 - cin % 4, device=dev)], -1)
            timeit(lambda: ops.convlstm_step(xp, h, c, pk, ph, b, ho2, co2, g), fl, 'lstm_step_fused_bf16 ' + name)
            print('   max |h_bf16 - h_f32| = %.3e  (|h| max %.3f)' % ((ho2 - ho).abs().max().item(), ho.abs().max().item()))
    if 'dgrad' in which:
        dz = r(B, hw, hw, 4 * F)
        kh = r(k, k, F, 4 * F, scale=0.02)
        wt = ops.flip_transpose(kh)
        out = torch.empty(B, hw, hw, F, device=dev)
        p = (k - 1) // 2
        fl = 2.0 * k * k * F * 4 * F * hw * hw * B
        timeit(lambda: ops.conv_raw([(dz, wt)], B, hw, hw, hw, hw, k, 1, 1, p, p, F, None, out), fl, 'rec_dgrad ' + name)
        if 'bf16' in which:
            pw = ops.pack_bf16(wt)
            timeit(lambda: ops.conv_raw([(dz, pw)], B, hw, hw, hw, hw, k, 1, 1, p, p, F, None, out), fl, 'rec_dgrad_bf16 ' + name)
    if 'wgrad' in which:
        T = 8
        xs, dy = r(T * B, hw, hw, F), r(T * B, hw, hw, 4 * F)
        dw = torch.empty(k, k, F, 4 * F, device=dev)
        fl = 2.0 * k * k * F * 4 * F * hw * hw * B * T
        timeit(lambda: ops.conv2d_wgrad(xs, dy, dw, 1), fl, 'rec_wgrad ' + name, reps=2)
        if 'bf16' in which:
            dw2 = torch.empty_like(dw)
            timeit(lambda: ops.conv2d_wgrad(xs, dy, dw2, 1, bf16=True), fl, 'rec_wgrad_bf16 ' + name, reps=2)
            print('   max |dw_bf16 - dw_f32| / max|dw| = %.3e' % ((dw2 - dw).abs().max().item() / dw.abs().max().item()))
        del xs, dy

if 'tape16' in which:    # the bf16-tape kernels exactly as the engine drives them (bf16 sources, bf16 gates / h copies)
    for name, hw, cin, F in levels:
        if os.environ.get('KB_LEVEL') and name not in os.environ['KB_LEVEL'].split(','):
            continue
        xs = r(B, hw, hw, cin)
        h, c = r(B, hw, hw, F, scale=0.5), r(B, hw, hw, F)
        kx, kh, b = r(k, k, cin, 4 * F, scale=0.05), r(k, k, F, 4 * F, scale=0.02), r(4 * F)
        ph = ops.pack_bf16(kh)
        if cin % 8:
            x16, pk, ctr = ops.im2col_bf16(xs, k), ops.pack_center_bf16(kx), True
        else:
            x16, pk, ctr = ops.to_bf16(xs), ops.pack_bf16(kx), False
        h16 = ops.to_bf16(h)
        ho, co = torch.empty_like(h), torch.empty_like(c)
        g16 = torch.empty(B, hw, hw, 4 * F, device=dev, dtype=torch.bfloat16)
        h16o = torch.empty_like(h16)
        fl = 2.0 * k * k * (cin + F) * 4 * F * hw * hw * B
        timeit(lambda: ops.convlstm_step(x16, h16, c, pk, ph, b, ho, co, g16, h16_out=h16o, x_center=ctr), fl,
               'step_tape16 ' + name, reps=10)
        wt = ops.pack_bf16(ops.flip_transpose(kh))
        out = torch.empty(B, hw, hw, F, device=dev)
        p = (k - 1) // 2
        fl = 2.0 * k * k * F * 4 * F * hw * hw * B
        timeit(lambda: ops.conv_raw([(g16, wt)], B, hw, hw, hw, hw, k, 1, 1, p, p, F, None, out), fl, 'rec_dgrad_tape16 ' + name,
               reps=10)

if 'misc' in which:     # the layers outside the halo kernels' domain (config-2 shapes, all 32 frames at once)
    FR = 32
    for name, hw, cin, n, kk_, st in [('D0.conv0 s2', 256, 128, 128, 3, 2), ('D1.conv0 s2', 128, 256, 256, 3, 2),
                                      ('U2.conv0 256->64', 128, 256, 64, 3, 1), ('U2.conv1 64->64', 128, 64, 64, 3, 1),
                                      ('U3.conv1 32->32', 256, 32, 32, 3, 1), ('U3.conv2 1x1 32->3', 256, 32, 3, 1, 1)]:
        x, w, b = r(FR, hw, hw, cin), r(kk_, kk_, cin, n, scale=0.05), r(n)
        ho = -(-hw // st)
        fl = 2.0 * kk_ * kk_ * cin * n * ho * ho * FR
        timeit(lambda: ops.conv2d([(x, w)], b, st), fl, 'conv ' + name)
        if 'bf16' in which:
            pw = ops.pack_bf16(w)
            timeit(lambda: ops.conv2d([(x, pw)], b, st), fl, 'conv_bf16 ' + name)
        if st == 2:
            dy = r(FR, ho, ho, n)
            timeit(lambda: ops.conv2d_dgrad(dy, w, (hw, hw), 2), fl, 'dgrad ' + name)
            if 'bf16' in which:
                timeit(lambda: ops.conv2d_dgrad(dy, w, (hw, hw), 2, bf16=True), fl, 'dgrad_bf16 ' + name)
        del x
