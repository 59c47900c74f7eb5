"""GPU diagnostic 2: isolate the BN/LeakyReLU backward of down.2.conv.1 inside a real C1 backward."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'lstm-unet_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
from oracle import np_oracle as npo, torch_oracle as tho
from conftest import c1_net
from test_engine import perturbed_params, make_engine, to_tb
from lu_native import ops, engine as eng_mod

dev = torch.device('cuda', 0)
net = c1_net(); cin = 1; B, T, H, W = 1, 4, 128, 128
rng = np.random.default_rng(5)
p = perturbed_params(net, cin, 7)
x = rng.standard_normal((B, T, H, W, cin)).astype(np.float32)
gt = rng.integers(-1, 3, size=(B, T, H, W)).astype(np.float32)
cwt = torch.tensor([0.15, 0.25, 0.6], dtype=torch.float32, device=dev)
e = make_engine(net, p, cin, dev, False)
orig = eng_mod.Engine._conv_unit_backward
cap = {}
def patched(self, rec, dz, need_dx):
    key = '%s.%d' % (rec['prefix'], rec['ci'])
    if rec['bn'] and rec['prefix'].startswith('down'):
        y = rec['y']
        d64, y64 = dz.double(), y.double()
        sc, sh, mean, inv = [rec[k].double() for k in ('scale', 'shift', 'mean', 'invstd')]
        z = y64 * sc + sh
        dzp = d64 * torch.where(z > 0, 1.0, 0.3)
        xhat = (y64 - mean) * inv
        s0 = dzp.sum(dim=(0, 1, 2)); s1 = (dzp * xhat).sum(dim=(0, 1, 2))
        n = y.numel() // y.shape[-1]
        dx_ref = sc * (dzp - s0 / n - xhat * s1 / n)
        ksums = ops.bn_lrelu_bwd_reduce(y, dz, rec['scale'], rec['shift'], rec['mean'], rec['invstd'], 0.3)
        C = y.shape[-1]
        print(key, 'rows', n, 'C', C, 'count', rec['count'],
              's0 err %.3e (max %.3e)  s1 err %.3e (max %.3e)' % (float((ksums[:C] - s0).abs().max()), float(s0.abs().max()),
                                                                float((ksums[C:] - s1).abs().max()), float(s1.abs().max())))
        cap[key] = dx_ref
        srcs = rec['srcs']
        out = orig(self, rec, dz, need_dx)
        # dz was overwritten in place with dy
        print('   dy(dx of BN) err %.3e (max %.3e)' % (float((dz.double() - dx_ref).abs().max()), float(dx_ref.abs().max())))
        gw = self.G['%s.conv.%d.kernel' % (rec['prefix'], rec['ci'])]
        xin = srcs[0][0].double()
        # reference wgrad via torch conv on GPU in fp64
        import torch.nn.functional as F
        from lu_native.calls import same_pad
        k = gw.shape[0]; s = rec['spec']['stride']
        _, pt, pb = same_pad(xin.shape[1], k, s); _, pl, pr = same_pad(xin.shape[2], k, s)
        xn = F.pad(xin.permute(0, 3, 1, 2), (pl, pr, pt, pb)).requires_grad_(False)
        w64 = self.P['%s.conv.%d.kernel' % (rec['prefix'], rec['ci'])].double().permute(3, 2, 0, 1).contiguous().requires_grad_(True)
        yy = F.conv2d(xn, w64, None, stride=s)
        (gref,) = torch.autograd.grad(yy, [w64], dx_ref.permute(0, 3, 1, 2).contiguous())
        gref = gref.permute(2, 3, 1, 0)
        print('   wgrad err %.3e (max %.3e)' % (float((gw.double() - gref).abs().max()), float(gref.abs().max())))
        return out
    return orig(self, rec, dz, need_dx)
eng_mod.Engine._conv_unit_backward = patched
lg = e.forward(torch.from_numpy(to_tb(x)).to(dev), T, B, True)
g = torch.from_numpy(to_tb(gt[..., None])).to(dev).view(-1)
sums, _ = ops.wce_forward(lg.view(-1, 3), g, cwt, False)
e.backward(ops.wce_backward(lg.view(-1, 3), g, cwt, sums, 1.0).view(lg.shape))
torch.cuda.synchronize()
