"""Second, independent CPU restatement of the ConvLSTM-UNet path, composed from torch CPU ops.
TEST INFRASTRUCTURE ONLY (see oracle/np_oracle.py header for who may import it).

Purpose:
  * cross-check the numpy fp64 oracle (two restatements written against SURVEY §8a must
    agree to <=1e-9 in fp64 before any HIP kernel is judged);
  * gradient oracle: torch autograd through this restatement gives d loss / d params and
    the post-Adam weights for whole-step parity (train2D.py:87-95);
  * ``bench.py``'s ``cpu_baseline`` leg (fp32, all host threads) -- kind "port".

PARITY STATUS: parity unpinned at the TensorFlow boundary (TensorFlow absent; the reference
ships no golden outputs).  Same semantics table as np_oracle.py.

Channels-last tensors at the API ([B,T,H,W,C]); internally torch conv2d wants NCHW, so the
helpers permute around each conv -- this file optimises for obviousness, not speed.
"""
from __future__ import annotations

import math
import torch
import torch.nn.functional as F

from . import np_oracle as npo


def _same_pad(n_in, k, s):
    return npo.tf_same_pad(n_in, k, s)


def round_bf16(t):
    """Round-to-nearest-even to bf16 through fp32 (what the bf16-mode kernels do to their MFMA operands), kept in t.dtype."""
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


def conv2d_same(x, w, b=None, stride=1, rounded=False):
    """x [N,H,W,C], w [k,k,Cin,Cout] (Keras layout) -> [N,Ho,Wo,Cout]; TF-SAME zero padding.
    rounded: both operands are rounded to bf16 first (products / sums stay in x.dtype) -- the arithmetic of the
    bf16-MFMA convolutions of Engine(precision='bf16')."""
    if rounded:
        x, w = round_bf16(x), round_bf16(w)
    k = w.shape[0]
    _, pt, pb = _same_pad(x.shape[1], k, stride)
    _, pl, pr = _same_pad(x.shape[2], k, stride)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn, w.permute(3, 2, 0, 1), b, stride=stride)
    return y.permute(0, 2, 3, 1)


def hard_sigmoid(z):
    return torch.clamp(0.2 * z + 0.5, 0.0, 1.0)


def bn_train(x, gamma, beta, eps=1e-3):
    mean = x.mean(dim=(0, 1, 2))
    var = x.var(dim=(0, 1, 2), unbiased=False)
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta, mean, var


def bn_infer(x, gamma, beta, mm, mv, eps=1e-3):
    return (x - mm) * torch.rsqrt(mv + eps) * gamma + beta


def resize_bilinear(x, f, resize='tf2.0'):
    """'half_pixel' = torch's align_corners=False; 'tf2.0' = the legacy v1 op (src = o / f): for f = 2,
    out[2i] = in[i], out[2i+1] = (in[i] + in[min(i+1, n-1)]) / 2 along each axis (see np_oracle.resize_bilinear)."""
    if f == 1:
        return x
    if resize == 'half_pixel':
        y = F.interpolate(x.permute(0, 3, 1, 2), scale_factor=f, mode='bilinear', align_corners=False)
        return y.permute(0, 2, 3, 1)
    assert resize == 'tf2.0' and f == 2

    def axis(t, dim):
        n = t.shape[dim]
        nxt = torch.index_select(t, dim, torch.clamp(torch.arange(n) + 1, max=n - 1))
        return torch.stack([t, 0.5 * (t + nxt)], dim + 1).reshape(t.shape[:dim] + (2 * n,) + t.shape[dim + 1:])
    return axis(axis(x, 1), 2)


def convlstm_seq(x, kernel, rec_kernel, bias, h0, c0, rounded=False):
    b, t, hh, ww, _ = x.shape
    f = rec_kernel.shape[2]
    h = torch.zeros(b, hh, ww, f, dtype=x.dtype) if h0 is None else h0
    c = torch.zeros(b, hh, ww, f, dtype=x.dtype) if c0 is None else c0
    # hoist the input projection over all T (mathematically identical)
    zx = conv2d_same(x.reshape(b * t, hh, ww, -1), kernel, bias, rounded=rounded).reshape(b, t, hh, ww, 4 * f)
    outs = []
    for ti in range(t):
        z = zx[:, ti] + conv2d_same(h, rec_kernel, None, rounded=rounded)
        i = hard_sigmoid(z[..., :f])
        fg = hard_sigmoid(z[..., f:2 * f])
        g = torch.tanh(z[..., 2 * f:3 * f])
        o = hard_sigmoid(z[..., 3 * f:])
        c = fg * c + i * g
        h = o * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, 1), h, c


def weighted_ce(gt, logits, class_weights):
    """gt [B,T,H,W] float, logits [B,T,H,W,3]; losses.py:13-27."""
    valid = (gt > -1).to(logits.dtype)
    gi = gt.to(torch.int64)
    cw = torch.as_tensor(class_weights, dtype=logits.dtype)
    pix_w = torch.where(gi >= 0, cw[gi.clamp(min=0)], torch.zeros((), dtype=logits.dtype))
    lse = torch.logsumexp(logits, dim=-1)
    picked = torch.gather(logits, -1, gi.clamp(min=0).unsqueeze(-1)).squeeze(-1)
    return ((lse - picked) * pix_w * valid).sum() / (valid.sum() + 0.00001)


class TorchULSTM:
    """Functional model over a name->tensor parameter dict (names as np_oracle.init_params)."""

    def __init__(self, net_params, in_channels, params, dtype=torch.float64, pad_image=False,
                 bn_eps=1e-3, bn_momentum=0.99, bf16_operands=False, resize='tf2.0'):
        """bf16_operands: restate Engine(precision='bf16') -- the convolutions that mode runs on the bf16 MFMA
        (ConvLSTM gate convolutions with 3x3 / 5x5 kernels and 4F > 64 columns; Conv2D layers with >= 64 output channels, or
        exactly 32 for stride-1 3x3 / 5x5 layers) see bf16-rounded operands, everything else stays in `dtype`."""
        self.bf16_operands = bool(bf16_operands)
        self.resize = resize
        self.net_params = net_params
        self.plan = npo.net_plan(net_params, in_channels)
        self.dtype = dtype
        self.pad_image = pad_image
        self.eps = bn_eps
        self.momentum = bn_momentum
        self.P = {k: torch.tensor(v, dtype=dtype) for k, v in params.items()}
        self.trainable = npo.trainable_names(params)
        self.states = None
        self.adam_m = {k: torch.zeros_like(self.P[k]) for k in self.trainable}
        self.adam_v = {k: torch.zeros_like(self.P[k]) for k in self.trainable}
        self.step = 0
        self.capture = None   # diagnostics: dict name -> activation tensors with retain_grad()

    # ---- forward ----------------------------------------------------------------
    def forward(self, x, training=True, update_moving=True):
        P = self.P
        b, t, h, w, cin = x.shape
        S = self.plan['total_stride']
        py, px = npo.model_pads(h, w, S, self.pad_image)
        if any(py) or any(px):
            xp = F.pad(x.reshape(b * t, h, w, cin).permute(0, 3, 1, 2), (px[0], px[1], py[0], py[1]), mode='reflect')
            xp = xp.permute(0, 2, 3, 1).reshape(b, t, h + sum(py), w + sum(px), cin)
        else:
            xp = x
        hp, wp = xp.shape[2], xp.shape[3]
        skips = []
        out_down = xp
        out_skip = xp.reshape(b * t, hp, wp, cin)
        new_states = []

        def cbl(prefix, ci, l, act, with_bn=True, cins=None):
            w_ = P[f'{prefix}.conv.{ci}.kernel']
            n_out, ksz = w_.shape[3], w_.shape[0]
            wide = n_out >= 64 or (n_out == 32 and ksz in (3, 5) and l['stride'] == 1)      # (narrow blocks of the halo kernel)
            rnd = self.bf16_operands and wide      # (thin sources are zero-padded to 4 channels by the engine, not excluded)
            y = conv2d_same(act, w_, P[f'{prefix}.conv.{ci}.bias'], l['stride'], rounded=rnd)
            if not with_bn:
                return y
            bnp = f'{prefix}.bn.{ci}'
            if training:
                z, mean, var = bn_train(y, P[bnp + '.gamma'], P[bnp + '.beta'], self.eps)
                if update_moving:
                    cnt = y.shape[0] * y.shape[1] * y.shape[2]
                    with torch.no_grad():
                        unb = var * (cnt / max(cnt - 1, 1))
                        P[bnp + '.moving_mean'] = self.momentum * P[bnp + '.moving_mean'] + (1 - self.momentum) * mean
                        P[bnp + '.moving_var'] = self.momentum * P[bnp + '.moving_var'] + (1 - self.momentum) * unb
            else:
                z = bn_infer(y, P[bnp + '.gamma'], P[bnp + '.beta'], P[bnp + '.moving_mean'],
                             P[bnp + '.moving_var'], self.eps)
            return F.leaky_relu(z, 0.3)

        for bi, blk in enumerate(self.plan['down']):
            skips.append(out_skip)
            seq = out_down
            blk_states = []
            for li, _ in enumerate(blk['lstm']):
                st = None if self.states is None else self.states[bi][li]
                h0, c0 = (None, None) if st is None else st
                rk = P[f'down.{bi}.lstm.{li}.recurrent_kernel']
                rnd = self.bf16_operands and rk.shape[0] in (3, 5) and rk.shape[3] > 64
                seq, hT, cT = convlstm_seq(seq, P[f'down.{bi}.lstm.{li}.kernel'], rk,
                                           P[f'down.{bi}.lstm.{li}.bias'], h0, c0, rounded=rnd)
                # carried state is a constant for the next window (truncated BPTT)
                blk_states.append((hT.detach(), cT.detach()))
            new_states.append(blk_states)
            act = seq.reshape((b * t,) + tuple(seq.shape[2:]))
            if self.capture is not None and act.requires_grad:
                act.retain_grad()
                self.capture[f'lstm_out.{bi}'] = act
            for ci, l in enumerate(blk['conv']):
                act = cbl(f'down.{bi}', ci, l, act)
            if self.capture is not None and act.requires_grad:
                act.retain_grad()
                self.capture[f'down_out.{bi}'] = act
            out_skip = act
            out_down = act.reshape((b, t) + tuple(act.shape[1:]))
        up_in = out_skip
        for bi, (blk, skip) in enumerate(zip(self.plan['up'], skips[::-1])):
            act = torch.cat([resize_bilinear(up_in, blk['up_factor'], self.resize), skip], dim=-1)
            n = len(blk['conv'])
            for ci, l in enumerate(blk['conv']):
                last = blk['return_logits'] and ci == n - 1
                act = cbl(f'up.{bi}', ci, l, act, with_bn=not last,
                          cins=(up_in.shape[-1], skip.shape[-1]) if ci == 0 else None)
            up_in = act
        logits = up_in.reshape((b, t) + tuple(up_in.shape[1:]))
        logits = logits[:, :, py[0]:py[0] + h, px[0]:px[0] + w, :]
        self.states = new_states
        return logits

    # ---- one optimiser step (train2D.py:87-95) --------------------------------------
    def train_step(self, x, gt, class_weights, lr=1e-5, b1=0.9, b2=0.999, eps=1e-7, apply=True):
        x = torch.as_tensor(x, dtype=self.dtype)
        gt = torch.as_tensor(gt, dtype=self.dtype)
        for k in self.trainable:
            self.P[k] = self.P[k].detach().requires_grad_(True)
        logits = self.forward(x, training=True)
        loss = weighted_ce(gt, logits, class_weights)
        grads = torch.autograd.grad(loss, [self.P[k] for k in self.trainable])
        grads = dict(zip(self.trainable, grads))
        if apply:
            self.step += 1
            alpha = lr * math.sqrt(1 - b2 ** self.step) / (1 - b1 ** self.step)
            with torch.no_grad():
                for k in self.trainable:
                    g = grads[k]
                    self.adam_m[k] = b1 * self.adam_m[k] + (1 - b1) * g
                    self.adam_v[k] = b2 * self.adam_v[k] + (1 - b2) * g * g
                    self.P[k] = (self.P[k] - alpha * self.adam_m[k] / (self.adam_v[k].sqrt() + eps)).detach()
        for k in self.trainable:
            self.P[k] = self.P[k].detach()
        return loss.detach(), logits.detach(), grads

    def reset_states_per_batch(self, keep):
        if self.states is None:
            return
        keep = torch.as_tensor(keep, dtype=self.dtype).reshape(-1, 1, 1, 1)
        self.states = [[(h * keep, c * keep) for (h, c) in blk] for blk in self.states]
