"""Static layer plan + Keras-default initialisers for the ConvLSTM-UNet.

Mirrors the constructor logic of the reference model builder:
  * list-length validation and its ValueError texts        Networks.py:188-193
  * stride 2 for every down block but the last             Networks.py:197
  * up_factor 1 for the first up block, 2 afterwards       Networks.py:202
  * last up block returns logits (no BN/LReLU on its last conv)  Networks.py:148-149,204
  * skips = [image, D0, D1, D2] reversed; concat [up, skip]      Networks.py:145,235-242
"""
import math

import torch


def down_block(convs, lstms, stride, cin):
    """-> (block plan, output channels)"""
    blk = {'lstm': [], 'conv': [], 'stride': stride}
    c = cin
    for (k, f) in lstms:
        blk['lstm'].append({'k': int(k), 'cin': c, 'f': int(f)})
        c = int(f)
    for ci, (k, f) in enumerate(convs):
        blk['conv'].append({'k': int(k), 'cin': c, 'cout': int(f), 'stride': stride if ci == 0 else 1})
        c = int(f)
    return blk, c


def up_block(convs, up_factor, return_logits, c_up, c_skip):
    blk = {'conv': [], 'up_factor': up_factor, 'return_logits': return_logits, 'c_up': c_up, 'c_skip': c_skip}
    cin = c_up + c_skip
    for (k, f) in convs:
        blk['conv'].append({'k': int(k), 'cin': cin, 'cout': int(f), 'stride': 1})
        cin = int(f)
    return blk, cin


def make_plan(net_params, in_channels):
    down, lstm, up = net_params['down_conv_kernels'], net_params['lstm_kernels'], net_params['up_conv_kernels']
    if not len(down) == len(lstm):
        raise ValueError('Number of layers in down path ({}) do not match number of LSTM layers ({})'.format(
            len(down), len(lstm)))
    if not len(down) == len(up):
        raise ValueError('Number of layers in down path ({}) do not match number of layers in up path ({})'.format(
            len(down), len(up)))
    plan = {'down': [], 'up': [], 'total_stride': 1, 'in_channels': in_channels}
    c = in_channels
    skip_ch = [in_channels]
    for bi, (convs, lstms) in enumerate(zip(down, lstm)):
        stride = 2 if bi < len(down) - 1 else 1
        blk, c = down_block(convs, lstms, stride, c)
        plan['total_stride'] *= stride
        plan['down'].append(blk)
        skip_ch.append(c)
    skip_ch = skip_ch[:-1][::-1]
    for bi, convs in enumerate(up):
        blk, c = up_block(convs, 2 if bi > 0 else 1, bi + 1 == len(up), c, skip_ch[bi])
        plan['up'].append(blk)
    plan['last_depth'] = c
    return plan


def param_specs(plan):
    """[(name, shape, kind)] in BACKWARD-COMPLETION order (decoder first, coarse encoder levels next):
    gradient buckets of the flat buffer become ready front-to-back during backward, so the RCCL
    all-reduce of bucket i overlaps the backward of everything behind it."""
    specs = []
    for bi in reversed(range(len(plan['up']))):
        blk = plan['up'][bi]
        n = len(blk['conv'])
        for ci, l in enumerate(blk['conv']):
            specs.append((f'up.{bi}.conv.{ci}.kernel', (l['k'], l['k'], l['cin'], l['cout']), 'glorot'))
            specs.append((f'up.{bi}.conv.{ci}.bias', (l['cout'],), 'zeros'))
            if not (blk['return_logits'] and ci == n - 1):
                specs.append((f'up.{bi}.bn.{ci}.gamma', (l['cout'],), 'ones'))
                specs.append((f'up.{bi}.bn.{ci}.beta', (l['cout'],), 'zeros'))
    for bi in reversed(range(len(plan['down']))):
        blk = plan['down'][bi]
        for ci, l in enumerate(blk['conv']):
            specs.append((f'down.{bi}.conv.{ci}.kernel', (l['k'], l['k'], l['cin'], l['cout']), 'glorot'))
            specs.append((f'down.{bi}.conv.{ci}.bias', (l['cout'],), 'zeros'))
            specs.append((f'down.{bi}.bn.{ci}.gamma', (l['cout'],), 'ones'))
            specs.append((f'down.{bi}.bn.{ci}.beta', (l['cout'],), 'zeros'))
        for li, l in enumerate(blk['lstm']):
            specs.append((f'down.{bi}.lstm.{li}.kernel', (l['k'], l['k'], l['cin'], 4 * l['f']), 'glorot'))
            specs.append((f'down.{bi}.lstm.{li}.recurrent_kernel', (l['k'], l['k'], l['f'], 4 * l['f']), 'orthogonal'))
            specs.append((f'down.{bi}.lstm.{li}.bias', (4 * l['f'],), 'forget_one'))
    return specs


def bn_stat_specs(plan):
    out = []
    for side in ('down', 'up'):
        for bi, blk in enumerate(plan[side]):
            n = len(blk['conv'])
            for ci, l in enumerate(blk['conv']):
                if side == 'up' and blk['return_logits'] and ci == n - 1:
                    continue
                out.append((f'{side}.{bi}.bn.{ci}.moving_mean', (l['cout'],), 'zeros'))
                out.append((f'{side}.{bi}.bn.{ci}.moving_var', (l['cout'],), 'ones'))
    return out


def init_tensor(shape, kind, gen):
    """Keras defaults: glorot_uniform kernels, orthogonal recurrent kernels, zero bias with the
    forget-gate quarter set to one (unit_forget_bias).  CPU float32, deterministic from `gen`."""
    if kind == 'zeros':
        return torch.zeros(shape)
    if kind == 'ones':
        return torch.ones(shape)
    if kind == 'forget_one':
        b = torch.zeros(shape)
        f = shape[0] // 4
        b[f:2 * f] = 1.0
        return b
    if kind == 'glorot':
        rf = shape[0] * shape[1]
        lim = math.sqrt(6.0 / (shape[2] * rf + shape[3] * rf))
        return (torch.rand(shape, generator=gen) * 2 - 1) * lim
    if kind == 'orthogonal':
        rows = shape[0] * shape[1] * shape[2]
        cols = shape[3]
        a = torch.randn((max(rows, cols), min(rows, cols)), generator=gen)
        q, r = torch.linalg.qr(a)
        q = q * torch.sign(torch.diagonal(r))
        if rows < cols:
            q = q.t()
        return q.reshape(shape).contiguous()
    raise ValueError(kind)
