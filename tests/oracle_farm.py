"""Oracle jobs in worker processes (TEST INFRASTRUCTURE: imports oracle/, is imported by tests/ only).

The production-geometry gradient tests (tests/test_fullsize_gpu.py, round 5) need torch autograd through the fp64 oracle at
256 x 256 and 208 x 248 -- a minute or more of host time per case -- plus the same pass in fp32 (the independent fp32
implementation the tolerances are calibrated against) and on bf16-rounded operands.  The GPU box has 256 host cores and torch's
CPU convolutions stop scaling at ~16 threads, so every (case, arithmetic) pair runs as its own process while the GPU works
through the rest of the suite; a test blocks only on the result it needs.

A job is described by a small JSON-able dict; inputs are regenerated from the seed on both sides (same code: `make_inputs`),
weights travel as one .npz written by the test (the engine's exported parameters)."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def labels(rng, B, T, H, W):
    """{-1, 0, 1, 2} maps with structure: blobs of cells with edges, ~10 % of the pixels unlabeled."""
    gt = np.zeros((B, T, H, W), np.float32)
    yy, xx = np.mgrid[0:H, 0:W]
    for b in range(B):
        for t in range(T):
            for _ in range(max(4, H * W // 6000)):
                cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(5, 16)
                d = np.hypot(yy - cy, xx - cx)
                gt[b, t][d < r] = 1
                gt[b, t][(d >= r) & (d < r + 2)] = 2
    gt[rng.random(gt.shape) < 0.1] = -1
    return gt


def make_inputs(spec, state_shapes=None):
    """-> x [B,T,H,W,1] fp32, gt [B,T,H,W], carried states or None.  spec['carried']: random (h, c) per ConvLSTM layer
    ([block][layer] -> (h, c), h in (-1, 1), c ~ N(0, 0.5)) and a keep mask that zeroes the last slot -- the state a training
    window inherits from the one before it (Networks.py:48-50 stateful=True, :77-84 reset_states_per_batch)."""
    B, T, H, W = spec['B'], spec['T'], spec['H'], spec['W']
    rng = np.random.default_rng(spec['seed'])
    x = rng.standard_normal((B, T, H, W, 1)).astype(np.float32)
    gt = labels(rng, B, T, H, W)
    states = keep = None
    if spec.get('carried'):
        assert state_shapes is not None
        states = []
        for blk in state_shapes:
            states.append([(np.tanh(rng.standard_normal(s)).astype(np.float32),
                            (0.5 * rng.standard_normal(s)).astype(np.float32)) for s in blk])
        keep = np.ones(B, np.float32)
        if B > 1:
            keep[-1] = 0.0
    return x, gt, states, keep


def state_shapes(net, B, H, W):
    """[block][layer] -> (B, H_l, W_l, F): ConvLSTM state shapes of the net at this frame size (stride 2 per down block but the last)."""
    out, h, w = [], H, W
    nd = len(net['lstm_kernels'])
    for bi, lst in enumerate(net['lstm_kernels']):
        out.append([(B, h, w, int(f)) for (_, f) in lst])
        if bi < nd - 1:
            h, w = -(-h // 2), -(-w // 2)
    return out


def run_job(spec, params):
    """kind 'grad' (default): one oracle train_step (no Adam) -> loss + every gradient tensor.
    kind 'forward': one forward pass (training or inference mode, no moving-statistics update) -> logits, loss (labels: True),
    carried h / c of every ConvLSTM layer.
    spec: H, W, T, B, seed, dtype ('float64' | 'float32'), bf16_operands, carried, threads, [kind, training, labels]."""
    import torch
    sys.path[:0] = [p for p in (ROOT, os.path.join(ROOT, 'lstm-unet_amd')) if p not in sys.path]
    from oracle import torch_oracle as tho
    import Params
    torch.set_num_threads(max(1, min(int(spec.get('threads', 16)), os.cpu_count() or 1)))
    net = Params.CTCParams.net_kernel_params
    if spec.get('kind') == 'forward':
        B, T, H, W = spec['B'], spec['T'], spec['H'], spec['W']
        rng = np.random.default_rng(spec['seed'])
        x = rng.standard_normal((B, T, H, W, 1)).astype(np.float32)
        gt = labels(rng, B, T, H, W) if spec.get('labels') else None
        dt = getattr(torch, spec['dtype'])
        tm = tho.TorchULSTM(net, 1, params, dtype=dt, bf16_operands=bool(spec.get('bf16_operands')))
        t0 = time.time()
        with torch.no_grad():
            ref = tm.forward(torch.tensor(x, dtype=dt), training=bool(spec['training']), update_moving=False)
            loss = float(tho.weighted_ce(torch.tensor(gt, dtype=dt), ref, [0.15, 0.25, 0.6])) if gt is not None else float('nan')
        out = {'logits': ref.numpy().astype(np.float64), 'loss': np.float64(loss), 'seconds': np.float64(time.time() - t0),
               'max_logit': np.float64(float(ref.abs().max()))}
        for bi, blk in enumerate(tm.states):
            for li, (h, c) in enumerate(blk):
                out['h:%d:%d' % (bi, li)] = h.numpy().astype(np.float64)
                out['c:%d:%d' % (bi, li)] = c.numpy().astype(np.float64)
        return out
    x, gt, states, keep = make_inputs(spec, state_shapes(net, spec['B'], spec['H'], spec['W']))
    dt = getattr(torch, spec['dtype'])
    tm = tho.TorchULSTM(net, 1, params, dtype=dt, bf16_operands=bool(spec.get('bf16_operands')))
    if states is not None:
        tm.states = [[(torch.tensor(h, dtype=dt), torch.tensor(c, dtype=dt)) for (h, c) in blk] for blk in states]
        tm.reset_states_per_batch(keep)
    t0 = time.time()
    loss, logits, grads = tm.train_step(x, gt, [0.15, 0.25, 0.6], apply=False)
    out = {'g:' + k: v.numpy().astype(np.float64) for k, v in grads.items()}
    out['loss'] = np.float64(float(loss))
    out['seconds'] = np.float64(time.time() - t0)
    out['max_logit'] = np.float64(float(logits.abs().max()))
    return out


class Farm(object):
    def __init__(self, params, workdir=None):
        self.dir = workdir or tempfile.mkdtemp(prefix='lu_oracle_farm_')
        self.params_path = os.path.join(self.dir, 'params.npz')
        np.savez(self.params_path, **params)
        self.jobs = {}

    def submit(self, key, spec):
        if key in self.jobs:
            return
        sp, out = os.path.join(self.dir, key + '.json'), os.path.join(self.dir, key + '.npz')
        with open(sp, 'w') as f:
            json.dump(spec, f)
        env = dict(os.environ)
        env['OMP_NUM_THREADS'] = str(int(spec.get('threads', 16)))
        env['HIP_VISIBLE_DEVICES'] = ''      # a worker never touches the GPU
        log = open(os.path.join(self.dir, key + '.log'), 'w')
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__), sp, self.params_path, out], stdout=log, stderr=log, env=env)
        self.jobs[key] = (p, out, log, time.time())

    def result(self, key, timeout=1500):
        p, out, log, t0 = self.jobs[key]
        try:
            rc = p.wait(timeout=max(1.0, timeout - (time.time() - t0)))
        except subprocess.TimeoutExpired:
            p.kill()
            raise RuntimeError('oracle job %s did not finish in %d s' % (key, timeout))
        log.close()
        if rc != 0:
            raise RuntimeError('oracle job %s failed (rc %d):\n%s' % (key, rc, open(log.name).read()[-2000:]))
        d = np.load(out)
        res = {'loss': float(d['loss']), 'seconds': float(d['seconds']), 'max_logit': float(d['max_logit']),
               'grads': {k[2:]: d[k] for k in d.files if k.startswith('g:')}}
        if 'logits' in d.files:
            res['logits'] = d['logits']
            n_blk = 1 + max(int(k.split(':')[1]) for k in d.files if k.startswith('h:'))
            res['states'] = [[(d['h:%d:%d' % (bi, li)], d['c:%d:%d' % (bi, li)])
                              for li in range(sum(1 for k in d.files if k.startswith('h:%d:' % bi)))] for bi in range(n_blk)]
        return res

    def close(self):
        for p, _, log, _ in self.jobs.values():
            if p.poll() is None:
                p.kill()
            if not log.closed:
                log.close()


if __name__ == '__main__':
    spec_path, params_path, out_path = sys.argv[1:4]
    with open(spec_path) as f:
        spec_ = json.load(f)
    with np.load(params_path) as z:
        params_ = {k: z[k] for k in z.files}
    res = run_job(spec_, params_)
    tmp = out_path + '.tmp.npz'
    np.savez(tmp, **res)
    os.replace(tmp, out_path)
