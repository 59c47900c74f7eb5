"""ConvLSTM-UNet execution engine: explicit forward tape + hand-written backward over the HIP ops.

Internal conventions
  * activations are channels-last fp32 [frames, H, W, C]
  * frames are TIME-MAJOR: frame index = t*B + b.  Every per-timestep tensor of the recurrence
    (x_t, h_{t-1}, gates_t, dz_t ...) is then one contiguous slab, h_all[0:T] / h_all[1:T+1] are the
    "previous" / "output" hidden sequences without copies, and the hoisted weight-gradient GEMMs see
    dense [T*B] frame ranges.
  * parameters / gradients / Adam moments live in flat buffers ordered by backward completion
    (plan.param_specs) so DP gradient buckets can be all-reduced while backward continues.

Reference behaviour reproduced (file:line into arbellea/LSTM-UNet):
  ULSTMnet2D.call   Networks.py:208-254     DownBlock2D.call Networks.py:60-75
  UpBlock2D.call    Networks.py:141-153     state API        Networks.py:77-98,279-291
  truncated BPTT: carried (h, c) are constants of the next window (stateful=True, Networks.py:48-50)
"""
import contextlib
import math

import numpy as np
import torch

from . import ops
from . import wbank
from .plan import make_plan, param_specs, bn_stat_specs, init_tensor

LRELU_ALPHA = 0.3   # k.layers.LeakyReLU() default
BN_EPS = 1e-3       # k.layers.BatchNormalization defaults
BN_MOMENTUM = 0.99


def model_pads(h, w, total_stride, pad_image):
    """Networks.py:210-228."""
    mp = total_stride if pad_image else 0
    return ((mp, mp + (total_stride - h % total_stride) % total_stride),
            (mp, mp + (total_stride - w % total_stride) % total_stride))


class _SideLaunch(object):
    """with-block of Engine._wgrad_side: the launches inside go to the engine's side stream, after everything the main
    stream has enqueued so far; the operands stay referenced until the side stream has passed them."""

    def __init__(self, eng, tensors):
        self.eng, self.tensors, self.ctx = eng, tensors, None

    def __enter__(self):
        eng = self.eng
        eng._side_release()
        eng._side_stream.wait_stream(torch.cuda.current_stream(eng.flat_params.device))
        self.ctx = torch.cuda.stream(eng._side_stream)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        ev = torch.cuda.Event()
        ev.record(self.eng._side_stream)
        self.eng._side_keep.append((ev, self.tensors))
        return self.ctx.__exit__(*exc)


class Engine:
    def __init__(self, net_params, pad_image=True, seed=0, dp=None, sync_bn=False, plan_fn=None, precision='fp32',
                 resize='tf2.0'):
        """precision: 'fp32' (default; v_mfma_f32_32x32x2_f32 everywhere -- the parity configuration) or 'bf16'
        (BASELINE config 5: stride-1 3x3 / 5x5 convolutions with more than 64 output channels -- ConvLSTM steps, their
        recurrent / input gradients, the wide encoder / decoder convs -- feed bf16-rounded operands to
        v_mfma_f32_32x32x16_bf16; fp32 master weights, activations, accumulators, statistics, loss and optimiser)
        or 'bf16x3' (fp32 ARITHMETIC on the bf16 MFMA: the ConvLSTM convolutions -- 93 % of the FLOPs -- run the bf16 kernels on
        the exact three-way bf16 split of their fp32 operands, six bf16 products per fp32 product, fp32 accumulation: every
        product to 2^-26, below fp32's rounding unit (_lstm_forward_x3); everything else is the fp32 path, nothing is
        stored rounded)."""
        if precision not in ('fp32', 'bf16', 'bf16x3'):
            raise ValueError("precision must be 'fp32', 'bf16' or 'bf16x3'")
        self.precision = precision
        # bilinear source-coordinate convention of UpBlock2D (Networks.py:143): 'tf2.0' = the legacy v1 op that
        # keras.backend.resize_images calls in the TensorFlow release the reference pins; 'half_pixel' = tf.image.resize v2
        ops._legacy(resize)
        self.resize = resize
        self._packed_version = -1
        self._packed = {}    # (param name, role, c_off, c_sub) -> ops.PackedW; dropped whenever the weights change
        self.bank = wbank.WeightBank()      # flipped / packed images of parameter VIEWS: kept, refreshed by two launches per step
        self._bank_stale = False
        self.prep_batch = True       # A/B (bench.py --ab-no-prep): False = derived weight images rebuilt one launch at a time per step,
                                     # state owned + masked eagerly (the round-2 behaviour)
        self.net_params = net_params
        self._plan_fn = plan_fn if plan_fn is not None else (lambda cin: make_plan(net_params, cin))
        self.pad_image = bool(pad_image)
        self.seed = seed
        self.dp = dp
        self.sync_bn = bool(sync_bn) and dp is not None and bool(getattr(dp, 'collectives', dp.world_size > 1))      # (a forced world of one counts: dp.py)
        self.plan = None
        self.device = None
        self.P = {}       # name -> view into flat_params
        self.G = {}       # name -> view into flat_grads
        self.S = {}       # BN moving statistics
        self._states = None  # [block][layer] -> [h, c] device tensors or None; read through `states` (see _own_states)
        self._keep = None    # state mask of reset_states_per_batch not applied yet: the next training window applies it while
                             # it copies the state into its tape (ops.state_begin), anything else through _own_states()
        self._alias = set()  # (block, layer) whose state tensors are VIEWS of the last training tape (slot T of h_all / c_all)
        self.batch = None
        self.tape = None
        self.segments = []   # [(name, start, end)] gradient buckets in backward-completion order
        self.on_bucket_ready = None  # callback(start, end) fired as each bucket's gradient completes
        self.debug = None            # tests/diag/diag_gpu.py: dict collecting clones of intermediate gradients
        self._h16_seq = None         # bf16 copy of the last ConvLSTM output of the current down block (bf16 tape)
        self._bn_infer = {}          # BN prefix -> (validity token, (scale, shift)) of the inference-mode affine
        self._bn_epoch = 0
        self._state16 = {}           # (block, layer) -> (h state tensor, its bf16 copy) left by the last inference step
        # GPU, bf16 mode: weight gradients go to a side HIP stream (see _wgrad_side: +0.6 % measured; neutral in fp32, where
        # the HBM-bound share of backward is 8x smaller); bench.py's per-kernel timing pass and the host emulator run in line
        self.overlap_wgrad = precision == 'bf16'
        self.narrow_bf16 = True      # A/B (bench.py --ab-old-tail): False keeps the N = 32 decoder layers on the fp32 kernels
        # bf16 mode, training: activations whose every consumer rounds them to bf16 MFMA operands (the next convolution, a
        # ConvLSTM input, the weight gradients) are STORED as bf16 -- BatchNorm'd outputs inside / between blocks, the up-sampled
        # decoder inputs.  No value a kernel computes with changes; the bytes written and re-read halve.  (A/B: bench.py --ab-f32-act)
        self.act_bf16 = True
        self.grad_bf16 = True        # bf16 mode: the gradient a BatchNorm backward hands to its convolution is stored as bf16 when that
                                     # layer's input gradient and weight gradient both round it to bf16 operands (A/B: --ab-f32-act)
        self.s2_fwd_bf16 = True      # stride-2 forward convs behind a ConvLSTM read its bf16 copy (A/B: bench.py --conv-flags 4096 turns it off)
        self._side_stream = None
        self._side_keep = []         # [(event behind the side-stream launches, the tensors they read)]
        self.x3_lean_bytes = None    # precision 'bf16x3': when the window-long split tensors of all layers together would exceed this many
                                     # bytes (None: 30 % of the device's memory) every layer forms its weight / input gradients step by step
                                     # from one step's split tensors instead (config-4: 273 GB of split dz; config-2: 22 GB, hoisted)
        self._x3_lean = False
        self.x3_conv_units = True    # precision 'bf16x3': the wide stride-1 3x3 / 5x5 Conv2D layers on split operands too (False: ConvLSTM layers only)
        self.x3_pad_wgrad = True     # precision 'bf16x3', W % 32 != 0: weight gradients on zero-padded copies of the split tensors (False: fp32 ones)
        self._persistent_states = False  # True (lu_native.graph): inference copies the new state INTO the existing state
                                         # tensors instead of adopting the step's output tensors as the state

    @property
    def states(self):
        """[block][layer] -> [h, c] (or None = zeros), materialised: own tensors, pending mask applied."""
        self._own_states()
        return self._states

    @states.setter
    def states(self, value):
        self._states = value
        self._keep = None
        self._alias = set()

    def _own_states(self):
        """A training window leaves the state as views of its tape and reset_states_per_batch only records its mask -- the
        next training window consumes both in one pass per tensor.  Every other reader (inference, get / set_states, tests,
        the hipGraph path) sees plain tensors: views are cloned (the tape may still be needed by backward, and must not stay
        pinned), then the pending mask is applied in place."""
        if self._states is None:
            return
        for (bi, li) in sorted(self._alias):
            st = self._states[bi][li]
            if st is not None:
                self._states[bi][li] = [st[0].clone(), st[1].clone()]
        self._alias = set()
        if self._keep is not None:
            keep, self._keep = self._keep, None
            for blk in self._states:
                for st in blk:
                    if st is not None:
                        ops.scale_frames(st[0], keep)
                        ops.scale_frames(st[1], keep)

    @property
    def persistent_states(self):
        return self._persistent_states

    @persistent_states.setter
    def persistent_states(self, on):
        self._persistent_states = bool(on)
        self._state16.clear()            # cached bf16 copies are keyed by tensor identity: in-place state updates defeat that

    def invalidate_state_copies(self, bi=None):
        """Every in-place writer of the recurrent state (block-level resets, graph resets) calls this: the bf16 copy of h
        that inference keeps beside the state is keyed by tensor identity and would otherwise outlive the edit."""
        if bi is None:
            self._state16.clear()
        else:
            for key in [k for k in self._state16 if k[0] == bi]:
                del self._state16[key]

    # ------------------------------------------------------------------ build
    def build(self, in_channels, device):
        if self.plan is not None:
            if in_channels != self.plan['in_channels']:
                raise ValueError('model was built for %d input channels' % self.plan['in_channels'])
            return
        self.plan = self._plan_fn(in_channels)
        self.device = device
        specs = param_specs(self.plan)
        offs, total = [], 0
        for name, shape, kind in specs:
            n = int(np.prod(shape))
            offs.append((name, shape, kind, total, n))
            total += (n + 3) // 4 * 4     # keep every tensor 16-byte aligned inside the flat buffer
        self.n_flat = total
        self.flat_params = torch.zeros(total, device=device, dtype=torch.float32)
        self.flat_grads = torch.zeros(total, device=device, dtype=torch.float32)
        gen = torch.Generator().manual_seed(self.seed)
        host = torch.zeros(total, dtype=torch.float32)
        for name, shape, kind, o, n in offs:
            host[o:o + n] = init_tensor(shape, kind, gen).reshape(-1)
            self.P[name] = self.flat_params[o:o + n].view(shape)
            self.G[name] = self.flat_grads[o:o + n].view(shape)
        self.flat_params.copy_(host)
        self._offsets = {name: (o, n) for name, shape, kind, o, n in offs}
        # buckets: one per block, in the order backward finishes them
        self.segments = []
        order = [f'up.{i}' for i in reversed(range(len(self.plan['up'])))] + \
                [f'down.{i}' for i in reversed(range(len(self.plan['down'])))]
        for blk in order:
            members = [(o, n) for name, _, _, o, n in offs if name.startswith(blk + '.')]
            self.segments.append((blk, min(o for o, _ in members), max((o + n + 3) // 4 * 4 for o, n in members)))
        for name, shape, kind in bn_stat_specs(self.plan):
            self.S[name] = init_tensor(shape, kind, None).to(device)
        self.states = [[None for _ in blk['lstm']] for blk in self.plan['down']]
        pending = getattr(self, '_pending_states', None)
        if pending is not None:
            self._pending_states = None
            self.set_states(pending)

    def trainable_names(self):
        return list(self.P.keys())

    def num_trainable(self):
        return sum(int(v.numel()) for v in self.P.values())

    def load_params(self, params):
        """name -> array; used by tests (oracle-initialised weights) and checkpoint restore."""
        with torch.no_grad():       # P[...] are views of a leaf that may require grad (Networks.parameters())
            for k, v in params.items():
                t = torch.as_tensor(np.asarray(v), dtype=torch.float32)
                dst = self.P.get(k, self.S.get(k))
                if dst is None:
                    raise KeyError(k)
                dst.copy_(t.reshape(dst.shape))
        self.weights_changed()

    def export_params(self):
        out = {k: v.detach().cpu().numpy().copy() for k, v in self.P.items()}
        out.update({k: v.detach().cpu().numpy().copy() for k, v in self.S.items()})
        return out

    # ------------------------------------------------------------------ helpers
    def weights_changed(self):
        """Optimiser step / checkpoint load: the cached bf16 weight images and inference BN affines are stale."""
        self._packed.clear()
        self._bank_stale = True
        self._bn_epoch += 1

    def _bf16_conv(self, k, stride, n_out):
        return self.precision == 'bf16' and stride == 1 and k in (3, 5) and n_out > 64 and n_out % 4 == 0

    def _bf16_unit(self, k, stride, n_out):
        """Conv2D layers (and their input gradients, n_out = the gradient's columns) that run on bf16 MFMA operands: every
        layer with >= 64 output columns, and the 32-column stride-1 3x3 / 5x5 layers of the decoder tail (narrow blocks of
        the halo kernel; oracle/torch_oracle.py restates the same rule)."""
        return self.precision == 'bf16' and (n_out >= 64 or (self.narrow_bf16 and n_out == 32 and stride == 1 and k in (3, 5)))

    def _lstm_route(self, k, F, cin, B, H, W):
        """(bf, tape16, x_center, src16) of a ConvLSTM layer's step -- THE routing predicate, shared by _lstm_forward and by
        _down_consumers_bf16 (which decides whether the producer of the layer's input may store it as bf16):
        bf = the gate convolution runs on bf16 MFMA operands; tape16 = fused bf16 step with the bf16 BPTT tape; x_center = the thin
        image enters as an im2col chunk with one tap; src16 = both operands of the step are bf16 TENSORS (x is read as bf16 when
        src16 and not x_center)."""
        bf = self._bf16_conv(k, 1, 4 * F)
        tape16 = bf and ops.fused_step_applies(B, H, W, F, True)
        x_center = tape16 and cin % 4 != 0 and k * k * cin <= 32
        src16 = tape16 and (x_center or cin % 8 == 0)
        return bf, tape16, x_center, src16

    def _down_consumers_bf16(self, plan, bi, ci, B, H, W):
        """True when every consumer of conv unit `ci` of down block `bi` ([.., H, W, cout]) reads it as a bf16 MFMA operand, i.e. when
        storing it as bf16 changes no value anywhere: the block's next convolution; after the last one the next block's first
        ConvLSTM (its fused bf16 step on bf16 tensors -- narrow layers with 4F <= 64 run the fp32 kernels and read fp32), the
        decoder's skip convolution of that level, and -- last block -- the first up block (a bilinear resize reads fp32)."""
        if self.precision != 'bf16':
            return False
        down, up = plan['down'], plan['up']
        blk = down[bi]
        if ci + 1 < len(blk['conv']):
            nxt = blk['conv'][ci + 1]
            return self._bf16_unit(nxt['k'], nxt['stride'], nxt['cout'])
        cout = blk['conv'][ci]['cout']
        if bi + 1 < len(down):
            l = down[bi + 1]['lstm'][0]      # next block's first ConvLSTM: does its step read its input as a bf16 tensor?
            _, _, x_center, src16 = self._lstm_route(l['k'], l['f'], cout, B, H, W)
            if not (src16 and not x_center):
                return False
            c0 = up[len(down) - 2 - bi]['conv'][0]      # skips = [image, D0, D1, D2] reversed: D_bi feeds up block n - 2 - bi
            return self._bf16_unit(c0['k'], c0['stride'], c0['cout'])
        if up[0]['up_factor'] == 2:
            return False
        c0 = up[0]['conv'][0]
        return self._bf16_unit(c0['k'], c0['stride'], c0['cout'])

    def _sync_weight_images(self):
        """Before any use of a derived weight image: drop the per-step cache and refresh the bank if the parameters changed."""
        ver = self.flat_params._version       # in-place torch updates (copy_, torch optimisers) bump it; the raw-pointer
        if ver != self._packed_version:       # Adam kernel does not, hence Adam.apply_gradients -> weights_changed()
            self._packed.clear()
            self._packed_version = ver
            self._bank_stale = True
        if self._bank_stale:
            self._bank_stale = False
            self.bank.refresh()

    def _pack(self, name, role, make, co=0, cs=None, packer=None, view=None):
        """bf16 fragment image of a kernel.  view: the parameter view it is packed from, when it is one (then the bank keeps
        the image across steps); otherwise make() builds a temporary (zero-padded / rearranged) kernel once per step."""
        self._sync_weight_images()
        if view is not None and packer is None and self.prep_batch:
            return self.bank.pack(view)
        key = (name, role, co, cs)
        pw = self._packed.get(key)
        if pw is None:
            pw = self._packed[key] = (packer or ops.pack_bf16)(make())
        return pw

    def _bn_forward(self, prefix, y, training, rec, z16=False):
        gamma, beta = self.P[prefix + '.gamma'], self.P[prefix + '.beta']
        mm, mv = self.S[prefix + '.moving_mean'], self.S[prefix + '.moving_var']
        if training:
            sums = ops.bn_stats(y)
            count = y.numel() // y.shape[-1]
            if self.sync_bn:
                self.dp.all_reduce_(sums)
                count *= self.dp.world_size
            scale, shift, mean, invstd = ops.bn_finalize_train(sums, count, gamma, beta, BN_EPS, BN_MOMENTUM, mm, mv)
            if rec is not None:
                rec.update(scale=scale, shift=shift, mean=mean, invstd=invstd, count=count)
            self._bn_epoch += 1      # the raw-pointer kernel moved mm / mv without bumping their torch versions
        else:
            scale, shift = self._bn_affine_infer(prefix)
        # (W % 32 == 0: the domain of the bf16 kernel-row weight gradient -- on other widths a weight gradient reads this
        # tensor in fp32, unrounded, and storing it as bf16 WOULD change a value)
        return ops.bn_lrelu_apply(y, scale, shift, LRELU_ALPHA, out_bf16=z16 and y.shape[-1] % 8 == 0 and y.shape[2] % 32 == 0)

    def _bn_affine_infer(self, prefix):
        """Inference-mode scale / shift of one BatchNorm: functions of the weights and moving statistics only, computed once
        per change of those, not per frame."""
        mm, mv = self.S[prefix + '.moving_mean'], self.S[prefix + '.moving_var']
        token = (self.flat_params._version, mm._version, mv._version, self._bn_epoch)
        hit = self._bn_infer.get(prefix)
        if hit is None or hit[0] != token:
            hit = self._bn_infer[prefix] = (token, ops.bn_finalize_infer(self.P[prefix + '.gamma'], self.P[prefix + '.beta'],
                                                                         mm, mv, BN_EPS))
        return hit[1]

    def _conv_unit(self, prefix, ci, spec, srcs, with_bn, training, tape, alt16=None, z16=False):
        """srcs: [(x, c_off, c_sub)]; Conv2D -> [BN -> LeakyReLU]  (Networks.py:69-72,146-151).
        alt16: a bf16 copy of the (single) source, if one exists (the ConvLSTM output of a down block): the x operand of the
        layer's weight gradient in bf16 mode.
        z16: the caller knows that every consumer of this unit's activation rounds it to bf16 -- store it as bf16 (training,
        bf16 mode; sources may themselves arrive as bf16 tensors for the same reason)."""
        wname = f'{prefix}.conv.{ci}.kernel'
        w = self.P[wname]
        if self._x3_unit(w.shape[0], spec['stride'], w.shape[3], srcs):
            return self._conv_unit_x3(prefix, ci, spec, srcs, with_bn, training, tape)
        bf = self._bf16_unit(w.shape[0], spec['stride'], w.shape[3])
        any16 = any(x.dtype == torch.bfloat16 for (x, _, _) in srcs)
        if any16 and not (bf and all(x.dtype == torch.bfloat16 or x.shape[3] % 4 or x.shape[3] % 8 == 0 for (x, _, _) in srcs)):
            # a consumer outside the bf16 kernels' domain (fp32 unit, odd channel counts): give it fp32 tensors
            srcs = [(ops.to_f32(x) if x.dtype == torch.bfloat16 else x, co, cs) for (x, co, cs) in srcs]
            any16 = False
        fsrcs = srcs      # what the forward launch reads (the tape keeps `srcs`: the weight gradients see the real tensors)
        if (bf and alt16 is not None and len(srcs) == 1 and spec['stride'] == 1 and w.shape[0] in (3, 5) and tape is not None and
                srcs[0][0].dtype == torch.float32 and alt16.shape == srcs[0][0].shape and alt16.shape[3] % 8 == 0 and
                srcs[0][1] == 0 and srcs[0][2] == alt16.shape[3]):
            # the stride-1 convolution behind a ConvLSTM (last down block): read the bf16 copy of its output -- the bits the
            # kernel would form from the fp32 tensor itself, half the bytes, and a bf16 source is what the tall 3x3 tile needs
            fsrcs = [(alt16, 0, alt16.shape[3])]
            any16 = True
        base = fsrcs
        if bf and (any16 or any(x.shape[3] % 4 for (x, _, _) in srcs)):
            # thin sources (the 1-channel image skip of the last up block): the bf16 kernel reads 16-byte channel groups, so
            # they get zero pad channels (and zero weight rows) for this launch -- 33 MB at config-2, once per step; all sources
            # of a launch share one element type, so next to a bf16 tensor the others become bf16 as well
            fsrcs = []
            for (x, co, cs) in base:
                if any16 and x.dtype != torch.bfloat16:
                    if x.shape[3] % 8:
                        xp = torch.zeros(x.shape[:3] + (-(-x.shape[3] // 8) * 8,), device=x.device, dtype=torch.bfloat16)
                        xp[..., :x.shape[3]] = x      # (round to nearest even, like the kernels' own conversion)
                        x = xp
                    else:
                        x = ops.to_bf16(x)
                elif x.shape[3] % 4:
                    xp = torch.zeros(x.shape[:3] + (-(-x.shape[3] // 4) * 4,), device=x.device, dtype=torch.float32)
                    xp[..., :x.shape[3]] = x
                    x = xp
                fsrcs.append((x, co, cs))
        q = 8 if any16 else 4
        vec = all(x.stride(2) % q == 0 and x.stride(0) % q == 0 and x.data_ptr() % 16 == 0 and x.shape[3] % q == 0
                  for (x, _, _) in fsrcs)      # the bf16 kernel reads 16-byte channel groups
        # bf16: halo kernel where it applies (N = 32 / 64: its narrow blocks), the gather kernel otherwise; N < 32 stays on the
        # fp32 tiles (128-column blocks would idle 3 of 4 column fragments; measured slower)
        if vec and bf:
            def padded(co, cs, cp):
                if cp == cs:
                    return w[:, :, co:co + cs, :]
                wp = torch.zeros((w.shape[0], w.shape[1], cp, w.shape[3]), device=w.device, dtype=torch.float32)
                wp[:, :, :cs] = w[:, :, co:co + cs, :]
                return wp
            pairs = [(x, self._pack(wname, 'fwd', lambda co=co, cs=cs, cp=x.shape[3]: padded(co, cs, cp), co, cs,
                                    view=w[:, :, co:co + cs, :] if x.shape[3] == cs else None))
                     for (x, co, cs) in fsrcs]
        else:
            if any16:
                srcs = [(ops.to_f32(x) if x.dtype == torch.bfloat16 else x, co, cs) for (x, co, cs) in srcs]
            pairs = [(x, w[:, :, co:co + cs, :]) for (x, co, cs) in srcs]
        if (bf and spec['stride'] == 2 and w.shape[0] == 3 and alt16 is not None and len(srcs) == 1 and self.s2_fwd_bf16 and
                alt16.shape[1] % 2 == 0 and alt16.shape[2] % 2 == 0 and alt16.shape[3] % 8 == 0 and tape is not None):
            # the stride-2 layer behind a ConvLSTM, training: read the bf16 copy of its output (same rounded operands as the
            # gather kernel forms from the fp32 tensor, half the bytes, input pixels staged by column parity)
            y = ops.conv2d_s2_fwd_bf16(alt16, self._pack(wname, 'fwd', lambda: w, view=w), self.P[f'{prefix}.conv.{ci}.bias'])
            rec = {'kind': 'conv', 'prefix': prefix, 'ci': ci, 'spec': spec, 'srcs': srcs, 'bn': with_bn, 'alt16': alt16}
            tape.append(rec)
            if not with_bn:
                return y
            rec['y'] = y
            return self._bn_forward(f'{prefix}.bn.{ci}', y, training, rec,
                                    z16=z16 and self.precision == 'bf16' and self.act_bf16)
        if with_bn and tape is None and not training:
            # inference: BatchNorm is a per-channel affine -- it and the LeakyReLU ride on the conv's store / slab reduce
            scale, shift = self._bn_affine_infer(f'{prefix}.bn.{ci}')
            return ops.conv2d(pairs, self.P[f'{prefix}.conv.{ci}.bias'], spec['stride'], post=(scale, shift, LRELU_ALPHA))
        y = ops.conv2d(pairs, self.P[f'{prefix}.conv.{ci}.bias'], spec['stride'])
        rec = None
        if tape is not None:
            rec = {'kind': 'conv', 'prefix': prefix, 'ci': ci, 'spec': spec, 'srcs': srcs, 'bn': with_bn, 'alt16': alt16}
            tape.append(rec)
        if not with_bn:
            return y
        if rec is not None:
            rec['y'] = y
        return self._bn_forward(f'{prefix}.bn.{ci}', y, training, rec,
                                z16=z16 and tape is not None and self.precision == 'bf16' and self.act_bf16)

    # ------------------------------------------------------------------ Conv2D unit, precision 'bf16x3'
    def _x3_unit(self, k, stride, n_out, srcs):
        """precision 'bf16x3': the wide stride-1 3x3 / 5x5 Conv2D layers (the bf16 halo kernel's domain: more than 64 output columns,
        sources of at least 32 channels in 16-byte groups) run on split operands like the ConvLSTM convolutions; stride-2, 1x1 and the
        narrow decoder tail stay on the fp32 kernels (HBM-bound layers: six times the MFMA work would not pay for the split passes)."""
        return (self._x3_conv_route(k, stride, n_out, [cs for (_, _, cs) in srcs]) and
                all(x.dtype == torch.float32 and x.shape[3] == cs for (x, _, cs) in srcs))

    def _x3_conv_route(self, k, stride, n_out, src_channels):
        """The shape half of _x3_unit (what forward()'s memory estimate can know before any tensor exists)."""
        return (self.precision == 'bf16x3' and self.x3_conv_units and stride == 1 and k in (3, 5) and n_out > 64 and n_out % 4 == 0 and
                all(cs >= 32 and cs % 8 == 0 for cs in src_channels))

    def _conv_unit_x3(self, prefix, ci, spec, srcs, with_bn, training, tape):
        """_conv_unit on split operands (Networks.py:69-72,146-151): every source as its split6 image (order A) against its slice of the
        kernel split in order B -- the fp32 convolution to 2^-26 per product; bias, BatchNorm and LeakyReLU as in fp32 mode."""
        wname = f'{prefix}.conv.{ci}.kernel'
        w = self.P[wname]
        bias = self.P[f'{prefix}.conv.{ci}.bias']
        x6s = [ops.split6(x) for (x, _, _) in srcs]
        pairs = [(x6, self._pack(wname, 'x3', lambda co=co, cs=cs: w[:, :, co:co + cs, :], co, cs,
                                 packer=lambda v: ops.pack_split6_bf16(v, None, 1)))
                 for x6, (_, co, cs) in zip(x6s, srcs)]
        if with_bn and tape is None and not training:
            scale, shift = self._bn_affine_infer(f'{prefix}.bn.{ci}')
            return ops.conv2d(pairs, bias, 1, post=(scale, shift, LRELU_ALPHA))
        y = ops.conv2d(pairs, bias, 1)
        rec = None
        if tape is not None:
            rec = {'kind': 'conv', 'x3': True, 'prefix': prefix, 'ci': ci, 'spec': spec, 'srcs': srcs, 'x6s': x6s, 'bn': with_bn,
                   'alt16': None}
            tape.append(rec)
        if not with_bn:
            return y
        if rec is not None:
            rec['y'] = y
        return self._bn_forward(f'{prefix}.bn.{ci}', y, training, rec)

    def _conv_unit_backward_x3(self, rec, dy, need_dx):
        """Input and weight gradients of a split Conv2D unit from dy (fp32, behind the BatchNorm backward): dy6 = split6(dy) in order B
        once; per source the weight gradient with the terms as frames (the bias gradient rides on the first source's launch: its dy
        blocks 0-2 are hi, mid, lo) and the input gradient as the convolution of dy6 with the flipped kernel slice split in order A."""
        prefix, ci = rec['prefix'], rec['ci']
        wname = f'{prefix}.conv.{ci}.kernel'
        w, gw = self.P[wname], self.G[wname]
        k = w.shape[0]
        p = (k - 1) // 2
        frames, H, W, N = dy.shape
        dy6 = ops.split6(dy, order=1)
        dy6_w = self._x3_pad_w(dy6) if self.x3_pad_wgrad else dy6
        dxs = []
        for si, ((x, co, cs), x6, need) in enumerate(zip(rec['srcs'], rec['x6s'], need_dx)):
            x6_w = self._x3_pad_w(x6) if self.x3_pad_wgrad else x6
            dbias = self.G[f'{prefix}.conv.{ci}.bias'] if si == 0 else None
            with self._wgrad_side(x6, dy6, x6_w, dy6_w, x, dy):
                if ops.bf16_row_wgrad_ok(ops.split_piece(x6_w, 'hi'), ops.split_piece(dy6_w, 'hi'), k, 1):
                    self._x3_wgrad(x6_w, dy6_w, gw[:, :, co:co + cs, :], dbias=dbias)
                else:
                    ops.conv2d_wgrad(x, dy, gw[:, :, co:co + cs, :], 1, dbias=dbias)
            if need and self._x3_dgrad_ok(cs):
                wt6 = self._pack(wname, 'x3t', lambda co=co, cs=cs: ops.flip_transpose(w, co, cs), co, cs,
                                 packer=lambda v: ops.pack_split6_bf16(v, None, 0))
                dx = torch.empty((frames, H, W, cs), device=dy.device, dtype=torch.float32)
                ops.conv_raw([(dy6, wt6)], frames, H, W, H, W, k, 1, 1, p, p, cs, None, dx)
                dxs.append(dx)
            elif need:      # (a source of 40 / 48 ... channels: outside the halo kernel's column counts -- the fp32 input gradient)
                dxs.append(ops.conv2d_dgrad(dy, w, (H, W), 1, co, cs, bank=self.bank if self.prep_batch else None))
            else:
                dxs.append(None)
        rec['srcs'] = rec['x6s'] = None
        return dxs

    def _dy16_ok(self, rec, dz, need_dx):
        """bf16 mode: may the gradient w.r.t. this unit's convolution output be STORED as bf16?  Yes when every reader rounds it
        to bf16 MFMA operands anyway: the weight gradient of every source on the bf16 kernel-row variant, every input gradient
        that is needed on the bf16 halo kernel.  (The conv bias in front of the BatchNorm then gets its exact gradient, zero,
        instead of a column sum of rounded values: see _conv_unit_backward.)"""
        spec, k = rec['spec'], self.P[f"{rec['prefix']}.conv.{rec['ci']}.kernel"].shape[0]
        N, W = dz.shape[3], dz.shape[2]
        if not (self.precision == 'bf16' and self.grad_bf16 and spec['stride'] == 1 and k in (3, 5) and N % 8 == 0 and
                W % 32 == 0 and dz.is_contiguous()):
            return False
        probe = torch.empty((1,) + tuple(dz.shape[1:]), device=dz.device, dtype=torch.bfloat16)
        a16 = rec.get('alt16')
        for (x, co, cs), need in zip(rec['srcs'], need_dx):
            xc = a16 if (a16 is not None and ops.bf16_row_wgrad_ok(a16, probe, k, 1)) else x
            if not ops.bf16_row_wgrad_ok(xc, probe, k, 1):
                return False
            if need and not (self._bf16_unit(k, 1, cs) and cs % 4 == 0):
                return False
        return True

    def _conv_unit_backward(self, rec, dz, need_dx):
        """-> list of input gradients (one per source, None where not needed)."""
        prefix, ci, spec = rec['prefix'], rec['ci'], rec['spec']
        w = self.P[f'{prefix}.conv.{ci}.kernel']
        gw = self.G[f'{prefix}.conv.{ci}.kernel']
        if rec['bn']:
            bn = f'{prefix}.bn.{ci}'
            y = rec['y']
            sums = ops.bn_lrelu_bwd_reduce(y, dz, rec['scale'], rec['shift'], rec['mean'], rec['invstd'], LRELU_ALPHA)
            if self.sync_bn:
                Cc = y.shape[-1]
                self.G[bn + '.beta'].copy_(sums[:Cc])        # local sums: the gradient all-reduce adds ranks
                self.G[bn + '.gamma'].copy_(sums[Cc:])
                self.dp.all_reduce_(sums)
                dy = ops.bn_lrelu_bwd_apply(y, dz, rec['scale'], rec['shift'], rec['mean'], rec['invstd'],
                                            LRELU_ALPHA, sums, rec['count'], None, None, out=dz,
                                            out_bf16=self._dy16_ok(rec, dz, need_dx))
            else:
                dy = ops.bn_lrelu_bwd_apply(y, dz, rec['scale'], rec['shift'], rec['mean'], rec['invstd'],
                                            LRELU_ALPHA, sums, rec['count'], self.G[bn + '.gamma'],
                                            self.G[bn + '.beta'], out=dz, out_bf16=self._dy16_ok(rec, dz, need_dx))
            rec['y'] = None
        else:
            dy = dz
        if rec.get('x3'):
            return self._conv_unit_backward_x3(rec, dy, need_dx)
        dxs = []
        for si, ((x, co, cs), need) in enumerate(zip(rec['srcs'], need_dx)):
            # the bias gradient (column sums of dy) rides on the first source's weight-gradient launch
            a16 = rec.get('alt16')
            if a16 is not None and self.precision == 'bf16' and ops.bf16_row_wgrad_ok(a16, dy, gw.shape[0], spec['stride']):
                x = a16
            if x.dtype == torch.bfloat16 and not ops.bf16_row_wgrad_ok(x, dy, gw.shape[0], spec['stride']):
                x = ops.to_f32(x)      # (a layer shape outside the bf16 kernel-row weight gradient: it reads fp32)
            # A conv bias in front of a BatchNorm has gradient sum(dy) = 0 identically (the backward above subtracts the mean);
            # what the column sums of an fp32 dy return is rounding noise far below Adam's epsilon.  Sums of a bf16-STORED dy
            # would be noise 1e4 times larger, which Adam normalises into full +-lr steps -- a random walk of a parameter that
            # cannot change the output.  Such a layer gets its exact gradient instead: zero.
            zero_bias = rec['bn'] and dy.dtype == torch.bfloat16
            if zero_bias and si == 0:
                self.G[f'{prefix}.conv.{ci}.bias'].zero_()
            with self._wgrad_side(x, dy):
                ops.conv2d_wgrad(x, dy, gw[:, :, co:co + cs, :], spec['stride'], bf16=self.precision == 'bf16',
                                 dbias=self.G[f'{prefix}.conv.{ci}.bias'] if (si == 0 and not zero_bias) else None)
            dxs.append(ops.conv2d_dgrad(dy, w, (x.shape[1], x.shape[2]), spec['stride'], co, cs,
                                        bf16=self._bf16_unit(gw.shape[0], spec['stride'], cs),
                                        bank=self.bank if self.prep_batch else None) if need else None)
        rec['srcs'] = rec['alt16'] = None
        return dxs

    # ------------------------------------------------------------------ ConvLSTM layer
    def _lstm_forward(self, bi, li, spec, x_seq, T, B, tape):
        """x_seq [T*B,H,W,C] time-major -> h sequence [T*B,H,W,F]; updates the carried state.
        bf16 mode, fused step: the hidden sequence is ALSO kept as bf16 (h16_all: the recurrent operand of the next step and
        the x operand of the hoisted recurrent weight gradient) and the BPTT tape (saved gates, later dz in place) is bf16 --
        every consumer rounds these tensors to bf16 MFMA operands anyway, so storing them rounded halves their HBM traffic
        and changes no forward value.  fp32 mode is untouched."""
        _, H, W, Cin = x_seq.shape
        F = spec['f']
        k = spec['k']
        dev = x_seq.device
        if self._x3_route(k, F, Cin, B, H, W, training=tape is not None):
            return self._lstm_forward_x3(bi, li, spec, x_seq, T, B, tape)
        pre = f'down.{bi}.lstm.{li}'
        kernel, rec_k, bias = self.P[pre + '.kernel'], self.P[pre + '.recurrent_kernel'], self.P[pre + '.bias']
        bf, tape16, x_center, src16 = self._lstm_route(k, F, Cin, B, H, W)      # (thin image: im2col chunk, one tap)
        if x_seq.dtype == torch.bfloat16 and not (src16 and not x_center):
            # _down_consumers_bf16 asks the same predicate before a producer stores bf16, so this cannot happen for a down-block
            # activation; a bf16 tensor from anywhere else would be a silently rounded fp32 operand -- refuse in training
            if tape is not None:
                raise RuntimeError('ConvLSTM %s reads fp32 but was handed a bf16 activation (routing predicates drifted)' % pre)
            x_seq = ops.to_f32(x_seq)
        x5 = x_seq.view(T, B, H, W, -1)
        x16 = None
        if bf:
            w_in = kernel
            if x_center:
                kernel = self._pack(pre + '.kernel', 'center', lambda w=w_in: w, packer=ops.pack_center_bf16)
                x5 = ops.im2col_bf16(x_seq, k).view(T, B, H, W, 32)
            elif src16:
                kernel = self._pack(pre + '.kernel', 'fwd', lambda w=w_in: w, view=w_in)
                # one pass per window; also the x operand of the hoisted weight gradient (or the producer stored it as bf16)
                x16 = x_seq if x_seq.dtype == torch.bfloat16 else ops.to_bf16(x_seq)
                x5 = x16.view(T, B, H, W, -1)
            else:
                kernel = self._pack(pre + '.kernel', 'fwd', lambda w=w_in: w, view=w_in)
                if Cin % 4 != 0:
                    # the bf16 kernel reads 16-byte channel groups: thin inputs get zero pad channels
                    cpad = -(-Cin // 4) * 4
                    x5 = torch.zeros((T, B, H, W, cpad), device=dev, dtype=torch.float32)
                    x5[..., :Cin] = x_seq.view(T, B, H, W, -1)
            rec_k = self._pack(pre + '.recurrent_kernel', 'fwd', lambda w=rec_k: w, view=rec_k)
        st = self._states[bi][li]      # (forward() materialised it unless this is a training window)
        if st is not None and tuple(st[0].shape) != (B, H, W, F):
            raise ValueError('stateful ConvLSTM: batch/shape changed from %s to %s' % (tuple(st[0].shape), (B, H, W, F)))
        if tape is None:
            # inference: the carried state is read in place and the last step's outputs BECOME the state (no copies in or out)
            h_seq = torch.empty((T, B, H, W, F), device=dev, dtype=torch.float32)
            c_seq = torch.empty((T, B, H, W, F), device=dev, dtype=torch.float32)
            h16_seq = torch.empty((T, B, H, W, F), device=dev, dtype=torch.bfloat16) if tape16 else None
            if st is None:
                h_prev, c_prev = h_seq.new_zeros((B, H, W, F)), h_seq.new_zeros((B, H, W, F))
            else:
                h_prev, c_prev = st
            h16_prev = None
            if src16:
                hit = self._state16.get((bi, li))
                h16_prev = hit[1] if (hit is not None and hit[0] is h_prev) else ops.to_bf16(h_prev)
            for t in range(T):
                ops.convlstm_step(x5[t], h16_prev if src16 else h_prev, c_prev, kernel, rec_k, bias, h_seq[t], c_seq[t], None,
                                  h16_out=h16_seq[t] if tape16 else None, x_center=x_center)
                h_prev, c_prev = h_seq[t], c_seq[t]
                h16_prev = h16_seq[t] if tape16 else None
            if self.persistent_states and st is not None:      # hipGraph replay: the state buffers keep their addresses
                st[0].copy_(h_prev)
                st[1].copy_(c_prev)
                self._state16.pop((bi, li), None)              # (a bf16 copy left by an earlier eager frame is stale now)
            elif self.persistent_states:
                self._states[bi][li] = [h_prev.clone(), c_prev.clone()]
                self._state16.pop((bi, li), None)
            else:
                self._states[bi][li] = [h_prev, c_prev]
                self._state16[(bi, li)] = (h_prev, h16_prev)
            self._h16_seq = None
            return h_seq.view(T * B, H, W, F)
        h_all = torch.empty((T + 1, B, H, W, F), device=dev, dtype=torch.float32)
        c_all = torch.empty((T + 1, B, H, W, F), device=dev, dtype=torch.float32)
        # slot 0 of the tape = the carried state times the pending mask of reset_states_per_batch (+ its bf16 copy): one pass
        h16_all = torch.empty((T + 1, B, H, W, F), device=dev, dtype=torch.bfloat16) if tape16 else None
        ops.state_begin(h_all[0], None if st is None else st[0], self._keep, h16_all[0] if tape16 else None)
        ops.state_begin(c_all[0], None if st is None else st[1], self._keep)
        gates = torch.empty((T, B, H, W, 4 * F), device=dev, dtype=torch.bfloat16 if tape16 else torch.float32)
        for t in range(T):
            ops.convlstm_step(x5[t], h16_all[t] if src16 else h_all[t], c_all[t], kernel, rec_k, bias, h_all[t + 1],
                              c_all[t + 1], gates[t], h16_out=h16_all[t + 1] if tape16 else None, x_center=x_center)
        # the new state IS slot T of this tape (no copy out); whoever needs it as a tensor of its own goes through `states`
        self._states[bi][li] = [h_all[T], c_all[T]]
        self._alias.add((bi, li))
        self._state16.pop((bi, li), None)
        if tape is not None:
            tape.append({'kind': 'lstm', 'bi': bi, 'li': li, 'spec': spec, 'x': x_seq, 'h_all': h_all, 'c_all': c_all,
                         'gates': gates, 'T': T, 'B': B, 'h16_all': h16_all, 'x25': x5 if x_center else None, 'x16': x16})
        self._h16_seq = h16_all[1:].view(T * B, H, W, F) if (h16_all is not None and tape is not None) else None
        return h_all[1:].view(T * B, H, W, F)

    def _lstm_backward(self, rec, dh_seq, need_dx):
        """BPTT inside the window; gradients do not flow into the carried state."""
        if rec.get('x3'):
            return self._lstm_backward_x3(rec, dh_seq, need_dx)
        bi, li, spec, T, B = rec['bi'], rec['li'], rec['spec'], rec['T'], rec['B']
        pre = f'down.{bi}.lstm.{li}'
        kernel, rec_k = self.P[pre + '.kernel'], self.P[pre + '.recurrent_kernel']
        h_all, c_all, gates, x_seq = rec['h_all'], rec['c_all'], rec['gates'], rec['x']
        h16_all, x25, x16 = rec.get('h16_all'), rec.get('x25'), rec.get('x16')
        tape16 = gates.dtype == torch.bfloat16
        _, _, H, W, F = h_all.shape
        k = spec['k']
        dev = h_all.device
        dz = gates      # in place: the saved gates of step t are dead once dz_t is formed (halves the BPTT tape)
        dh5 = dh_seq.view(T, B, H, W, F)
        dc = torch.empty((2, B, H, W, F), device=dev, dtype=torch.float32)
        dh_rec = None
        self._sync_weight_images()
        rec_bf = T > 1 and self._bf16_conv(k, 1, F)
        if not self.prep_batch:
            rec_kt = ops.flip_transpose(rec_k) if T > 1 else None
            rec_kt = ops.pack_bf16(rec_kt) if rec_bf else rec_kt
        else:
            rec_kt = (self.bank.flip_pack(rec_k) if rec_bf else self.bank.flip(rec_k)) if T > 1 else None
        p = (k - 1) // 2
        for t in reversed(range(T)):
            dc_in = dc[(t + 1) & 1] if t < T - 1 else None
            if tape16:
                ops.lstm_gates_bwd_bf16(dz[t], c_all[t], c_all[t + 1], dh5[t], dh_rec, dc_in, dc[t & 1])
            else:
                ops.lstm_gates_bwd(gates[t], c_all[t], c_all[t + 1], dh5[t], dh_rec, dc_in, dz[t], dc[t & 1])
            if t > 0:
                if dh_rec is None:
                    dh_rec = torch.empty((B, H, W, F), device=dev, dtype=torch.float32)
                dz_t = dz[t] if (rec_bf or not tape16) else ops.to_f32(dz[t])       # (fp32 kernels read fp32)
                ops.conv_raw([(dz_t, rec_kt)], B, H, W, H, W, k, 1, 1, p, p, F, None, dh_rec)
        rec['gates'] = None
        dz_seq = dz.view(T * B, H, W, 4 * F)
        dz32 = [None]

        def dz_f32():      # fallback for consumers without a bf16-operand kernel (small / odd layer shapes)
            if not tape16:
                return dz_seq
            if dz32[0] is None:
                dz32[0] = ops.to_f32(dz_seq)
            return dz32[0]

        # hoisted over all T: one big reduction per weight, on the side stream (_wgrad_side)
        bf = self.precision == 'bf16'
        hp16 = h16_all[:T].view(T * B, H, W, F) if tape16 else None
        rec_16 = tape16 and ops.bf16_row_wgrad_ok(hp16, dz_seq, k, 1)
        ker_25 = x25 is not None and ops.bf16_row_wgrad_ok(x25.view(T * B, H, W, 32), dz_seq, 1, 1)
        ker_16 = (not ker_25) and tape16 and ops.bf16_row_wgrad_ok(x16 if x16 is not None else x_seq, dz_seq, k, 1)
        dx_bf = self._bf16_conv(k, 1, kernel.shape[2])
        if tape16 and (not rec_16 or not (ker_25 or ker_16) or (need_dx and not dx_bf)):
            dz_f32()      # a consumer without a bf16-operand kernel: the fp32 copy is made on the main stream, before the fork
        gk = self.G[pre + '.kernel']
        with self._wgrad_side(dz_seq, dz32[0], hp16, h_all, x25, x16, x_seq):
            if rec_16:
                ops.conv2d_wgrad(hp16, dz_seq, self.G[pre + '.recurrent_kernel'], 1, bf16=True,
                                 dbias=self.G[pre + '.bias'])       # + the bias gradient = column sums of dz, on the side
            else:
                ops.conv2d_wgrad(h_all[:T].view(T * B, H, W, F), dz_f32(), self.G[pre + '.recurrent_kernel'], 1, bf16=bf,
                                 dbias=self.G[pre + '.bias'])
            if ker_25:
                # thin image: the weight gradient of the im2col chunk is a 1x1 problem over its 32 (tap, c) rows
                tmp = torch.empty((1, 1, 32, 4 * F), device=dev, dtype=torch.float32)
                ops.conv2d_wgrad(x25.view(T * B, H, W, 32), dz_seq, tmp, 1, bf16=True)
                rows = gk.shape[0] * gk.shape[1] * gk.shape[2]
                gk.view(rows, 4 * F).copy_(tmp.view(32, 4 * F)[:rows])
            elif ker_16:
                ops.conv2d_wgrad(x16 if x16 is not None else x_seq, dz_seq, gk, 1, bf16=True)
            else:
                ops.conv2d_wgrad(ops.to_f32(x_seq) if x_seq.dtype == torch.bfloat16 else x_seq, dz_f32(), gk, 1, bf16=bf)
        dx = None
        if need_dx:
            dx = ops.conv2d_dgrad(dz_seq if (dx_bf or not tape16) else dz_f32(), kernel, (H, W), 1, bf16=dx_bf,
                                  bank=self.bank if self.prep_batch else None)
        rec['h_all'] = rec['c_all'] = rec['x'] = rec['h16_all'] = rec['x25'] = rec['x16'] = None
        return dx

    # ------------------------------------------------------------------ ConvLSTM layer, precision 'bf16x3'
    def _x3_route(self, k, F, cin, B, H, W, training=True):
        """precision 'bf16x3': does this ConvLSTM layer run on the split operands?  The domain of the fused bf16 step (F % 32 == 0,
        F >= 64; any frame size); other layers take the fp32 kernels.  The bf16 kernel-row weight gradient wants W % 32 == 0:
        on other widths (config-4's 496 / 248 / 124-pixel levels) it runs on zero-padded copies of the split tensors (_x3_pad_w;
        x3_pad_wgrad = False: the fp32 weight gradients on the fp32 tape instead)."""
        return self.precision == 'bf16x3' and k in (3, 5) and F % 32 == 0 and F >= 64 and ops.fused_step_applies(B, H, W, F, True)

    def _x3_weight(self, name, role, make, cp=None, order=1):
        """bf16 fragment image of the three-way split of a kernel (block order B; A for the kernels that meet dz, which is
        split in order B), once per weight change."""
        return self._pack(name, role, make, packer=lambda w: ops.pack_split6_bf16(w, cp, order))

    def _lstm_forward_x3(self, bi, li, spec, x_seq, T, B, tape):
        """The ConvLSTM layer of precision 'bf16x3' (reference Networks.py:48-50,62-63; the fp32 form: _lstm_forward).  x_t and
        h_{t-1} enter the fused bf16 step as split6 tensors (six channel blocks of the exact three-way bf16 split), the
        kernels in the matching block order, so the step's MFMAs sum x w and h r to 2^-26 per product in fp32 accumulators;
        gates, c and h come out of the epilogue in fp32 exactly as in fp32 mode and NOTHING is stored rounded: the tape is
        the fp32 tape plus the split copy of the hidden sequence (the recurrent operand of the next step and the x operand
        of the hoisted recurrent weight gradient)."""
        _, H, W, Cin = x_seq.shape
        F, k = spec['f'], spec['k']
        dev = x_seq.device
        pre = f'down.{bi}.lstm.{li}'
        kernel, rec_k, bias = self.P[pre + '.kernel'], self.P[pre + '.recurrent_kernel'], self.P[pre + '.bias']
        cp = -(-Cin // 4) * 4      # channels per block: the bf16 kernels read 16-byte groups (6 cp % 8 == 0)
        k6 = self._x3_weight(pre + '.kernel', 'x3', lambda: kernel, cp)
        r6 = self._x3_weight(pre + '.recurrent_kernel', 'x3', lambda: rec_k)
        st = self._states[bi][li]
        if st is not None and tuple(st[0].shape) != (B, H, W, F):
            raise ValueError('stateful ConvLSTM: batch/shape changed from %s to %s' % (tuple(st[0].shape), (B, H, W, F)))
        if tape is None:
            x6 = ops.split6(x_seq, cp).view(T, B, H, W, 6 * cp)
            h_seq = torch.empty((T, B, H, W, F), device=dev, dtype=torch.float32)
            c_seq = torch.empty((T, B, H, W, F), device=dev, dtype=torch.float32)
            if st is None:
                h_prev, c_prev = h_seq.new_zeros((B, H, W, F)), h_seq.new_zeros((B, H, W, F))
            else:
                h_prev, c_prev = st
            hit = self._state16.get((bi, li))
            h6 = hit[1] if (hit is not None and hit[0] is h_prev) else ops.split6(h_prev)
            for t in range(T):
                h6_next = torch.empty((B, H, W, 6 * F), device=dev, dtype=torch.bfloat16)      # (written by the gate epilogue: LU_CONV_F_H16_SPLIT)
                ops.convlstm_step(x6[t], h6, c_prev, k6, r6, bias, h_seq[t], c_seq[t], None, h6_out=h6_next)
                h_prev, c_prev = h_seq[t], c_seq[t]
                h6 = h6_next
            if self.persistent_states and st is not None:
                st[0].copy_(h_prev)
                st[1].copy_(c_prev)
                self._state16.pop((bi, li), None)
            elif self.persistent_states:
                self._states[bi][li] = [h_prev.clone(), c_prev.clone()]
                self._state16.pop((bi, li), None)
            else:
                self._states[bi][li] = [h_prev, c_prev]
                self._state16[(bi, li)] = (h_prev, h6)
            self._h16_seq = None
            return h_seq.view(T * B, H, W, F)
        h_all = torch.empty((T + 1, B, H, W, F), device=dev, dtype=torch.float32)
        c_all = torch.empty((T + 1, B, H, W, F), device=dev, dtype=torch.float32)
        # lean: the split images live for one step only (two rolling slots of h6 here; backward re-splits h_t, x_t and forms dz6_t
        # per step) -- for layers whose window-long split dz would not fit beside the fp32 tape
        lean = self._x3_lean
        if not lean:
            x6 = ops.split6(x_seq, cp).view(T, B, H, W, 6 * cp)      # one pass per window; also the x operand of the kernel gradient
        else:         # (the window-long split of x is not kept either)
            x6 = None
            x6_t = torch.empty((B, H, W, 6 * cp), device=dev, dtype=torch.bfloat16)
            x5f = x_seq.view(T, B, H, W, Cin)
        h6_all = torch.empty((2 if lean else T + 1, B, H, W, 6 * F), device=dev, dtype=torch.bfloat16)
        ops.state_begin(h_all[0], None if st is None else st[0], self._keep)
        ops.state_begin(c_all[0], None if st is None else st[1], self._keep)
        ops.split6(h_all[0], out=h6_all[0])
        gates = torch.empty((T, B, H, W, 4 * F), device=dev, dtype=torch.float32)
        for t in range(T):
            # (the gate epilogue writes the split image of h_t next to h_t: LU_CONV_F_H16_SPLIT)
            src, dst = (h6_all[t & 1], h6_all[(t + 1) & 1]) if lean else (h6_all[t], h6_all[t + 1])
            ops.convlstm_step(ops.split6(x5f[t], cp, out=x6_t) if lean else x6[t], src, c_all[t], k6, r6, bias, h_all[t + 1],
                              c_all[t + 1], gates[t], h6_out=dst if t + 1 < T else None)
        self._states[bi][li] = [h_all[T], c_all[T]]
        self._alias.add((bi, li))
        self._state16.pop((bi, li), None)
        tape.append({'kind': 'lstm', 'x3': True, 'bi': bi, 'li': li, 'spec': spec, 'x': x_seq, 'x6': None if lean else x6, 'h_all': h_all,
                     'c_all': c_all, 'gates': gates, 'T': T, 'B': B, 'h6_all': None if lean else h6_all, 'cp': cp})
        self._h16_seq = None
        return h_all[1:].view(T * B, H, W, F)

    @staticmethod
    def _x3_dgrad_ok(n_cols):
        """An input gradient is a convolution with `n_cols` = the layer's INPUT channels as output columns: on split operands it needs
        the bf16 halo kernel's column counts (more than 64, or its narrow blocks of 32 / 64); other layers (a 16-channel input) take the
        fp32 input gradient on the fp32 dz."""
        return n_cols % 4 == 0 and (n_cols > 64 or n_cols in (32, 64))

    @staticmethod
    def _x3_pad_w(t6, out=None):
        """split6 tensor [frames,H,W,C6] -> the same rows zero-padded to a multiple of 32 pixels: the domain of the bf16 kernel-row weight
        gradient.  Zero columns of dz add nothing to a weight gradient and zero columns of x are what SAME padding reads there anyway,
        so the gradient of the padded problem IS the gradient (config-4's 496 / 248 / 124-pixel levels).  out: a buffer whose pad
        columns are already zero (the step-by-step route reuses it)."""
        fr, H, W, C6 = t6.shape
        Wp = -(-W // 32) * 32
        if Wp == W:
            return t6
        if out is None:
            out = torch.empty((fr, H, Wp, C6), device=t6.device, dtype=t6.dtype)
        if C6 % 8 == 0 and t6.is_contiguous() and out.is_contiguous():
            # the library's window copy (zero fill outside) on the rows seen as fp32: 16-byte transactions, one launch -- as a torch
            # strided copy of 2-byte elements this was 74 ms of a config-4 step
            ops.window_copy(t6.view(torch.float32), (H, Wp), (0, 0), 0, out=out.view(torch.float32))
        else:
            out[:, :, W:].zero_()
            out[:, :, :W].copy_(t6)
        return out

    def _x3_wgrad(self, x6, dy6, dw, dbias=None, beta0=0.0):
        """dw = x (*) dy on the split operands (x6 in order A, dy6 in order B: block t against block t is term t of
        ops.SPLIT_TERMS).  Two launches of the bf16 kernel-row weight gradient with the terms as extra frames
        (lu_wgrad_desc.terms): the three small products -- whose dy blocks are hi, mid, lo, i.e. dy itself, so the bias gradient
        rides on this launch as the column sums of the three pieces -- then the three large ones on top (beta = 1)."""
        if ops.x3_pieces_ok(x6, dy6, dw.shape[0]):      # round 6: each piece staged once, the six products from registers -- one launch
            ops.conv2d_wgrad(x6, dy6, dw, 1, beta=beta0, bf16=True, dbias=dbias, dbias_beta=beta0, terms=(0, 6), pieces=True)
            return
        ops.conv2d_wgrad(x6, dy6, dw, 1, beta=beta0, bf16=True, dbias=dbias, dbias_beta=beta0, terms=(0, 3))
        ops.conv2d_wgrad(x6, dy6, dw, 1, beta=1.0, bf16=True, terms=(3, 3))

    def _lstm_backward_x3(self, rec, dh_seq, need_dx):
        """BPTT of a 'bf16x3' ConvLSTM layer: the fp32 gate backward, then every convolution of the fp32 path -- the recurrent
        gradient per step, the hoisted weight gradients, the input gradient -- on split6 operands."""
        bi, li, spec, T, B = rec['bi'], rec['li'], rec['spec'], rec['T'], rec['B']
        pre = f'down.{bi}.lstm.{li}'
        kernel, rec_k = self.P[pre + '.kernel'], self.P[pre + '.recurrent_kernel']
        h_all, c_all, gates, x_seq, x6, h6_all = rec['h_all'], rec['c_all'], rec['gates'], rec['x'], rec['x6'], rec['h6_all']
        _, _, H, W, F = h_all.shape
        k = spec['k']
        Cin = x_seq.shape[3]
        dev = h_all.device
        p = (k - 1) // 2
        if h6_all is None:
            return self._lstm_backward_x3_lean(rec, dh_seq, need_dx)
        dz = gates      # in place, as in fp32 mode
        dz6 = torch.empty((T, B, H, W, 24 * F), device=dev, dtype=torch.bfloat16)
        dh5 = dh_seq.view(T, B, H, W, F)
        dc = torch.empty((2, B, H, W, F), device=dev, dtype=torch.float32)
        dh_rec = None
        self._sync_weight_images()
        rt6 = self._x3_weight(pre + '.recurrent_kernel', 'x3t', lambda: ops.flip_transpose(rec_k), order=0) if T > 1 else None
        for t in reversed(range(T)):
            dc_in = dc[(t + 1) & 1] if t < T - 1 else None
            # dz6 in order B: block t of dz6 meets block t of the order-A x6 / h6 in the weight gradients
            ops.lstm_gates_bwd_split(dz[t], c_all[t], c_all[t + 1], dh5[t], dh_rec, dc_in, dz6[t], dc[t & 1])
            if t > 0:
                if dh_rec is None:
                    dh_rec = torch.empty((B, H, W, F), device=dev, dtype=torch.float32)
                ops.conv_raw([(dz6[t], rt6)], B, H, W, H, W, k, 1, 1, p, p, F, None, dh_rec)
        rec['gates'] = None
        dz_seq = dz.view(T * B, H, W, 4 * F)
        dz6_seq = dz6.view(T * B, H, W, 24 * F)
        hp6 = h6_all[:T].view(T * B, H, W, 6 * F)
        x6s = x6.view(T * B, H, W, -1)
        # W % 32 != 0: the weight gradients see zero-padded copies of the split tensors (_x3_pad_w)
        dz6_w, hp6_w = (self._x3_pad_w(dz6_seq), self._x3_pad_w(hp6)) if self.x3_pad_wgrad else (dz6_seq, hp6)
        x6s_w = self._x3_pad_w(x6s) if (self.x3_pad_wgrad and 6 * Cin == x6s.shape[3] and Cin >= 32) else x6s
        with self._wgrad_side(dz_seq, dz6_seq, hp6, x6s, x_seq, h_all, dz6_w, hp6_w, x6s_w):
            if ops.bf16_row_wgrad_ok(ops.split_piece(hp6_w, 'hi'), ops.split_piece(dz6_w, 'hi'), k, 1):
                # (+ the bias gradient: the column sums of the hi, mid and lo blocks of dz6 add up to those of dz)
                self._x3_wgrad(hp6_w, dz6_w, self.G[pre + '.recurrent_kernel'], dbias=self.G[pre + '.bias'])
            else:      # outside the bf16 kernel-row weight gradient (x3_pad_wgrad = False): the fp32 one on the fp32 tape
                ops.conv2d_wgrad(h_all[:T].view(T * B, H, W, F), dz_seq, self.G[pre + '.recurrent_kernel'], 1, dbias=self.G[pre + '.bias'])
            if 6 * Cin == x6s.shape[3] and ops.bf16_row_wgrad_ok(ops.split_piece(x6s_w, 'hi'), ops.split_piece(dz6_w, 'hi'), k, 1):
                self._x3_wgrad(x6s_w, dz6_w, self.G[pre + '.kernel'])
            else:      # thin image (or an odd channel count / width): the fp32 weight gradient, as in fp32 mode
                ops.conv2d_wgrad(x_seq, dz_seq, self.G[pre + '.kernel'], 1)
        dx = None
        if need_dx and self._x3_dgrad_ok(Cin):
            kt6 = self._x3_weight(pre + '.kernel', 'x3t', lambda: ops.flip_transpose(kernel), order=0)
            dx = torch.empty((T * B, H, W, Cin), device=dev, dtype=torch.float32)
            ops.conv_raw([(dz6_seq, kt6)], T * B, H, W, H, W, k, 1, 1, p, p, Cin, None, dx)
        elif need_dx:
            dx = ops.conv2d_dgrad(dz_seq, kernel, (H, W), 1, bank=self.bank if self.prep_batch else None)
        rec['h_all'] = rec['c_all'] = rec['x'] = rec['x6'] = rec['h6_all'] = None
        return dx

    def _lstm_backward_x3_lean(self, rec, dh_seq, need_dx):
        """_lstm_backward_x3 for layers whose window-long split tensors would not fit (config-4): per step t one split dz6_t, the
        split images of h_{t-1} and x_t re-formed from the fp32 tape, and the step's share of every gradient -- recurrent gradient,
        both weight gradients (accumulated over t with beta = 1: the fp32 sum the hoisted launch forms inside its slabs), input
        gradient -- before the buffers are reused.  Same products, same accumulators; the sum over t happens in dw."""
        bi, li, spec, T, B = rec['bi'], rec['li'], rec['spec'], rec['T'], rec['B']
        pre = f'down.{bi}.lstm.{li}'
        kernel, rec_k = self.P[pre + '.kernel'], self.P[pre + '.recurrent_kernel']
        h_all, c_all, gates, x_seq, cp = rec['h_all'], rec['c_all'], rec['gates'], rec['x'], rec['cp']
        _, _, H, W, F = h_all.shape
        k = spec['k']
        Cin = x_seq.shape[3]
        dev = h_all.device
        p = (k - 1) // 2
        dz = gates
        x5 = x_seq.view(T, B, H, W, Cin)
        dh5 = dh_seq.view(T, B, H, W, F)
        dz6 = torch.empty((B, H, W, 24 * F), device=dev, dtype=torch.bfloat16)
        h6 = torch.empty((B, H, W, 6 * F), device=dev, dtype=torch.bfloat16)
        x6 = torch.empty((B, H, W, 6 * cp), device=dev, dtype=torch.bfloat16)
        dc = torch.empty((2, B, H, W, F), device=dev, dtype=torch.float32)
        dh_rec = None
        self._sync_weight_images()
        rt6 = self._x3_weight(pre + '.recurrent_kernel', 'x3t', lambda: ops.flip_transpose(rec_k), order=0) if T > 1 else None
        dx_split = need_dx and self._x3_dgrad_ok(Cin)
        kt6 = self._x3_weight(pre + '.kernel', 'x3t', lambda: ops.flip_transpose(kernel), order=0) if dx_split else None
        dx = torch.empty((T, B, H, W, Cin), device=dev, dtype=torch.float32) if dx_split else None
        # W % 32 != 0: the weight gradients see zero-padded copies (_x3_pad_w; the pad columns of the reused buffers stay zero)
        Wp = -(-W // 32) * 32 if self.x3_pad_wgrad else W
        pad = Wp != W
        dz6_w = torch.zeros((B, H, Wp, 24 * F), device=dev, dtype=torch.bfloat16) if pad else dz6
        h6_w = torch.zeros((B, H, Wp, 6 * F), device=dev, dtype=torch.bfloat16) if pad else h6
        x6_w = torch.zeros((B, H, Wp, 6 * cp), device=dev, dtype=torch.bfloat16) if (pad and Cin == cp and Cin >= 32) else x6
        h_split = ops.bf16_row_wgrad_ok(ops.split_piece(h6_w, 'hi'), ops.split_piece(dz6_w, 'hi'), k, 1)
        x_split = Cin == cp and ops.bf16_row_wgrad_ok(ops.split_piece(x6_w, 'hi'), ops.split_piece(dz6_w, 'hi'), k, 1)
        first = True
        for t in reversed(range(T)):
            dc_in = dc[(t + 1) & 1] if t < T - 1 else None
            ops.lstm_gates_bwd_split(dz[t], c_all[t], c_all[t + 1], dh5[t], dh_rec, dc_in, dz6, dc[t & 1])
            if t > 0:
                if dh_rec is None:
                    dh_rec = torch.empty((B, H, W, F), device=dev, dtype=torch.float32)
                ops.conv_raw([(dz6, rt6)], B, H, W, H, W, k, 1, 1, p, p, F, None, dh_rec)
            beta0 = 0.0 if first else 1.0
            if pad and (h_split or x_split):
                self._x3_pad_w(dz6, out=dz6_w)
            if h_split:
                ops.split6(h_all[t], out=h6)
                if pad:
                    self._x3_pad_w(h6, out=h6_w)
                self._x3_wgrad(h6_w, dz6_w, self.G[pre + '.recurrent_kernel'], dbias=self.G[pre + '.bias'], beta0=beta0)
            if x_split:
                ops.split6(x5[t], cp, out=x6)
                if pad:
                    self._x3_pad_w(x6, out=x6_w)
                self._x3_wgrad(x6_w, dz6_w, self.G[pre + '.kernel'], beta0=beta0)
            if dx_split:
                ops.conv_raw([(dz6, kt6)], B, H, W, H, W, k, 1, 1, p, p, Cin, None, dx[t])
            first = False
        # weight gradients outside the bf16 kernel-row variant's domain (thin image, W % 32 != 0): the fp32 ones, hoisted over the
        # window on the fp32 tape exactly as in fp32 mode
        dz_seq = dz.view(T * B, H, W, 4 * F)
        if not h_split:
            ops.conv2d_wgrad(h_all[:T].view(T * B, H, W, F), dz_seq, self.G[pre + '.recurrent_kernel'], 1, dbias=self.G[pre + '.bias'])
        if not x_split:
            ops.conv2d_wgrad(x_seq, dz_seq, self.G[pre + '.kernel'], 1)
        if need_dx and not dx_split:      # (an input of few channels: the fp32 input gradient, hoisted as in fp32 mode)
            dx = ops.conv2d_dgrad(dz_seq, kernel, (H, W), 1, bank=self.bank if self.prep_batch else None)
        rec['gates'] = rec['h_all'] = rec['c_all'] = rec['x'] = None
        return None if dx is None else dx.view(T * B, H, W, Cin)

    # ------------------------------------------------------------------ forward / backward
    def forward(self, x_tb, T, B, training):
        """x_tb [T*B,H,W,C] time-major -> logits [T*B,H,W,last_depth]; records a tape when training."""
        self.build(x_tb.shape[-1], x_tb.device)
        if self.batch is None:
            self.batch = B
        elif self.batch != B:
            raise ValueError('stateful model: batch size is fixed at first call (%d), got %d' % (self.batch, B))
        plan = self.plan
        tape = [] if training else None
        if not training:
            self._own_states()      # inference reads the carried state in place
        _, H, W, _ = x_tb.shape
        py, px = model_pads(H, W, plan['total_stride'], self.pad_image)
        if any(py) or any(px):
            x_in = ops.window_copy(x_tb, (H + sum(py), W + sum(px)), (py[0], px[0]), 1)
        else:
            x_in = x_tb
        if self.precision == 'bf16x3' and training:
            # window-long split tensors of every ConvLSTM layer (dz6 + the split hidden sequence), were they all kept
            # (bytes per pixel and frame: dz6 48 F + h6 12 F, the layer input's x6 12 cin, the split copy every Conv2D unit on split
            # operands keeps on the tape 12 cin; levels whose width is not a multiple of 32 also hold zero-padded full-window copies
            # of all of them for the kernel-row weight gradient, _x3_pad_w: about twice the bytes)
            extra, hh, ww = 0.0, x_in.shape[1], x_in.shape[2]
            n_px = lambda h_, w_: float(T * B * h_ * w_) * (2.0 if w_ % 32 else 1.0)      # noqa: E731
            for blk in plan['down']:
                extra += sum(n_px(hh, ww) * (60.0 * l['f'] + 12.0 * (-(-l['cin'] // 8) * 8)) for l in blk['lstm'])
                for l in blk['conv']:
                    ho, wo = -(-hh // l['stride']), -(-ww // l['stride'])
                    if self._x3_conv_route(l['k'], l['stride'], l['cout'], [l['cin']]):
                        extra += n_px(ho, wo) * 12.0 * l['cin']
                    hh, ww = ho, wo
            for blk in plan['up']:
                hh, ww = hh * blk['up_factor'], ww * blk['up_factor']
                for ci, l in enumerate(blk['conv']):
                    if self._x3_conv_route(l['k'], 1, l['cout'], [blk['c_up'], blk['c_skip']] if ci == 0 else [l['cin']]):
                        extra += n_px(hh, ww) * 12.0 * l['cin']
            limit = self.x3_lean_bytes
            if limit is None:
                limit = 0.3 * torch.cuda.get_device_properties(x_tb.device).total_memory if x_tb.device.type == 'cuda' else 1e18
            self._x3_lean = extra > limit
        skips = []
        act = x_in
        for bi, blk in enumerate(plan['down']):
            skips.append(act)
            seq = act
            self._h16_seq = None
            for li, l in enumerate(blk['lstm']):
                seq = self._lstm_forward(bi, li, l, seq, T, B, tape)
            n_dc = len(blk['conv'])
            for ci, l in enumerate(blk['conv']):
                # bf16 storage only if EVERY consumer of this activation rounds it to a bf16 MFMA operand anyway (_down_consumers_bf16)
                z16 = self._down_consumers_bf16(plan, bi, ci, B, -(-seq.shape[1] // l['stride']), -(-seq.shape[2] // l['stride']))
                seq = self._conv_unit(f'down.{bi}', ci, l, [(seq, 0, l['cin'])], True, training, tape,
                                      alt16=self._h16_seq if ci == 0 else None, z16=z16)
            self._h16_seq = None
            act = seq
        up_in = act
        for bi, (blk, skip) in enumerate(zip(plan['up'], skips[::-1])):
            if blk['up_factor'] == 2:
                c0 = blk['conv'][0]
                u16 = (tape is not None and self.precision == 'bf16' and self.act_bf16 and blk['c_up'] % 8 == 0 and
                       (2 * up_in.shape[2]) % 32 == 0 and self._bf16_unit(c0['k'], c0['stride'], c0['cout']))      # its consumer rounds it to bf16 anyway
                if up_in.dtype == torch.bfloat16:
                    up_in = ops.to_f32(up_in)
                u = ops.upsample2x(up_in, self.resize, out_bf16=u16)
                if tape is not None:
                    tape.append({'kind': 'up', 'in_hw': (up_in.shape[1], up_in.shape[2])})
            else:
                u = up_in
            n = len(blk['conv'])
            a = None
            for ci, l in enumerate(blk['conv']):
                last = blk['return_logits'] and ci == n - 1
                srcs = [(u, 0, blk['c_up']), (skip, blk['c_up'], blk['c_skip'])] if ci == 0 else [(a, 0, l['cin'])]
                nxt = blk['conv'][ci + 1] if ci + 1 < n else None      # the consumer inside the block (the block's output is resized)
                a = self._conv_unit(f'up.{bi}', ci, l, srcs, not last, training, tape,
                                    z16=nxt is not None and self._bf16_unit(nxt['k'], nxt['stride'], nxt['cout']))
            up_in = a
        logits = up_in
        if any(py) or any(px):
            logits = ops.window_copy(logits, (H, W), (-py[0], -px[0]), 0)
        if training:
            self.tape = {'ops': tape, 'pads': (py, px), 'hw': (H, W), 'T': T, 'B': B}
            self._keep = None       # every ConvLSTM layer applied the pending mask while it read its state
            if not self.prep_batch:
                self._own_states()
        return logits

    def _wgrad_side(self, *tensors):
        """Context for the weight-gradient launches of backward.  Nothing in backward consumes a weight gradient -- only the
        optimiser (and the DP buckets) do -- so they run on a second HIP stream: the MFMA-bound gradient GEMMs then overlap
        with the HBM-bound chain the main stream continues with (BatchNorm backward passes, gate backward, slab reduces,
        resizes) and fill the tails of the dgrad launches.  Same kernels on the same data: results are unchanged.
        `tensors`: main-stream tensors the side-stream launches read.  They are HELD (a reference each) until an event
        recorded behind those launches has completed -- tensor.record_stream() would do the same inside the caching
        allocator, but its deferred frees made config-4 (832x992, 194 GB resident) 50 % slower: measured in round 2 (profiles/HISTORY.md)."""
        dev = self.flat_params.device
        if not self.overlap_wgrad or dev.type != 'cuda' or ops.EVENT_LOG is not None:
            return contextlib.nullcontext()
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=dev)
        return _SideLaunch(self, tensors)

    def _side_release(self, everything=False):
        keep = self._side_keep
        while keep and (everything or keep[0][0].query()):
            keep.pop(0)

    def _join_side(self):
        if self._side_stream is not None:
            torch.cuda.current_stream(self.flat_params.device).wait_stream(self._side_stream)
            self._side_keep.clear()      # whatever reuses that memory is main-stream work enqueued behind the join

    def backward(self, dlogits):
        """dlogits [T*B,H,W,last_depth] -> fills flat_grads (all trainable tensors)."""
        if self.tape is None:
            raise RuntimeError('backward() needs a training forward first')
        tp, self.tape = self.tape, None
        self._sync_weight_images()
        tape = tp['ops']
        py, px = tp['pads']
        H, W = tp['hw']
        plan = self.plan
        d = dlogits
        if any(py) or any(px):
            d = ops.window_copy(dlogits, (H + sum(py), W + sum(px)), (py[0], px[0]), 0)
        nd = len(plan['down'])
        g_down = [None] * nd    # gradient w.r.t. the output of each down block
        seg = 0
        # ---- decoder ----
        for bi in reversed(range(len(plan['up']))):
            blk = plan['up'][bi]
            for ci in reversed(range(len(blk['conv']))):
                rec = tape.pop()
                assert rec['kind'] == 'conv' and rec['prefix'] == f'up.{bi}' and rec['ci'] == ci
                if ci > 0:
                    (d,) = self._conv_unit_backward(rec, d, [True])
                else:
                    skip_level = nd - 2 - bi       # up0 <- D2 ... up(nd-1) <- image (level -1: no grad)
                    d_u, d_skip = self._conv_unit_backward(rec, d, [True, skip_level >= 0])
                    if skip_level >= 0:
                        g_down[skip_level] = d_skip
                    if blk['up_factor'] == 2:
                        urec = tape.pop()
                        assert urec['kind'] == 'up'
                        d = ops.upsample2x_bwd(d_u, urec['in_hw'], self.resize)
                    else:
                        d = d_u
            self._bucket_done(seg)
            seg += 1
        g_down[nd - 1] = d
        # ---- encoder ----
        for bi in reversed(range(nd)):
            blk = plan['down'][bi]
            d = g_down[bi]
            g_down[bi] = None
            if self.debug is not None:
                self.debug[f'g_down.{bi}'] = d.clone()
            for ci in reversed(range(len(blk['conv']))):
                rec = tape.pop()
                assert rec['kind'] == 'conv' and rec['prefix'] == f'down.{bi}' and rec['ci'] == ci
                (d,) = self._conv_unit_backward(rec, d, [True])
            for li in reversed(range(len(blk['lstm']))):
                rec = tape.pop()
                assert rec['kind'] == 'lstm' and rec['bi'] == bi and rec['li'] == li
                if self.debug is not None:
                    self.debug[f'dh_seq.{bi}.{li}'] = d.clone()
                d = self._lstm_backward(rec, d, need_dx=(bi > 0 or li > 0))
                if self.debug is not None and d is not None:
                    self.debug[f'lstm_dx.{bi}.{li}'] = d.clone()
            if bi > 0:
                ops.add_(g_down[bi - 1], d)
            self._bucket_done(seg)
            seg += 1
        assert not tape
        self._join_side()      # the optimiser step follows on the main stream

    def _bucket_done(self, seg):
        if self.on_bucket_ready is not None:
            self._join_side()      # the bucket's weight gradients ran on the side stream
            _, s, e = self.segments[seg]
            self.on_bucket_ready(s, e)

    # ------------------------------------------------------------------ recurrent state API
    def reset_states_per_batch(self, keep):
        """h, c *= keep[b]  (1 = clip continues, 0 = clip ended; Networks.py:77-84)."""
        if self._states is None:
            return
        keep = torch.as_tensor(keep, dtype=torch.float32).reshape(-1).to(self.device)
        self._state16.clear()
        self._keep = keep if self._keep is None else self._keep * keep      # applied by the next reader (see _own_states)
        if not self.prep_batch:
            self._own_states()

    def get_states(self):
        if self.states is None:
            return None
        out = []
        for blk in self.states:
            out.append([[None, None] if st is None else [st[0].cpu().numpy(), st[1].cpu().numpy()] for st in blk])
        return out

    def set_states(self, states):
        if self.states is None:     # not built yet: apply at first call
            self._pending_states = states
            return
        self._state16.clear()
        for bi, blk in enumerate(states):
            for li, st in enumerate(blk):
                if st is None or st[0] is None:
                    self.states[bi][li] = None      # reset_states(None) -> zeros
                else:
                    h = torch.as_tensor(np.asarray(st[0]), dtype=torch.float32).to(self.device).contiguous()
                    c = torch.as_tensor(np.asarray(st[1]), dtype=torch.float32).to(self.device).contiguous()
                    self.states[bi][li] = [h, c]


class Adam:
    """tf.keras Adam over the engine's flat buffers (train2D.py:61: lr from params, beta .9/.999, eps 1e-7)."""

    def __init__(self, engine, lr=1e-5, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.engine = engine
        self.lr, self.b1, self.b2, self.eps = lr, beta_1, beta_2, epsilon
        self.iterations = 0
        self.m = None
        self.v = None

    def apply_gradients(self, grad_scale=1.0):
        e = self.engine
        if self.m is None:
            self.m = torch.zeros_like(e.flat_params)
            self.v = torch.zeros_like(e.flat_params)
        self.iterations += 1
        t = self.iterations
        alpha = self.lr * math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        ops.adam_step(e.flat_params, e.flat_grads, self.m, self.v, alpha, self.b1, self.b2, self.eps, grad_scale)
        e.weights_changed()
