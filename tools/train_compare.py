"""fp32 vs bf16-mode training on the same synthetic clip stream (Params.py network, 128x128, T=4, B=4, lr 1e-4):
loss curves, then held-out agreement of the two trained models (per-pixel argmax agreement, 3-class IoU, SEG measure).
Evidence for the bf16 accuracy contract of DESIGN.md §3.3 / SURVEY §8c ("bf16 path judged on IoU / argmax agreement").
usage: python tools/train_compare.py [steps] [precision = bf16 | bf16x3] > profiles/r01_bf16_vs_fp32_training.json
(second argument bf16x3: the same comparison for the fp32-arithmetic-on-the-bf16-MFMA mode of DESIGN 3.3a; the keys keep the name 'bf16')"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'lstm-unet_amd'))
import numpy as np
import torch
import DataHandeling
import Params
import losses
import train2D

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
OTHER = sys.argv[2] if len(sys.argv) > 2 else 'bf16'
net = Params.CTCParams.net_kernel_params
H = W = 128
B, T = 4, 4


def provider(seed):
    return DataHandeling.SyntheticSequence2D(image_crop_size=(H, W), unroll_len=T, batch_size=B, data_format='NCHW', seed=seed)


def run(precision):
    tr = train2D.Trainer(Params.CTCParams.net_model, net, 'NCHW', Params.CTCParams.class_weights, 1e-4, seed=0,
                         precision=precision)
    data = provider(11)
    curve = []
    for i in range(steps):
        img, seg, _, keep = data.get_batch()
        _, _, loss = tr.train_step(img, seg, want_outputs=True)
        tr.model.reset_states_per_batch(keep)
        curve.append(float(loss))
    # held-out clips, fresh recurrent state; BatchNorm with batch statistics (after a few hundred steps Keras' moving
    # averages, momentum 0.99, still carry ~10 % of their initial values, which would dominate the comparison)
    from lu_native import ops
    held = provider(999)
    tr.model.reset_states_per_batch(np.zeros(B, np.float32))
    preds, losses_, segs = [], [], []
    for _ in range(8):
        img, seg, _, keep = held.get_batch()
        x_tb, g, T_, B_ = tr._prep(img, seg)
        logits = tr.engine.forward(x_tb, T_, B_, True)
        tr.engine.tape = None
        sums, _ = ops.wce_forward(logits.view(-1, 3), g, tr._cw, False)
        losses_.append(float(ops.wce_loss(sums).cpu()[0]))
        lg = logits.view(T_, B_, H, W, 3).permute(1, 0, 2, 3, 4).contiguous()       # [B,T,H,W,3]
        preds.append(lg.cpu().numpy())
        segs.append(seg)
        tr.model.reset_states_per_batch(keep)
    return curve, np.concatenate(preds), float(np.mean(losses_)), np.concatenate(segs)


def iou3(pred_logits, gt):
    am = pred_logits.argmax(-1) if pred_logits.shape[-1] == 3 else pred_logits.argmax(2)
    gt = gt[:, :, 0] if gt.ndim == 5 else gt
    out = []
    for c in range(3):
        m = gt >= 0
        inter = ((am == c) & (gt == c) & m).sum()
        union = (((am == c) | (gt == c)) & m).sum()
        out.append(float(inter) / max(float(union), 1.0))
    return out, am


c32, p32, v32, gt = run('fp32')
c16, p16, v16, _ = run(OTHER)
i32, a32 = iou3(p32, gt)
i16, a16 = iou3(p16, gt)
res = {
    'what': 'fp32 vs %s: same init (seed 0), same synthetic stream, %d optimiser steps at lr 1e-4, 128x128 T=4 B=4, Params.py network' % (OTHER, steps),
    'train_loss_fp32_every10': [round(float(np.mean(c32[i:i + 10])), 5) for i in range(0, steps, 10)],
    'train_loss_bf16_every10': [round(float(np.mean(c16[i:i + 10])), 5) for i in range(0, steps, 10)],
    'heldout_loss': {'fp32': round(v32, 5), 'bf16': round(v16, 5)},
    'heldout_iou_bg_cell_edge': {'fp32': [round(v, 4) for v in i32], 'bf16': [round(v, 4) for v in i16]},
    'heldout_argmax_agreement_bf16_vs_fp32': round(float((a32 == a16).mean()), 5),
}
print(json.dumps(res, indent=1))
