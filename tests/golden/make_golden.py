"""Generate golden fixtures by IMPORTING the reference (arbellea/LSTM-UNet at /root/reference)
in the build container.  Run once here; only the small .npz/.json outputs travel (the
reference itself never does).  TensorFlow and OpenCV are absent, so stub modules are put
in sys.modules: only the reference's pure numpy/scipy code is executed.

    python tests/golden/make_golden.py

Outputs (all in tests/golden/):
  seg_unit_fixture.npz   inputs of losses.seg_measure_unit_test (losses.py:91-115) + SEG value
  seg_random.npz         12 random-mask cases through the reference's seg_numpy closure
  edge_rule.npz          DataHandeling.CTCRAMReaderSequence2D._fix_transformed_segmentation cases
  bbox.npz               utils.bbox_crop / bbox_fill cases
  default_params.json    Networks.DEFAULT_NET_DOWN_PARAMS and Params.CTCParams defaults
  elastic.npz            the importable augmentation helpers of DataHandeling.py:152-197,239-260 with a seeded RandomState:
                         control points of the random affine (as handed to cv2.getAffineTransform), elastic sampling
                         coordinates, brightness / contrast, and the resampling half of _get_transformed_image_ (the
                         cv2.warpAffine call is recorded -- dsize, border mode / value, interpolation flag -- and passed
                         through as the identity warp, OpenCV being absent)
"""
import json
import os
import sys
import types

import numpy as np

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    class _Any:
        def __getattr__(self, name):
            return _Any()

        def __call__(self, *a, **k):
            return _Any()

    class _AnyModule(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith('__'):
                raise AttributeError(name)
            return _Any()

    tf = _AnyModule('tensorflow')
    tf.__version__ = '2.0.0'
    tf_python = _AnyModule('tensorflow.python')
    keras = _AnyModule('tensorflow.python.keras')

    keras.Model = object
    keras.layers = _Any()
    keras.backend = _Any()
    tf.python = tf_python
    tf_python.keras = keras
    tf.keras = keras
    for name, mod in [('tensorflow', tf), ('tensorflow.python', tf_python), ('tensorflow.python.keras', keras),
                      ('tensorflow.keras', keras), ('cv2', _AnyModule('cv2')),
                      ('requests', _AnyModule('requests'))]:
        sys.modules.setdefault(name, mod)


def _closure_fn(fn, name):
    for cell, var in zip(fn.__closure__, fn.__code__.co_freevars):
        if var == name:
            return cell.cell_contents
    raise KeyError(name)


class _FakeTensor:
    def __init__(self, a):
        self._a = a

    def numpy(self):
        return self._a


def main():
    _install_stubs()
    sys.path.insert(0, REF)
    import losses as ref_losses
    import Networks as ref_nets
    import utils as ref_utils
    import DataHandeling as ref_data
    import Params as ref_params

    calc = ref_losses.seg_measure(channel_axis=4, three_d=False, foreground_class_index=1)
    seg_numpy = _closure_fn(calc, 'seg_numpy')

    def ref_seg(gt, logits):
        """gt [B,T,H,W] float, logits [B,T,H,W,3]: the tf part of calc_seg restated with numpy
        (squeeze/valid/argmax are trivially numpy), then the reference's own seg_numpy."""
        valid = (gt > -1).astype(np.float32)
        gt_fg = (gt * valid) == 1
        pred_fg = np.argmax(logits, axis=-1) == 1
        return np.float32(seg_numpy(_FakeTensor(gt_fg), _FakeTensor(pred_fg)))

    # ---- the reference's own unit-test fixture (losses.py:96-114) ----
    h = w = 30
    bsz, unroll = 3, 2
    gt = np.zeros((bsz, unroll, h, w, 1), np.float32)
    out = np.zeros((bsz, unroll, h, w, 3), np.float32)
    out[:, :, :, :, 0] = 0.25
    objects = [(12, 20, 0, 5), (0, 9, 0, 5), (12, 20, 9, 20), (0, 9, 9, 20)]
    i = 0
    for b in range(bsz):
        for u in range(unroll):
            for obj_id, (xs, xe, ys, ye) in enumerate(objects):
                gt[b, u, ys + i:ye + i, xs + i:xe + i] = obj_id + 1
                out[b, u, max(ys + i + 2, 0):max(ye + i, 0), max(xs + i, 0):max(xe + i, 0), 1] = 0.5
            i += 1
    seg_val = ref_seg(gt[..., 0], out)
    print('reference SEG unit fixture =', repr(seg_val))
    np.savez_compressed(os.path.join(OUT, 'seg_unit_fixture.npz'), gt=gt, logits=out, seg=seg_val)

    # ---- random-mask SEG goldens ----
    rng = np.random.default_rng(7)
    gts, lgs, vals = [], [], []
    for case in range(12):
        hh = ww = 48
        g = np.zeros((2, 2, hh, ww), np.float32)
        lg = np.zeros((2, 2, hh, ww, 3), np.float32)
        lg[..., 0] = 0.25
        if case != 0:  # case 0: no GT objects -> NaN
            for bb in range(2):
                for tt in range(2):
                    for _ in range(int(rng.integers(1, 5))):
                        cy, cx = rng.integers(6, hh - 6, size=2)
                        ry, rx = rng.integers(2, 7, size=2)
                        yy, xx = np.mgrid[:hh, :ww]
                        m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1
                        g[bb, tt][m] = 1
                        # prediction: shifted / shrunk / split copy
                        dy, dx = rng.integers(-3, 4, size=2)
                        m2 = ((yy - cy - dy) / max(ry - (case % 3), 1)) ** 2 + ((xx - cx - dx) / rx) ** 2 <= 1
                        if case % 4 == 1:
                            m2[:, cx] = False  # split the predicted object
                        lg[bb, tt][m2, 1] += 2.0
                    if case % 5 == 2:
                        g[bb, tt][: hh // 4] = -1  # unlabeled band
        gts.append(g)
        lgs.append(lg)
        vals.append(ref_seg(g, lg))
    print('random SEG goldens:', vals)
    np.savez_compressed(os.path.join(OUT, 'seg_random.npz'), gt=np.stack(gts), logits=np.stack(lgs),
                        seg=np.array(vals, np.float32))

    # ---- edge rule ----
    fix = ref_data.CTCRAMReaderSequence2D._fix_transformed_segmentation
    insts, outs = [], []
    for case in range(6):
        inst = np.zeros((40, 40), np.float32)
        yy, xx = np.mgrid[:40, :40]
        for lab in range(1, int(rng.integers(2, 6))):
            cy, cx = rng.integers(5, 35, size=2)
            r = int(rng.integers(3, 8))
            inst[(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = lab
        if case == 0:
            inst[:] = 0
            inst[10:20, 10:20] = 1
            inst[10:20, 20:30] = 2  # touching cells
        if case == 1:
            inst = inst + 0.3 * (rng.random(inst.shape) < 0.05)  # fractional values get rounded
        insts.append(inst)
        outs.append(fix(inst.copy()))
    np.savez_compressed(os.path.join(OUT, 'edge_rule.npz'), inst=np.stack(insts), classes=np.stack(outs))

    # ---- bbox helpers ----
    img = np.zeros((30, 40), bool)
    img[8:15, 12:30] = True
    img[10:12, 15:20] = False
    crop, loc = ref_utils.bbox_crop(img, margin=3)
    filled = ref_utils.bbox_fill(img.astype(np.float32), np.ones_like(crop, np.float32), loc)
    np.savez_compressed(os.path.join(OUT, 'bbox.npz'), img=img, crop=crop, loc=np.array(loc), filled=filled)

    # ---- augmentation helpers: seeded RandomState in place of RandomState(None) ----
    cv2 = sys.modules['cv2']
    cv2.BORDER_CONSTANT, cv2.BORDER_REFLECT_101, cv2.INTER_NEAREST, cv2.INTER_LINEAR = 0, 4, 0, 1      # OpenCV's enum values
    R = ref_data.CTCRAMReaderSequence2D
    real_state = np.random.RandomState
    el = {}
    for ci, (shape, seed) in enumerate([((32, 32), 11), ((24, 40), 12), ((40, 24), 13)]):
        seen = {}

        def get_affine(p1, p2):
            seen['pts1'], seen['pts2'] = np.array(p1), np.array(p2)
            return np.float64([[1, 0, 0], [0, 1, 0]])

        cv2.getAffineTransform = get_affine
        np.random.RandomState = lambda s_=None: real_state(seed)
        try:
            _, state = R._get_elastic_affine_matrix_(shape, shape[1] * 0.08)
        finally:
            np.random.RandomState = real_state
        idx = R._get_indices4elastic_transform(shape, shape[1] * 2, shape[1] * 0.15, state)
        calls = []

        def warp(image, m, dsize, **kw):
            calls.append((tuple(dsize), kw.get('borderMode'), kw.get('borderValue', 0), kw.get('flags', cv2.INTER_LINEAR)))
            return image

        cv2.warpAffine = warp
        img = rng.standard_normal(shape)
        lab = np.zeros(shape, np.float32)
        lab[shape[0] // 4: shape[0] // 2, shape[1] // 4: 3 * shape[1] // 4] = 1
        lab[shape[0] // 2 + 2: shape[0] - 3, 3: shape[1] // 2] = 2
        out_img = R._get_transformed_image_(img, None, idx, seg=False)
        out_lab = R._get_transformed_image_(lab, None, idx, seg=True)
        el.update({'shape%d' % ci: np.array(shape), 'seed%d' % ci: seed, 'pts1_%d' % ci: seen['pts1'], 'pts2_%d' % ci: seen['pts2'],
                   'rows%d' % ci: idx[0], 'cols%d' % ci: idx[1], 'img%d' % ci: img, 'lab%d' % ci: lab,
                   'out_img%d' % ci: out_img, 'out_lab%d' % ci: out_lab,
                   'warp_calls%d' % ci: np.array([[c[0][0], c[0][1], c[1], c[2], c[3]] for c in calls], np.float64)})
    frame = rng.standard_normal((20, 28)).astype(np.float64) * 3 + 1
    el['pc_in'] = frame
    el['pc_factor'], el['pc_delta'] = 1.37, -0.42
    el['pc_contrast'] = R._adjust_contrast_(frame, 1.37)
    el['pc_both'] = R._adjust_brightness_(R._adjust_contrast_(frame, 1.37), -0.42)
    np.savez_compressed(os.path.join(OUT, 'elastic.npz'), **el)

    # ---- parameter dicts ----
    P = ref_params.CTCParams
    dump = {
        'DEFAULT_NET_DOWN_PARAMS': ref_nets.DEFAULT_NET_DOWN_PARAMS,
        'CTCParams.net_kernel_params': P.net_kernel_params,
        'CTCParams.defaults': {k: getattr(P, k) for k in (
            'crop_size', 'batch_size', 'unroll_len', 'data_format', 'class_weights', 'learning_rate',
            'num_iterations', 'validation_interval', 'print_to_console_interval', 'save_checkpoint_iteration',
            'save_checkpoint_every_N_hours', 'save_checkpoint_max_to_keep', 'write_to_tb_interval',
            'train_q_capacity', 'val_q_capacity', 'num_val_threads', 'num_train_threads', 'tb_sub_folder',
            'dry_run', 'profile', 'load_checkpoint', 'continue_run', 'experiment_name', 'gpu_id')},
        'CTCInferenceParams.defaults': {k: getattr(ref_params.CTCInferenceParams, k) for k in (
            'gpu_id', 'filename_format', 'data_format', 'FOV', 'min_cell_size', 'max_cell_size', 'edge_dist',
            'pre_sequence_frames', 'dry_run', 'save_intermediate')},
    }
    # ---- CLI flag surface (option strings + dest), extracted from the reference's argparse calls ----
    import re
    flag_re = re.compile(r"add_argument\(([^)]*?)dest='(\w+)'", re.S)
    cli = {}
    for fname in ('train2D.py', 'Inference2D.py'):
        text = open(os.path.join(REF, fname)).read()
        flags = []
        for m in flag_re.finditer(text):
            opts = re.findall(r"'(--?[\w]+)'", m.group(1))
            flags.append({'options': opts, 'dest': m.group(2)})
        cli[fname] = flags
    dump['cli_flags'] = cli
    with open(os.path.join(OUT, 'default_params.json'), 'w') as f:
        json.dump(dump, f, indent=1, sort_keys=True)
    print('wrote goldens to', OUT)


if __name__ == '__main__':
    main()
