"""CPU tests of the host side: C-ABI surface, drop-in API contracts pinned by the reference-derived
goldens (parameter dicts, CLI flags, SEG metric, edge rule, bbox helpers), data contract, and the
world_size-2 gradient-bucket / loss-sum exchange on gloo."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def golden(golden_dir):
    with open(os.path.join(golden_dir, 'default_params.json')) as f:
        return json.load(f)


def test_cabi_library_exports_every_declared_symbol():
    """The HIP .so loads without a GPU and exports every function include/lstm_unet_hip.h declares."""
    from lu_native import build, cabi
    lib = cabi.bind(build.build(verbose=False))
    header = open(os.path.join(ROOT, 'include', 'lstm_unet_hip.h')).read()
    declared = set(re.findall(r'\b(lu_[a-z0-9_]+)\s*\(', header))
    declared -= {'lu_stream_t'}
    assert declared == set(cabi.PROTOTYPES.keys()), declared ^ set(cabi.PROTOTYPES.keys())
    for name in declared:
        assert hasattr(lib, name)
    assert lib.lu_abi_version() == cabi.ABI_VERSION
    import ctypes
    d = cabi.ConvDesc()
    assert lib.lu_conv2d_fwd(ctypes.byref(d), None) != 0          # argument validation, no GPU touched
    assert b'n_src' in lib.lu_last_error()


def test_ops_refuse_host_tensors_and_missing_device():
    from lu_native import ops
    import Networks
    with pytest.raises(ops.NativeError):
        ops.bn_stats(torch.zeros(4, 4))
    if not torch.cuda.is_available():
        with pytest.raises(ops.NativeError):
            Networks.ULSTMnet2D()(np.zeros((1, 1, 1, 8, 8), np.float32), True)


def test_default_param_dicts_match_reference(golden):
    import Networks
    import Params
    as_lists = lambda d: json.loads(json.dumps(d))   # noqa: E731  tuples -> lists
    assert as_lists(Networks.DEFAULT_NET_DOWN_PARAMS) == golden['DEFAULT_NET_DOWN_PARAMS']
    assert as_lists(Params.CTCParams.net_kernel_params) == golden['CTCParams.net_kernel_params']
    for k, v in golden['CTCParams.defaults'].items():
        assert as_lists(getattr(Params.CTCParams, k)) == v, k
    for k, v in golden['CTCInferenceParams.defaults'].items():
        assert as_lists(getattr(Params.CTCInferenceParams, k)) == v, k


def test_cli_flag_surface(golden):
    import train2D
    import Inference2D
    for mod, key in ((train2D, 'train2D.py'), (Inference2D, 'Inference2D.py')):
        mine = {kw['dest']: set(names) for names, kw in mod.FLAGS}
        for flag in golden['cli_flags'][key]:
            assert flag['dest'] in mine, flag
            assert set(flag['options']) <= mine[flag['dest']], flag
    ns = train2D.build_arg_parser().parse_args(['--crop_size', '64', '64', '--dataset', 'A', '01', 'B', '02',
                                                '--data_format', 'NWHC', '--class_weights', '.1', '.2', '.7'])
    assert ns.crop_size == [64, 64] and ns.train_sequence_list == [('A', '01'), ('B', '02')]
    with pytest.raises(ValueError):
        train2D.build_arg_parser().parse_args(['--dataset', 'A'])


def test_model_constructor_contract():
    import Networks
    m = Networks.ULSTMnet2D()
    assert m.total_stride == 8 and m.last_depth == 3 and len(m.DownLayers) == 4 and len(m.UpLayers) == 4
    assert [u.up_factor for u in m.UpLayers] == [1, 2, 2, 2] and m.UpLayers[-1].return_logits
    assert m.get_states() == [[[None, None]]] * 4
    bad = dict(Networks.DEFAULT_NET_DOWN_PARAMS)
    bad['up_conv_kernels'] = bad['up_conv_kernels'][:3]
    with pytest.raises(ValueError, match='up path'):
        Networks.ULSTMnet2D(bad)
    bad = dict(Networks.DEFAULT_NET_DOWN_PARAMS)
    bad['lstm_kernels'] = bad['lstm_kernels'][:2]
    with pytest.raises(ValueError, match='LSTM layers'):
        Networks.ULSTMnet2D(bad)
    d = Networks.DownBlock2D([(3, 16), (3, 32)], [(3, 16)], 2, 'NHWC')
    assert d.total_stride == 2 and len(d.Conv) == 2 and len(d.ConvLSTM) == 1
    import utils
    assert utils.get_model('ULSTMnet2D') is Networks.ULSTMnet2D


def test_plan_and_flat_layout():
    from lu_native.plan import make_plan, param_specs
    import Params
    plan = make_plan(Params.CTCParams.net_kernel_params, 1)
    specs = param_specs(plan)
    assert sum(int(np.prod(s)) for _, s, _ in specs) == 74606531
    assert [b['conv'][0]['cin'] for b in plan['up']] == [768, 512, 256, 65]
    names = [n for n, _, _ in specs]
    assert names[0].startswith('up.3.') and names[-1].startswith('down.0.')  # backward-completion order
    from lu_native.engine import model_pads
    assert model_pads(35, 35, 8, True) == ((8, 13), (8, 13)) and model_pads(256, 256, 8, False) == ((0, 0), (0, 0))


def test_seg_measure_goldens(golden_dir):
    import losses
    d = np.load(os.path.join(golden_dir, 'seg_unit_fixture.npz'))
    calc = losses.seg_measure(channel_axis=4)
    assert abs(float(calc(d['gt'], d['logits'])) - 0.59999996) < 1e-6
    r = np.load(os.path.join(golden_dir, 'seg_random.npz'))
    for gt, lg, exp in zip(r['gt'], r['logits'], r['seg']):
        got = float(calc(gt[..., None], lg))
        assert (np.isnan(got) and np.isnan(exp)) or abs(got - float(exp)) < 1e-6
    calc_nchw = losses.seg_measure(channel_axis=2)
    assert abs(float(calc_nchw(np.transpose(d['gt'], (0, 1, 4, 2, 3)), np.transpose(d['logits'], (0, 1, 4, 2, 3)))) -
               0.59999996) < 1e-6


def test_seg_measure_three_d_volumes():
    """losses.py:33-36: volumes [B, T, D, H, W], 6-connected components per (b, t) volume.  Two cells that touch only through the
    depth axis are ONE object; a depth-stacked prediction equal to the ground truth scores 1; a 2-D frame handled as a one-slice
    volume gives the 2-D metric."""
    import losses
    calc3 = losses.seg_measure(channel_axis=5, three_d=True)
    gt = np.zeros((1, 1, 3, 8, 8, 1), np.float32)
    gt[0, 0, 0, 1:4, 1:4, 0] = 1                      # slice 0
    gt[0, 0, 1, 1:4, 1:4, 0] = 1                      # same cell, next slice: connected through depth
    gt[0, 0, 2, 5:7, 5:7, 0] = 1                      # a second cell
    logits = np.zeros(gt.shape[:-1] + (3,), np.float32)
    logits[..., 1] = gt[..., 0] * 4 - 2
    assert abs(float(calc3(gt, logits)) - 1.0) < 1e-6
    half = logits.copy()
    half[0, 0, 1, ..., 1] = -2                        # the prediction misses slice 1 of the first cell: 9 of 18 voxels, not > 50 %
    assert abs(float(calc3(gt, half)) - 0.5) < 1e-6   # first cell scores 0, second 1
    rng = np.random.default_rng(3)
    g2 = (rng.random((2, 3, 16, 16, 1)) > 0.6).astype(np.float32)
    l2 = rng.standard_normal((2, 3, 16, 16, 3)).astype(np.float32)
    a = float(losses.seg_measure(channel_axis=4)(g2, l2))
    b = float(calc3(g2[:, :, None], l2[:, :, None]))
    assert abs(a - b) < 1e-6


def test_post_pipeline_depth_is_validated():
    import Inference2D
    with pytest.raises(ValueError):
        Inference2D.PostPipeline(depth=0)
    assert Inference2D.PostPipeline(depth=1).depth == 1


def test_launch_plan_cache_follows_its_knobs():
    """conv_splits / conv_plan / fused_step_cost_us are memoised -- on a key that carries SPLIT_CAP, SPLIT_MIN_IT,
    IT_US and FORCE_SPLITS, so an A/B tool that changes a knob between runs measures what it thinks it measures."""
    from lu_native import calls
    shape = (1, 34, 34, 2048, 5, 768)
    base = calls.conv_splits(*shape)
    assert base > 1 and calls.conv_plan(*shape)[0] >= 1
    old = (calls.SPLIT_CAP, calls.IT_US, calls.FORCE_SPLITS)
    try:
        calls.SPLIT_CAP = 2
        assert calls.conv_splits(*shape) <= 2 and calls.conv_plan(*shape)[0] <= 2
        calls.SPLIT_CAP = old[0]
        assert calls.conv_splits(*shape) == base
        calls.FORCE_SPLITS = 7
        assert calls.conv_splits(*shape) == 7 and calls.conv_plan(*shape)[0] == 7
        calls.FORCE_SPLITS = None
        c0 = calls.fused_step_cost_us(1, 34, 34, 512, 5, 768)
        calls.IT_US = 2 * old[1]
        assert abs(calls.fused_step_cost_us(1, 34, 34, 512, 5, 768) - 2 * c0) < 1e-6 * c0
    finally:
        calls.SPLIT_CAP, calls.IT_US, calls.FORCE_SPLITS = old
    assert calls.conv_splits(*shape) == base


def test_edge_rule_bbox_and_data_contract(golden_dir):
    import DataHandeling
    import utils
    d = np.load(os.path.join(golden_dir, 'edge_rule.npz'))
    for inst, cls in zip(d['inst'], d['classes']):
        assert np.array_equal(DataHandeling.instances_to_classes(inst), cls)
    b = np.load(os.path.join(golden_dir, 'bbox.npz'))
    crop, loc = utils.bbox_crop(b['img'], margin=3)
    assert np.array_equal(crop, b['crop']) and tuple(loc) == tuple(b['loc'])
    assert np.array_equal(utils.bbox_fill(b['img'].astype(np.float32), np.ones_like(crop, np.float32), loc), b['filled'])
    prov = DataHandeling.SyntheticSequence2D(image_crop_size=(32, 40), unroll_len=4, batch_size=3, data_format='NCHW',
                                             clip_len=8)
    img, seg, full, keep = prov.get_batch()
    assert img.shape == seg.shape == (3, 4, 1, 32, 40) and full.shape == (3, 4) and keep.shape == (3,)
    assert img.dtype == np.float32 and set(np.unique(seg)) <= {-1.0, 0.0, 1.0, 2.0}
    assert np.allclose(img.mean(axis=(2, 3, 4)), 0, atol=1e-5) and np.allclose(img.std(axis=(2, 3, 4)), 1, atol=1e-4)
    assert np.all(keep == 1.0)
    _, _, _, keep2 = prov.get_batch()
    assert np.all(keep2 == 0.0)              # clip_len 8 = two windows: second window ends every clip
    nhwc = DataHandeling.SyntheticSequence2D(image_crop_size=(16, 16), unroll_len=2, batch_size=1, data_format='NHWC')
    assert nhwc.get_batch()[0].shape == (1, 2, 16, 16, 1)


def test_inference_postprocess_needs_the_device():
    """The post-processing runs on the GPU (tests/test_postprocess.py); without a device it fails loudly, no CPU path."""
    import Inference2D
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is present')
    with pytest.raises(Exception, match='no HIP device|no CPU'):
        Inference2D.postprocess(np.zeros((3, 8, 8), np.float32))


DP_WORKER = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, 'lstm-unet_amd'))
import torch
from lu_native.dp import DataParallel
dp = DataParallel(backend='gloo', bucket_bytes=1024)
assert dp.world_size == 2
flat = torch.arange(1000, dtype=torch.float32) * (dp.rank + 1)
dp.attach(flat)
segs = [(0, 100), (100, 400), (400, 404), (404, 1000)]     # backward-completion order, contiguous
for s, e in segs:
    dp.bucket_ready(s, e)
dp.finish()
exp = torch.arange(1000, dtype=torch.float32) * 3
assert torch.equal(flat, exp), (flat[:5], exp[:5])
assert dp.launched == 2, dp.launched          # (0,400) | (400,1000): the 16-byte piece joined its neighbour
# the opposite direction (a caller walking the flat buffer from its end): adjacent ranges still merge
flat2 = torch.arange(1000, dtype=torch.float32) * (dp.rank + 1)
dp.attach(flat2)
n0 = dp.launched
for s, e in [(990, 1000), (900, 990), (600, 900), (500, 600), (0, 500)]:
    dp.bucket_ready(s, e)
dp.finish()
assert torch.equal(flat2, exp) and dp.launched - n0 == 2, dp.launched - n0     # (600,1000) | (0,600)
sums = torch.tensor([1.0 + dp.rank, 10.0], dtype=torch.float64)
dp.all_reduce_(sums)
assert sums.tolist() == [3.0, 20.0]
assert list(dp.shard_slots(8)) == list(range(dp.rank * 4, dp.rank * 4 + 4))
w = torch.full((4,), float(dp.rank))
dp.broadcast_(w, 0)
assert w.sum().item() == 0.0
dp.barrier()
print('rank', dp.rank, 'ok')
'''


def test_dp_bucketed_allreduce_world2_gloo(tmp_path):
    script = tmp_path / 'dp_worker.py'
    script.write_text(DP_WORKER % {'root': ROOT})
    port = 29500 + os.getpid() % 2000
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert 'ok' in o


def test_tensorboard_event_writer(tmp_path):
    """tb_events: CRC-32C known answer, TFRecord framing and proto fields read back (scalars exact in float32)."""
    import tb_events
    assert tb_events.crc32c(b'123456789') == 0xE3069283 and tb_events.crc32c(b'') == 0
    w = tb_events.SummaryWriter(str(tmp_path))
    w.scalar('Loss', 0.15625, 3)
    w.image('Image', np.linspace(0, 1, 6 * 7).reshape(6, 7), 3)
    w.image('GT', np.zeros((6, 7, 3), np.float32), 5)
    w.close()
    ev = tb_events.read_events(w.path)
    assert ev[0][2] == b'brain.Event:2' and ev[1][:2] == (3, {'Loss': 0.15625})
    assert ev[2][0] == 3 and ev[2][1]['Image'][:3] == (6, 7, 1) and ev[3][0] == 5 and ev[3][1]['GT'][:3] == (6, 7, 3)
    from PIL import Image
    import io
    img = np.asarray(Image.open(io.BytesIO(ev[2][1]['Image'][3])))
    assert img.shape == (6, 7) and img[0, 0] == 0 and img[-1, -1] == 255
