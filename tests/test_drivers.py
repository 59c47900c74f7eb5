"""Driver-level contracts of train2D.py / Inference2D.py (loop, validation state swap, checkpoints, saved-model
directory, streaming inference from image files) on both backends ('emu' tiny on CPU, 'hip' on the GPU)."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import tiny_net
from engine_backend import engine_backend

BACKENDS = ['emu', pytest.param('hip', marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def dev(request):
    with engine_backend(request.param) as d:
        yield d


def test_train_loop_checkpoint_and_inference_roundtrip(dev, tmp_path, monkeypatch):
    import DataHandeling
    import Inference2D
    import Networks
    import Params
    import train2D
    from PIL import Image
    big = dev.type == 'cuda'
    size = 32 if big else 16
    net = tiny_net(3, (32, 16, 16, 32), (16, 8, 8, 8)) if big else tiny_net(3)
    monkeypatch.setattr(Params.CTCParams, 'net_kernel_params', net)
    args = dict(experiment_name='t', crop_size=(size, size), batch_size=2, unroll_len=2, num_iterations=3,
                validation_interval=2, print_to_console_interval=1, save_checkpoint_iteration=2,
                save_checkpoint_dir=str(tmp_path), save_log_dir=str(tmp_path), data_format='NCHW',
                learning_rate=1e-3, write_to_tb_interval=2, resize='half_pixel')
    params = Params.CTCParams(args)
    assert isinstance(params.train_data_provider, DataHandeling.SyntheticSequence2D) and params.channel_axis == 1
    trainer = train2D.train(params)
    assert trainer.step == 4          # range(step, num_iterations + 1), as the reference loop (train2D.py:145)
    save_dir = params.experiment_save_dir
    assert os.path.exists(os.path.join(save_dir, 'model.ckpt.index'))      # a TensorFlow tensor bundle, like the reference's
    with open(os.path.join(save_dir, 'model_params.pickle'), 'rb') as f:
        meta = pickle.load(f)
    assert meta['name'] == 'ULSTMnet2D' and meta['params'][0] == net
    # the bilinear convention the weights were trained under travels with them ...
    assert meta['resize'] == 'half_pixel' == trainer.engine.resize and meta['precision'] == 'fp32'
    logged = []
    monkeypatch.setattr(Inference2D, 'log_print', lambda *a: logged.append(' '.join(map(str, a))))
    assert Inference2D.resolve_resize(meta) == 'half_pixel' and not logged
    assert Inference2D.resolve_resize(meta, 'tf2.0') == 'tf2.0' and 'trained with' in logged[-1]
    # ... and a pickle without the record (the reference's own, train2D.py:236-239) is run under 'tf2.0', loudly
    assert Inference2D.resolve_resize({'name': 'ULSTMnet2D', 'params': (net,)}) == 'tf2.0' and 'does not record' in logged[-1]
    monkeypatch.undo()
    monkeypatch.setattr(Params.CTCParams, 'net_kernel_params', net)
    ckpts = sorted(os.listdir(os.path.join(save_dir, 'tf_ckpts')))
    assert ckpts, 'no periodic checkpoint written'
    # TensorBoard event files: Loss / SEG scalars and Image / GT / Output images at steps 2 and 4, train and val
    import tb_events
    for run in ('train', 'val'):
        files = os.listdir(os.path.join(params.experiment_log_dir, run))
        assert len(files) == 1 and files[0].startswith('events.out.tfevents.')
        events = tb_events.read_events(os.path.join(params.experiment_log_dir, run, files[0]))      # verifies every CRC
        assert events[0][2] == b'brain.Event:2'
        tags = {}
        for step, vals, _ in events[1:]:
            for k, v in vals.items():
                tags.setdefault(k, []).append(step)
        assert set(tags) == {'Loss', 'SEG', 'Image', 'GT', 'Output'} and tags['Loss'] == [2, 4], (run, tags)
        h, w, ch, png = [v for _, vals, _ in events for k, v in vals.items() if k == 'Output'][0]
        assert (h, w, ch) == (size, size, 3) and png[:8] == b'\x89PNG\r\n\x1a\n'
    # resume from the checkpoint: weights / Adam state / step / recurrent state come back
    sd = torch.load(os.path.join(save_dir, 'tf_ckpts', ckpts[-1]), map_location='cpu')
    t2 = train2D.Trainer(Networks.ULSTMnet2D, net, 'NCHW', learning_rate=1e-3)
    t2.load_state_dict(sd)
    assert t2.step == sd['step'] and t2.optimizer.iterations == sd['adam_iterations']
    assert torch.equal(t2.engine.flat_params.cpu(), sd['params'])
    # saved-model directory -> streaming inference over image files (Inference2D.py:27-62)
    seq_dir = tmp_path / 'seq'
    seq_dir.mkdir()
    rng = np.random.default_rng(0)
    for t in range(3):
        Image.fromarray((rng.random((size + 3, size + 5)) * 255).astype(np.uint8)).save(seq_dir / ('t%03d.tif' % t))
    out_dir = tmp_path / 'out'
    iparams = Params.CTCInferenceParams(dict(model_path=save_dir, sequence_path=str(seq_dir), output_path=str(out_dir),
                                             save_intermediate=False, pre_sequence_frames=2, min_cell_size=1,
                                             max_cell_size=10 ** 6, data_format='NCHW'))
    Inference2D.inference(iparams)
    masks = sorted(os.listdir(out_dir))
    assert masks == ['mask000.tif', 'mask001.tif', 'mask002.tif']
    m = np.asarray(Image.open(out_dir / masks[0]))
    assert m.shape == (size + 3, size + 5) and m.dtype == np.uint16


def test_single_process_reader_error_is_checkpointed_and_reraised_as_itself(dev, tmp_path, monkeypatch):
    """A reader failure the reference does not handle (the worker threads' RuntimeError), ONE process: the loop checkpoints and
    then re-raises the reader's own exception -- not the AgreedFailure wrapper the multi-rank agreement uses (callers
    that catch the concrete type must keep working); a ValueError still ends the loop the reference's way (save, return)."""
    import Params
    import train2D
    big = dev.type == 'cuda'
    size = 32 if big else 16
    monkeypatch.setattr(Params.CTCParams, 'net_kernel_params', tiny_net(3, (32, 16, 16, 32), (16, 8, 8, 8)) if big else tiny_net(3))
    for kind in (RuntimeError, ValueError):
        args = dict(experiment_name='t', crop_size=(size, size), batch_size=1, unroll_len=2, num_iterations=6, validation_interval=100,
                    print_to_console_interval=1, save_checkpoint_iteration=100, save_checkpoint_dir=str(tmp_path / kind.__name__),
                    save_log_dir=str(tmp_path / kind.__name__), data_format='NCHW', learning_rate=1e-3, write_to_tb_interval=100)
        params = Params.CTCParams(args)
        real, calls = params.train_data_provider.get_batch, [0]

        def flaky():
            calls[0] += 1
            if calls[0] == 3:
                raise kind('non-finite: the reader workers stopped')
            return real()
        params.train_data_provider.get_batch = flaky
        logged = []
        monkeypatch.setattr(train2D, 'log_print', lambda *a: logged.append(' '.join(map(str, a))))
        if kind is RuntimeError:
            with pytest.raises(RuntimeError, match='reader workers stopped') as ei:
                train2D.train(params)
            assert type(ei.value) is RuntimeError
        else:
            assert train2D.train(params).step == 2      # two completed steps, then the third batch fails
        assert any(m.startswith('Saving Model Before closing due to error: non-finite') for m in logged), logged
        assert os.path.exists(os.path.join(params.experiment_save_dir, 'model.ckpt.index'))


DP_LOOP_WORKER = r'''
import os, sys
ROOT = %(root)r
for p in (ROOT, os.path.join(ROOT, 'lstm-unet_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np, torch
from engine_backend import engine_backend
from conftest import tiny_net
import Params, train2D
from lu_native.dp import DataParallel
rank = int(os.environ['RANK'])
world = int(os.environ['WORLD_SIZE'])
with engine_backend(os.environ.get('LU_TEST_BACKEND', 'emu')):
    Params.CTCParams.net_kernel_params = tiny_net(3)
    params = Params.CTCParams(dict(experiment_name='t', crop_size=(16, 16), batch_size=1, unroll_len=2, num_iterations=8,
                                   validation_interval=100, print_to_console_interval=1, save_checkpoint_iteration=1,
                                   save_checkpoint_max_to_keep=2, save_checkpoint_dir=%(tmp)r, save_log_dir=%(tmp)r,
                                   data_format='NCHW', learning_rate=1e-3, write_to_tb_interval=100))
    prov = params.train_data_provider
    real, calls = prov.get_batch, [0]
    def flaky():
        calls[0] += 1
        if rank == world - 1 and calls[0] == 5:
            if os.environ['LU_TEST_ERR'] == 'ValueError':
                raise ValueError('non-finite values in frame 3 after augmentation')
            raise RuntimeError('non-finite: the reader workers stopped')      # DataHandeling._next_item's type
        return real()
    prov.get_batch = flaky
    collectives = []
    orig = DataParallel.all_reduce_
    DataParallel.all_reduce_ = lambda self, t: (collectives.append(tuple(t.shape)), orig(self, t))[1]
    made = []
    real_trainer = train2D.Trainer
    train2D.Trainer = lambda *a, **k: (made.append(real_trainer(*a, **k)), made[-1])[1]
    try:
        train2D.train(params)
        assert os.environ['LU_TEST_ERR'] == 'ValueError'      # a type the reference handles: EVERY rank saves and returns
        print('returned normally', flush=True)
    except train2D.AgreedFailure as exc:       # not one of the reference's three: checkpointed like them, then re-raised on EVERY rank
        assert os.environ['LU_TEST_ERR'] == 'RuntimeError' and not exc.handled, exc
        assert (rank == world - 1) == isinstance(exc.__cause__, RuntimeError)      # the failing rank carries its own error as the cause
        print('re-raised:', exc, flush=True)
    trainer = made[0]
    n_bn = len(trainer.engine.S)
    # after the agreed failure: exactly one more round of BN-statistics averaging -- by the error-path checkpoint, which is
    # collective when (and only when) the failure was agreed on; the `finally` does not repeat it
    tail = collectives[-n_bn:]
    print('RANK', rank, 'steps', trainer.step, 'tail', len(tail), flush=True)
    np.save(os.path.join(%(tmp)r, 'done_%%d.npy' %% rank), np.array([trainer.step, len(collectives)]))
'''


@pytest.mark.parametrize('err_kind', ['ValueError', 'RuntimeError'])
def test_dp_loop_failure_on_one_rank_stops_all_ranks_without_hanging(tmp_path, err_kind):
    """A data error on ONE rank -- the reader's ValueError, or the RuntimeError / OSError its worker threads hand over -- must
    end the loop on every rank at the same step; the checkpoint of an AGREED failure averages the BatchNorm statistics like a
    regular one (all ranks are in step), and per-rank state files rotate with save_checkpoint_max_to_keep (train2D.py:145-246
    is single-device)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'worker.py'
    script.write_text(DP_LOOP_WORKER % {'root': root, 'tmp': str(tmp_path)})
    port = 29700 + os.getpid() % 1200 + (0 if err_kind == 'ValueError' else 1201)
    procs = [subprocess.Popen([sys.executable, str(script)],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), LU_DP_BACKEND='gloo',
                                       LU_TEST_ERR=err_kind, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    a, b = np.load(tmp_path / 'done_0.npy'), np.load(tmp_path / 'done_1.npy')
    assert a[0] == b[0] == 4 and a[1] == b[1]              # four completed steps on both ranks, same number of collectives
    assert 'another data-parallel rank reported an error' in outs[0] and 'non-finite' in outs[1]
    # one outcome for one agreed failure: both ranks return, or both ranks re-raise
    marker = 'returned normally' if err_kind == 'ValueError' else 're-raised:'
    assert all(marker in o for o in outs), outs
    ck = os.listdir(os.path.join(str(tmp_path), 'LSTMUNet', 't'))
    assert len(ck) == 1                                     # one run directory for the job (rank 0's time stamp)
    run_dir = os.path.join(str(tmp_path), 'LSTMUNet', 't', ck[0])
    files = sorted(os.listdir(os.path.join(run_dir, 'tf_ckpts')))
    for r in range(2):
        mine = [f for f in files if f.endswith('.rank%d.pt' % r)]
        assert len(mine) == 2, files                        # rotated like ckpt-*.pt (max_to_keep = 2)
    assert os.path.exists(os.path.join(run_dir, 'model.ckpt.index'))


def _launch_dp8_loop(tmp_path):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'worker.py'
    script.write_text(DP_LOOP_WORKER % {'root': root, 'tmp': str(tmp_path)})
    port = 29700 + (os.getpid() + 77) % 1200
    W = 8
    procs = [subprocess.Popen([sys.executable, str(script)],
                              env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(W), LOCAL_RANK='0', LU_DP_BACKEND='gloo',
                                       LU_TEST_BACKEND='hip', LU_TEST_ERR='ValueError', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(W)]
    return {'tmp': tmp_path, 'procs': procs}


@pytest.mark.gpu
def test_dp8_loop_on_one_gpu_failure_agreement_and_per_rank_state_files(tmp_path):
    """EIGHT ranks over gloo on the one GPU of the test box, HIP kernels underneath (make the first
    8-GPU box boring): the training loop of train2D with 8 contributors -- slot sharding, the per-step failure flag, a reader
    error on the LAST rank at iteration 5 that every rank must leave the loop with at the same step, the collective error-path
    checkpoint (BatchNorm statistics averaged over 8 ranks), one run directory, eight rotating per-rank state files."""
    W = 8
    h = _launch_dp8_loop(tmp_path)
    tmp_path, procs = h['tmp'], h['procs']
    outs = [p.communicate(timeout=900)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    done = [np.load(tmp_path / ('done_%d.npy' % r)) for r in range(W)]
    assert all(d[0] == 4 for d in done) and len(set(int(d[1]) for d in done)) == 1      # same step, same number of collectives
    assert all('returned normally' in o for o in outs)
    assert all('another data-parallel rank reported an error' in o for o in outs[:-1]) and 'non-finite' in outs[-1]
    ck = os.listdir(os.path.join(str(tmp_path), 'LSTMUNet', 't'))
    assert len(ck) == 1
    files = sorted(os.listdir(os.path.join(str(tmp_path), 'LSTMUNet', 't', ck[0], 'tf_ckpts')))
    for r in range(W):
        assert len([f for f in files if f.endswith('.rank%d.pt' % r)]) == 2, files


@pytest.mark.gpu
def test_bf16x3_train_loop_saved_model_and_streaming_inference(tmp_path, monkeypatch):
    """--precision bf16x3 through the drivers: train2D.train (Params.precision) on a net whose ConvLSTM layers and wide Conv2D
    units take the split route, the saved-model directory records the precision (model_params.pickle, next to the reference's
    two keys, train2D.py:236-239), Inference2D.inference runs the streaming call pattern from it (Inference2D.py:27-62) in both
    precisions."""
    import Inference2D
    import Params
    import train2D
    from PIL import Image
    with engine_backend('hip'):
        net = {'down_conv_kernels': [[(3, 32), (3, 128)], [(3, 32)]], 'lstm_kernels': [[(5, 64)], [(3, 64)]],
               'up_conv_kernels': [[(3, 128)], [(3, 16), (1, 3)]]}
        monkeypatch.setattr(Params.CTCParams, 'net_kernel_params', net)
        params = Params.CTCParams(dict(experiment_name='x3', crop_size=(32, 64), batch_size=2, unroll_len=3, num_iterations=3,
                                       validation_interval=2, print_to_console_interval=1, save_checkpoint_iteration=2,
                                       save_checkpoint_dir=str(tmp_path), save_log_dir=str(tmp_path), data_format='NCHW',
                                       learning_rate=1e-3, write_to_tb_interval=100, precision='bf16x3'))
        trainer = train2D.train(params)
        assert trainer.engine.precision == 'bf16x3' and trainer.step == 4
        save_dir = params.experiment_save_dir
        with open(os.path.join(save_dir, 'model_params.pickle'), 'rb') as f:
            meta = pickle.load(f)
        assert meta['precision'] == 'bf16x3' and meta['name'] == 'ULSTMnet2D'
        seq_dir = tmp_path / 'seq'
        seq_dir.mkdir()
        rng = np.random.default_rng(0)
        for t in range(3):
            Image.fromarray((rng.random((37, 45)) * 255).astype(np.uint8)).save(seq_dir / ('t%03d.tif' % t))
        outs, sms, seen = {}, {}, []
        real_stream = Inference2D.stream_softmax

        def spy(model, frames, *a, **k):      # which engine does the driver hand the frames to, and what comes out of it
            seen.append(model.engine.precision)
            for t, sm in real_stream(model, frames, *a, **k):
                sms.setdefault(seen[-1], []).append(sm.detach().cpu().numpy().copy())
                yield t, sm
        monkeypatch.setattr(Inference2D, 'stream_softmax', spy)
        for prec in ('bf16x3', 'fp32'):
            out_dir = tmp_path / ('out_' + prec)
            Inference2D.inference(Params.CTCInferenceParams(dict(
                model_path=save_dir, sequence_path=str(seq_dir), output_path=str(out_dir), save_intermediate=False,
                pre_sequence_frames=2, min_cell_size=1, max_cell_size=10 ** 6, data_format='NCHW', precision=prec)))
            outs[prec] = [np.asarray(Image.open(out_dir / m)) for m in sorted(os.listdir(out_dir))]
        # the driver routed each request to an engine of that precision, and the two engines agree on the softmax of the saved model
        # at the fp32 tolerance test_engine uses for the mode (this test used to print a pixel count and assert nothing)
        assert seen == ['bf16x3', 'fp32'] and len(sms['bf16x3']) == len(sms['fp32']) == 3
        sm_err = max(float(np.abs(a - b).max()) for a, b in zip(sms['bf16x3'], sms['fp32']))
        print('bf16x3 vs fp32 softmax from the saved model: max |diff| %.3e' % sm_err)
        assert sm_err <= 2e-5
        for prec in outs:
            assert len(outs[prec]) == 3 and outs[prec][0].shape == (37, 45) and outs[prec][0].dtype == np.uint16
        # (three optimiser steps from a random init leave the softmax near its tie everywhere, so instance maps of two fp32-accurate
        # engines need not agree pixel for pixel; the logits-level comparison is test_engine's and test_fullsize_gpu's)
        diff = sum(int((a != b).sum()) for a, b in zip(outs['fp32'], outs['bf16x3']))
        print('bf16x3 vs fp32 instance maps: %d of %d pixels differ' % (diff, 3 * 37 * 45))
