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
                learning_rate=1e-3, write_to_tb_interval=2)
    params = Params.CTCParams(args)
    assert isinstance(params.train_data_provider, DataHandeling.SyntheticSequence2D) and params.channel_axis == 1
    trainer = train2D.train(params)
    assert trainer.step == 4          # range(step, num_iterations + 1), as the reference loop (train2D.py:145)
    save_dir = params.experiment_save_dir
    assert os.path.exists(os.path.join(save_dir, 'model.ckpt'))
    with open(os.path.join(save_dir, 'model_params.pickle'), 'rb') as f:
        meta = pickle.load(f)
    assert meta['name'] == 'ULSTMnet2D' and meta['params'][0] == net
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
