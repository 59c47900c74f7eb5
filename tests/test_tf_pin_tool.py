"""tools/tf_pin.py has never met a TensorFlow (none can be installed here).  This test runs it end to end against a STAND-IN
`tensorflow` module (tests/fake_tensorflow.py: every layer returns the oracle's own output -- it pins nothing) and then feeds
the fixtures it wrote to the consumers in tests/test_tf_pinned.py: the script's control flow and the fixture schema are
proven, so that on a machine with TensorFlow 2.x the pin is one command.  Fixtures go to a temporary directory; the status
of the real tree stays "parity unpinned at the TensorFlow boundary"."""
import importlib.util
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_tf_pin_runs_end_to_end_on_a_stand_in_tensorflow(tmp_path, monkeypatch):
    import fake_tensorflow
    assert 'tensorflow' not in sys.modules or getattr(sys.modules['tensorflow'], '__version__', '').endswith('standin')
    names = fake_tensorflow.install()
    try:
        tool = _load(os.path.join(ROOT, 'tools', 'tf_pin.py'), 'tf_pin_tool')
        monkeypatch.setattr(tool, 'GOLDEN', str(tmp_path))
        assert tool.main() == 0
    finally:
        for n in names:
            sys.modules.pop(n, None)
    written = sorted(os.listdir(tmp_path))
    assert written == ['tf_bn_lrelu.npz', 'tf_conv2d.npz', 'tf_convlstm_k3.npz', 'tf_convlstm_k5.npz', 'tf_loss_adam.npz',
                       'tf_pin.json', 'tf_resize_pad.npz']
    info = json.load(open(tmp_path / 'tf_pin.json'))
    assert info['tensorflow'].endswith('standin') and info['resize_images_bilinear'] == 'tf2.0'
    assert info['bn_moving_variance_rule'] == 'unbiased'
    assert info['tf_bundle_reads_tf_checkpoint'] and info['tf_reads_tf_bundle_checkpoint'] and info['attribute_path_keys']
    # the consumers read what the tool wrote: every test of test_tf_pinned.py passes on these fixtures (and must, the stand-in
    # being the oracle itself) -- schema and tolerances are exercised, nothing is pinned
    pinned = _load(os.path.join(ROOT, 'tests', 'test_tf_pinned.py'), 'tf_pinned_consumers')
    monkeypatch.setattr(pinned, 'GOLDEN', str(tmp_path))
    assert pinned.parity_status().startswith('pinned (TensorFlow 2.0.0-standin')
    for name in ('convlstm_k5', 'convlstm_k3'):
        pinned.test_convlstm_against_tensorflow(name)
    pinned.test_conv2d_same_against_tensorflow()
    pinned.test_batchnorm_lrelu_against_tensorflow()
    pinned.test_resize_and_reflect_pad_against_tensorflow()
    pinned.test_loss_and_adam_against_tensorflow()
    # ... and the real tree is untouched
    assert not os.path.exists(os.path.join(ROOT, 'tests', 'golden', 'tf_pin.json'))
    f = np.load(tmp_path / 'tf_loss_adam.npz')
    assert f['dlogits'].shape == f['logits'].shape and np.isfinite(f['dlogits']).all()
