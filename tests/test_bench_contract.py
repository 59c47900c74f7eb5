"""The FLOP accounting bench.py reports against, pinned to SURVEY §8d / BASELINE.md §2."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_flops_match_survey():
    import Params
    b = _bench()
    net = Params.CTCParams.net_kernel_params
    total, fwd = b.step_flops(net, 256, 256, 4, 8)
    assert abs(fwd / 1e9 - 778.4) < 0.1                      # forward GFLOP/frame at config-2
    assert abs(total / 32 / 1e9 - 2266.6) < 0.2              # training GFLOP/frame
    assert abs(total / 1e12 - 72.53) < 0.01                  # TFLOP per step per GPU
    total4, fwd4 = b.step_flops(net, 832, 992, 2, 16)
    assert abs(fwd4 / 1e9 - 9803.6) < 1.0 and abs(total4 / 1e12 - 926.9) < 0.2     # config-4
    per_launch = b.lstm_step_flops(net, 256, 256, 4)
    assert [round(x / 1e9) for x in per_launch] == [866, 1288, 429, 322]
    assert b.PEAK_FP32_MFMA_TFLOPS == 157.3
