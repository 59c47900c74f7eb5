"""Build the gfx950 kernel library in-tree with hipcc (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), 'csrc')
SOURCES = ['lu_conv.hip', 'lu_wgrad.hip', 'lu_pointwise.hip', 'lu_postprocess.hip']
LIB = os.path.join(CSRC, 'liblstmunet_hip.so')


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, 'lu_device.h'),
            os.path.join(os.path.dirname(os.path.dirname(HERE)), 'include', 'lstm_unet_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-x', 'hip'] + \
          [os.path.join(CSRC, s) for s in SOURCES] + ['-o', LIB]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def build_ablation(verbose=True):
    """TOOLS ONLY (tools/kbench.py): the same sources with -DLU_ABLATION, which lets lu_conv_desc.flags >> 16 switch parts of
    the fragment kernel's loop off (weight loads, LDS reads, halo prefetch, epilogue, barrier) to see what bounds it.  A
    separate file: the product library never contains these switches."""
    out = os.path.join(CSRC, 'liblstmunet_abl.so')
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-DLU_ABLATION', '-x', 'hip'] + \
          [os.path.join(CSRC, s) for s in SOURCES] + ['-o', out]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


if __name__ == '__main__':
    if '--ablation' in sys.argv:
        build_ablation()
    else:
        build(force='--force' in sys.argv)
