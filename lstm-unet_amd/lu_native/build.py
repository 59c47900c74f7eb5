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


def build_ablation(bits, verbose=True):
    """TOOLS ONLY (tools/kbench.py): lu_conv.hip with -DLU_ABLATION=<bits>, which compiles parts of the fragment kernel's loop
    out (1 weight loads, 2 LDS fragment reads, 4 halo prefetch, 8 epilogue, 16 chunk barrier) to see what bounds it.
    Separate files (liblstmunet_abl<bits>.so): the product library never contains these switches."""
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    base = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, ('abl%d_' % bits if src == 'lu_conv.hip' else 'abl_') + src.replace('.hip', '.o'))
        if src == 'lu_conv.hip' or not os.path.exists(obj) or os.path.getmtime(obj) < os.path.getmtime(os.path.join(CSRC, src)):
            cmd = base + (['-DLU_ABLATION=%d' % bits] if src == 'lu_conv.hip' else []) + ['-c', '-x', 'hip', os.path.join(CSRC, src), '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    out = os.path.join(CSRC, 'liblstmunet_abl%d.so' % bits)
    subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out])
    return out


if __name__ == '__main__':
    if '--ablation' in sys.argv:
        for b_ in sys.argv[sys.argv.index('--ablation') + 1:]:
            print(build_ablation(int(b_)))
    else:
        build(force='--force' in sys.argv)
