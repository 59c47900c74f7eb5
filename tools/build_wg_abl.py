"""TOOLS ONLY: ablation builds of the bf16 kernel-row weight gradient (lu_wgrad.hip with -DLU_WG_ABL=<bits>: 1 global loads, 2 LDS
stores, 4 stage barrier, 8 bias sums, 16 LDS fragment reads compiled OUT of the 64-pixel-stage loop) as whole libraries under
abl_tmp/ (git-ignored; travels with the gpurun snapshot, never part of the product).  Cross-compiled here so that no GPU minute is
spent in hipcc.   usage: python tools/build_wg_abl.py 1 2 3 16 ..."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'lstm-unet_amd', 'csrc')
OUT = os.path.join(ROOT, 'abl_tmp')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC']
HIPCC = '/opt/rocm/bin/hipcc'


def cc(src, obj, extra=()):
    subprocess.check_call([HIPCC] + FLAGS + list(extra) + ['-c', '-x', 'hip', os.path.join(CSRC, src), '-o', obj])
    return obj


def main(bits_list, extra=()):
    os.makedirs(OUT, exist_ok=True)
    jobs = [('lu_conv.hip', os.path.join(OUT, 'lu_conv.o'), ()), ('lu_pointwise.hip', os.path.join(OUT, 'lu_pointwise.o'), ()),
            ('lu_postprocess.hip', os.path.join(OUT, 'lu_postprocess.o'), ())]
    jobs = [j for j in jobs if not os.path.exists(j[1]) or os.path.getmtime(j[1]) < os.path.getmtime(os.path.join(CSRC, j[0]))]
    jobs += [('lu_wgrad.hip', os.path.join(OUT, 'wg%d.o' % b), tuple(extra) + ('-DLU_WG_ABL=%d' % b,)) for b in bits_list]
    with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(lambda j: cc(*j), jobs))
    for b in bits_list:
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', os.path.join(OUT, 'lu_conv.o'), os.path.join(OUT, 'lu_pointwise.o'),
                               os.path.join(OUT, 'lu_postprocess.o'), os.path.join(OUT, 'wg%d.o' % b), '-o', os.path.join(OUT, 'libwg%d.so' % b)])
        print(os.path.join(OUT, 'libwg%d.so' % b))
    for f in os.listdir(OUT):
        if f.startswith('wg') and f.endswith('.o'):
            os.remove(os.path.join(OUT, f))


if __name__ == '__main__':
    main([int(a) for a in sys.argv[1:] if not a.startswith('-')], [a for a in sys.argv[1:] if a.startswith('-')])
