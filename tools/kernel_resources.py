"""Register / scratch / LDS usage per kernel from a hipcc -Rpass-analysis=kernel-resource-usage log.
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -c -x hip FILE.hip -o /tmp/x.o -Rpass-analysis=kernel-resource-usage 2> log
       python tools/kernel_resources.py log [substring ...]"""
import re
import subprocess
import sys


def main(path, *subs):
    txt = open(path).read()
    for b in re.split(r'remark: Function Name: ', txt)[1:]:
        name = b.split(' ')[0].strip()
        dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace('(anonymous namespace)::', '').replace('void ', '')
        dem = re.sub(r'\(.*$', '', dem)
        if subs and not any(s in dem for s in subs):
            continue

        def g(k):
            m = re.search(k + r': (\d+)', b)
            return int(m.group(1)) if m else -1
        print('%-64s VGPR %3d AGPR %3d vspill %3d scratch %4d occ %d LDS %6d' % (
            dem[:64], g('    VGPRs'), g('AGPRs'), g('VGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]'),
            g(r'LDS Size \[bytes/block\]')))


if __name__ == '__main__':
    main(*sys.argv[1:])
