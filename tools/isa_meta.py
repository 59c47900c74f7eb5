"""Register / LDS / scratch figures of every kernel in a `hipcc -S --cuda-device-only` listing (the amdhsa.kernels metadata).
usage: python tools/isa_meta.py file.s [name-substring ...]"""
import re
import sys


def kernels(path):
    txt = open(path).read()
    meta = txt[txt.index('amdhsa.kernels:'):]
    out = []
    for blk in re.split(r'\n  - ', meta)[1:]:
        g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]      # noqa: E731
        out.append(dict(name=g('name'), vgpr=g('vgpr_count'), agpr=g('agpr_count'), spill=g('vgpr_spill_count'),
                        sgpr=g('sgpr_count'), lds=g('group_segment_fixed_size'), scratch=g('private_segment_fixed_size')))
    return out


if __name__ == '__main__':
    pats = sys.argv[2:]
    for k in kernels(sys.argv[1]):
        if not pats or any(p in k['name'] for p in pats):
            print('%-110s vgpr %4s agpr %3s spill %3s sgpr %3s lds %6s scratch %s' % (
                k['name'][:110], k['vgpr'], k['agpr'], k['spill'], k['sgpr'], k['lds'], k['scratch']))
