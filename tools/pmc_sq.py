"""Per-kernel averages of arbitrary SQ counters from several rocprofv3 --pmc passes of the same command (one directory per pass), as
ratios that say where a kernel's wave-cycles go: issue-side counters per SQ_WAVE_CYCLES / SQ_BUSY_CYCLES, instruction mix per MFMA.
usage: python tools/pmc_sq.py out.json label=dir [label=dir ...]"""
import glob
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'lstm-unet_amd'))


def read_pass(path, agg):
    db = sorted(glob.glob(path + '/**/*.db', recursive=True))[0]
    cur = sqlite3.connect(db).cursor()
    for k, c, n, v, d in cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(end-start) from "
                                     "counters_collection group by kernel_name, counter_name"):
        e = agg.setdefault(k, {'launches': n, 'avg_duration_us': round(d / 1e3, 1)})
        e[c] = v


def ratios(e):
    g = lambda k: e.get(k)      # noqa: E731
    out = {}
    wc, mf = g('SQ_WAVE_CYCLES'), g('SQ_INSTS_MFMA')
    if wc:
        for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_SCA',
                  'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_MISC'):
            if g(k) is not None:
                out[k + '/WAVE_CYCLES'] = round(g(k) / wc, 4)
    if mf:
        for k in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM', 'SQ_INSTS_SMEM', 'SQ_INSTS_BRANCH'):
            if g(k) is not None:
                out[k + '/MFMA'] = round(g(k) / mf, 3)
    if g('SQ_BUSY_CYCLES'):
        for k in ('SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_DATA_FIFO_FULL', 'SQ_LDS_CMD_FIFO_FULL', 'SQ_LDS_ADDR_CONFLICT'):
            if g(k) is not None:
                out[k + '/BUSY_CYCLES'] = round(g(k) / g('SQ_BUSY_CYCLES'), 4)
    return out


if __name__ == '__main__':
    from lu_native import build
    res = {'build_id': build.build_id(), 'note': 'rocprofv3 --kernel-trace --pmc <counters> passes of bench.py --steps 1 --warmup 1 (config-2), averages '
           'per launch summed over the chip; */WAVE_CYCLES: share of resident-wave cycles, */MFMA: instructions per MFMA instruction (VALU '
           'includes the MFMAs), SQ_LDS_*/BUSY_CYCLES: per SQ-busy cycle'}
    for arg in sys.argv[2:]:
        label, dirs = arg.split('=')
        agg = {}
        for d in dirs.split(','):
            read_pass(d, agg)
        rows = sorted(agg.items(), key=lambda kv: -kv[1]['launches'] * kv[1]['avg_duration_us'])[:10]
        res[label] = {k: dict(launches=e['launches'], avg_duration_us=e['avg_duration_us'], ratios=ratios(e),
                              raw={c: round(v, 1) for c, v in e.items() if c.startswith('SQ_')}) for k, e in rows}
        for k, e in rows[:5]:
            print(label, k[:70], e['avg_duration_us'], json.dumps(ratios(e)))
    with open(sys.argv[1], 'w') as fh:
        json.dump(res, fh, indent=1)
