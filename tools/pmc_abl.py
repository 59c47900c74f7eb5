"""One table row per library variant from rocprofv3 --pmc passes of tools/wgbench.py: duration, shader clock, MFMA-busy share and the SQ
wait / instruction counters of the bf16 kernel-row weight gradient (the kernel whose name contains argv[2]).
usage: python tools/pmc_abl.py out.txt <kernel substring> label=dir[,dir...] [label=dir ...]"""
import glob
import sqlite3
import sys


def read(dirs, sub):
    agg = {}
    for d in dirs.split(','):
        dbs = sorted(glob.glob(d + '/**/*.db', recursive=True))
        if not dbs:
            continue
        cur = sqlite3.connect(dbs[0]).cursor()
        for k, c, n, v, dur in cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(end-start) from counters_collection "
                                           "group by kernel_name, counter_name"):
            if sub in k:
                e = agg.setdefault(k, {})
                e[c] = v
                e['n'], e['dur'] = n, dur
    return agg


if __name__ == '__main__':
    out, sub = sys.argv[1], sys.argv[2]
    lines = ['%-22s %-34s %5s %9s %6s %6s | %6s %6s %6s %6s | %6s %6s %6s' % ('variant', 'kernel instance', 'n', 'us/launch', 'GHz', 'MFMA%', 'WAIT', 'W_INST', 'W_LDS',
                                                                           'ISSUE', 'VALU/M', 'LDS/M', 'FIFO')]
    for arg in sys.argv[3:]:
        label, dirs = arg.split('=')
        for k, e in sorted(read(dirs, sub).items(), key=lambda kv: -kv[1]['dur'] * kv[1]['n']):
            g = lambda c: e.get(c)      # noqa: E731
            gui, wc, mf, bc = g('GRBM_GUI_ACTIVE'), g('SQ_WAVE_CYCLES'), g('SQ_INSTS_MFMA'), g('SQ_BUSY_CYCLES')
            f = lambda a, b: ('%6.3f' % (a / b)) if (a is not None and b) else '     -'      # noqa: E731
            inst = k[k.find('<'):k.find('>') + 1][:34]
            lines.append('%-22s %-34s %5d %9.1f %6s %6s | %s %s %s %s | %s %s %s' % (
                label, inst, e['n'], e['dur'] / 1e3, ('%6.3f' % (gui / 8 / e['dur'])) if gui else '     -',
                ('%6.1f' % (100 * g('SQ_VALU_MFMA_BUSY_CYCLES') / (gui / 8 * 1024))) if gui and g('SQ_VALU_MFMA_BUSY_CYCLES') is not None else '     -',
                f(g('SQ_WAIT_ANY'), wc), f(g('SQ_WAIT_INST_ANY'), wc), f(g('SQ_WAIT_INST_LDS'), wc), f(g('SQ_ACTIVE_INST_ANY'), wc),
                f(g('SQ_INSTS_VALU'), mf), f(g('SQ_INSTS_LDS'), mf), f(g('SQ_LDS_DATA_FIFO_FULL'), bc)))
    txt = '\n'.join(lines)
    print(txt)
    with open(out, 'a') as fh:
        fh.write(txt + '\n')
