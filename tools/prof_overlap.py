"""Kernel-trace timeline of a rocprofv3 run: busy time (union of kernel intervals), summed kernel time and span of the last
`frac` of the trace -- how much of the work overlapped across streams.  usage: python tools/prof_overlap.py <dir> [frac]"""
import glob
import sqlite3
import sys


def main(src, frac=0.5):
    db = sorted(glob.glob(src + '/**/*.db', recursive=True))[0]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select start, end, name from kernels order by start").fetchall()
    rows = rows[int(len(rows) * (1.0 - float(frac))):]
    total = sum(e - s for s, e, _ in rows)
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    for s, e, _ in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    span = max(e for _, e, _ in rows) - rows[0][0]
    print('kernels %d  span %.2f ms  busy (union) %.2f ms  summed kernel time %.2f ms  overlapped %.2f ms' %
          (len(rows), span / 1e6, busy / 1e6, total / 1e6, (total - busy) / 1e6))


if __name__ == '__main__':
    main(*sys.argv[1:3])
