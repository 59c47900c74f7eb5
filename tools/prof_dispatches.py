"""Per-dispatch durations (in dispatch order) of the kernels whose name contains a substring -- rocprofv3 --kernel-trace run.
usage: python tools/prof_dispatches.py <dir> <substring> [first N]"""
import glob
import sqlite3
import sys


def main(src, sub, n=80):
    db = sorted(glob.glob(src + '/**/*.db', recursive=True))[0]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select start, end, grid_x / workgroup_x, name from kernels where name like ? order by start", ('%' + sub + '%',)).fetchall()
    for s, e, g, name in rows[-int(n):]:
        print('%8d blocks  %9.1f us' % (g, (e - s) / 1e3))


if __name__ == '__main__':
    main(*sys.argv[1:4])
