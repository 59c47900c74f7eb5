"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite output) into CSV/markdown for profiles/.
usage: python tools/prof_summary.py gpurun_out/prof2 profiles/r01_bench_kernel_stats"""
import glob
import sqlite3
import sys


def main(src, dst, ngrid=40):
    db = sorted(glob.glob(src + '/**/*.db', recursive=True))[0]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(dst + '.csv', 'w') as f:
        f.write('Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage\n')
        for n, c, t, a, lo, hi in rows:
            f.write('"%s",%d,%d,%.1f,%d,%d,%.4f\n' % (n, c, t, a, lo, hi, 100.0 * t / total))
    grid = cur.execute("select name, grid_x/workgroup_x, grid_y, grid_z, count(*), avg(end-start), sum(end-start) from kernels "
                       "group by name, grid_x, grid_y, grid_z order by sum(end-start) desc limit %d" % int(ngrid)).fetchall()
    with open(dst + '_by_grid.csv', 'w') as f:
        f.write('Name,BlocksX,BlocksY,BlocksZ,Calls,AverageNs,TotalNs\n')
        for r in grid:
            f.write('"%s",%d,%d,%d,%d,%.1f,%d\n' % r)
    print('wrote', dst + '.csv', 'total kernel time %.1f ms' % (total / 1e6))
    for n, c, t, a, lo, hi in rows[:12]:
        print('%6.2f%%  %6d calls  avg %10.1f us  %s' % (100.0 * t / total, c, a / 1e3, n[:110]))


if __name__ == '__main__':
    main(*sys.argv[1:4])
