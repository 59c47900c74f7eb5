"""Where does a step's time go that no kernel accounts for?  Reads a rocprofv3 --kernel-trace (rocpd sqlite) run and prints,
for the kernels between two marks (default: the last `--tail` fraction of the trace), busy time (union of intervals over all
streams), span, the idle gaps by the kernel that FOLLOWS them, the largest single gaps, and time by kernel name -- including the
bandwidth-bound helpers bench.py's per-class tables do not list.
usage: python tools/trace_gaps.py gpurun_out/prof_c4 [--tail 0.5] [--out profiles/r05_c4_gaps.json]"""
import argparse
import glob
import json
import re
import sqlite3


def short(name):
    """rocprof kernel name -> 'conv_halo_kernel<5, 1, true>' (no return type, namespace or argument list)."""
    n = re.sub(r'^void ', '', name or '?').replace('(anonymous namespace)::', '')
    depth = 0
    for i, ch in enumerate(n):
        if ch == '<':
            depth += 1
        elif ch == '>':
            depth -= 1
        elif ch == '(' and depth == 0:
            return n[:i][:90]
    return n[:90]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('src')
    ap.add_argument('--tail', type=float, default=1.0, help='analyse the last fraction of the launches (1 = everything)')
    ap.add_argument('--min-gap-us', type=float, default=20.0)
    ap.add_argument('--out', default=None)
    ap.add_argument('--step-marker', default=None, help='kernel that ends a step (e.g. adam_kernel): print per-step span / busy / '
                    'time by kernel, so that a slow step can be compared with a fast one')
    a = ap.parse_args()
    db = sorted(glob.glob(a.src + '/**/*.db', recursive=True))[0]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute('select start, end, name from kernels order by start').fetchall()
    if a.step_marker:
        steps, cur_rows = [], []
        for r in rows:
            cur_rows.append(r)
            if a.step_marker in r[2]:
                steps.append(cur_rows)
                cur_rows = []
        per_step = []
        for i, st in enumerate(steps):
            span_ = st[-1][1] - st[0][0]
            busy_ = sum(e - s_ for s_, e, _ in st)
            names = {}
            for s_, e, n in st:
                v = names.setdefault(short(n), [0, 0])
                v[0] += e - s_
                v[1] += 1
            per_step.append({'launches': len(st), 'span_ms': span_ / 1e6, 'kernel_sum_ms': busy_ / 1e6,
                             'by_kernel_ms': {k: [t / 1e6, c] for k, (t, c) in sorted(names.items(), key=lambda kv: -kv[1][0])[:25]}})
            print('step %d: %d launches, span %.1f ms, sum of kernel durations %.1f ms' % (i, len(st), span_ / 1e6, busy_ / 1e6))
        if len(per_step) >= 2:
            a_, b_ = per_step[0]['by_kernel_ms'], per_step[-1]['by_kernel_ms']
            print('first vs last step, by kernel (ms):')
            for k in list(b_)[:14]:
                print('   %9.1f -> %9.1f   %s' % (a_.get(k, [0])[0], b_[k][0], k))
        if a.out:
            json.dump(per_step, open(a.out.replace('.json', '_per_step.json'), 'w'), indent=1)
    rows = rows[int(len(rows) * (1.0 - a.tail)):]
    span = rows[-1][1] - rows[0][0] if rows else 0
    busy, cur_end, gaps = 0, rows[0][0], []
    prev = None
    for s, e, n in rows:
        if s > cur_end:
            gaps.append((s - cur_end, prev, n))
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
        prev = n
    by_next, by_name = {}, {}
    for g, p, n in gaps:
        k = short(n)
        v = by_next.setdefault(k, [0, 0])
        v[0] += g
        v[1] += 1
    for s, e, n in rows:
        k = short(n)
        v = by_name.setdefault(k, [0, 0])
        v[0] += e - s
        v[1] += 1
    idle = span - busy
    print('launches %d, span %.1f ms, busy %.1f ms (%.2f %%), idle %.1f ms in %d gaps (%d above %.0f us: %.1f ms)' %
          (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / max(span, 1), idle / 1e6, len(gaps),
           sum(1 for g in gaps if g[0] >= a.min_gap_us * 1e3), a.min_gap_us, sum(g[0] for g in gaps if g[0] >= a.min_gap_us * 1e3) / 1e6))
    print('idle time by the kernel that follows the gap:')
    for k, (t, c) in sorted(by_next.items(), key=lambda kv: -kv[1][0])[:18]:
        print('  %9.2f ms  %6d gaps  avg %8.1f us  before %s' % (t / 1e6, c, t / c / 1e3, k))
    print('largest gaps:')
    for g, p, n in sorted(gaps, reverse=True)[:12]:
        print('  %9.2f ms  after %s  before %s' % (g / 1e6, short(p)[:60], short(n)[:60]))
    print('time by kernel:')
    for k, (t, c) in sorted(by_name.items(), key=lambda kv: -kv[1][0])[:30]:
        print('  %9.2f ms  %5.2f %%  %6d launches  avg %9.1f us  %s' % (t / 1e6, 100.0 * t / max(span, 1), c, t / c / 1e3, k))
    if a.out:
        json.dump({'launches': len(rows), 'span_ms': span / 1e6, 'busy_ms': busy / 1e6, 'idle_ms': idle / 1e6,
                   'idle_by_next_kernel_ms': {k: [t / 1e6, c] for k, (t, c) in sorted(by_next.items(), key=lambda kv: -kv[1][0])[:40]},
                   'largest_gaps_ms': [[g / 1e6, (p or '?')[:80], n[:80]] for g, p, n in sorted(gaps, reverse=True)[:40]],
                   'time_by_kernel_ms': {k: [t / 1e6, c] for k, (t, c) in sorted(by_name.items(), key=lambda kv: -kv[1][0])}},
                  open(a.out, 'w'), indent=1)


if __name__ == '__main__':
    main()
