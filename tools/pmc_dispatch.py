"""Per-DISPATCH counter values of the kernels whose name contains a substring (rocprofv3 --pmc run, rocpd sqlite), in
dispatch order -- separates the launches of one kernel name (e.g. the five 5x5 weight gradients of a step).
usage: python tools/pmc_dispatch.py <pmc dir> <COUNTER> <name substring> [out.json]"""
import glob
import json
import sqlite3
import sys


def main(path, counter, sub, dst=None):
    db = sorted(glob.glob(path + '/**/*.db', recursive=True))[0]
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    print('columns:', cols)
    want = [c for c in ('dispatch_id', 'grid_size', 'grid_size_x', 'workgroup_size', 'workgroup_size_x', 'kernel_name', 'value') if c in cols]
    order = 'dispatch_id' if 'dispatch_id' in cols else 'rowid'
    rows = cur.execute("select %s from counters_collection where counter_name=? and kernel_name like ? order by %s" %
                       (', '.join(want), order), (counter, '%' + sub + '%')).fetchall()
    out = [dict(zip(want, r)) for r in rows]
    for r in out:
        r['kernel_name'] = r['kernel_name'][:120]
        print({k: (v if k != 'kernel_name' else v[28:100]) for k, v in r.items()})
    if dst:
        json.dump(out, open(dst, 'w'), indent=1)


if __name__ == '__main__':
    main(*sys.argv[1:5])
