"""Extract per-kernel FETCH_SIZE / WRITE_SIZE averages from rocprofv3 --pmc runs (rocpd sqlite) into
profiles/<name>.json.  Units: the counters are in KiB; on gfx950 FETCH_SIZE reports HALF of the bytes of a
wide coalesced read (MI355X_MICROARCH.md §HBM), so reads are doubled; WRITE_SIZE is used as reported.
usage: python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE profiles/r01_pmc_traffic.json"""
import glob
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'lstm-unet_amd'))


def _build_id():
    from lu_native import build
    return build.build_id()


def per_kernel(path, counter):
    db = sorted(glob.glob(path + '/**/*.db', recursive=True))[0]
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, n, avg in cur.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? "
                                    "group by kernel_name", (counter,)):
        out[name] = (n, avg)
    return out


def main(fetch_dir, write_dir, dst):
    f, w = per_kernel(fetch_dir, 'FETCH_SIZE'), per_kernel(write_dir, 'WRITE_SIZE')
    res = {}
    for k in sorted(set(f) | set(w)):
        fk, wk = f.get(k, (0, 0.0)), w.get(k, (0, 0.0))
        res[k] = {'launches': fk[0] or wk[0], 'fetch_bytes_per_launch_corrected': 2.0 * fk[1] * 1024.0,
                  'fetch_size_raw_kib': fk[1], 'write_bytes_per_launch': wk[1] * 1024.0,
                  'traffic_bytes_per_launch': 2.0 * fk[1] * 1024.0 + wk[1] * 1024.0}
    with open(dst, 'w') as fh:
        json.dump({'build_id': _build_id(), 'note': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of '
                           'bench.py --steps 1 --warmup 1; FETCH_SIZE doubled per the gfx950 correction; L2 fabric-side '
                           'bytes (Infinity-Cache hits included)', 'kernels': res}, fh, indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]['traffic_bytes_per_launch'] * kv[1]['launches'])[:6]:
        print('%-90s launches %4d  traffic/launch %.1f MB' % (k[:90], v['launches'], v['traffic_bytes_per_launch'] / 1e6))


if __name__ == '__main__':
    main(*sys.argv[1:4])
