"""MFMA utilisation and shader clock per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass.
GRBM_GUI_ACTIVE is reported summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES is summed over all SIMDs, so
    clock = GUI_ACTIVE / 8 / duration,   MFMA utilisation = BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs).
usage: python tools/pmc_mfma.py gpurun_out/pmc_mfma_fp32 gpurun_out/pmc_mfma_bf16 [gpurun_out/pmc_mfma_bf16x3] profiles/r01_pmc_mfma_util.json"""
import glob
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'lstm-unet_amd'))


def _build_id():
    from lu_native import build
    return build.build_id()


def summarize(path):
    dbs = sorted(glob.glob(path + '/**/*.db', recursive=True))
    if not dbs:          # (PMC_MODES of tools/gpu/pmc_traffic.sh: a pass that was not run)
        return {}
    cur = sqlite3.connect(dbs[0]).cursor()
    agg = {}
    for k, c, n, v, d in cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(end-start) from "
                                     "counters_collection group by kernel_name, counter_name"):
        agg.setdefault(k, {})[c] = (n, v, d)
    out = {}
    for k, v in agg.items():
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in v and 'GRBM_GUI_ACTIVE' in v and v['GRBM_GUI_ACTIVE'][1] > 0:
            busy, (n, gui, dur) = v['SQ_VALU_MFMA_BUSY_CYCLES'][1], v['GRBM_GUI_ACTIVE']
            if busy <= 0:
                continue
            out[k] = {'launches': n, 'avg_duration_us': round(dur / 1e3, 1), 'shader_clock_ghz': round(gui / 8 / dur, 3),
                      'mfma_utilisation': round(busy / (gui / 8 * 1024), 4)}
    return dict(sorted(out.items(), key=lambda kv: -kv[1]['launches'] * kv[1]['avg_duration_us']))


if __name__ == '__main__':
    res = {'build_id': _build_id(), 'note': 'rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE of bench.py --steps 1 --warmup 1 '
                   '(config-2); utilisation = MFMA-busy cycles / SIMD cycles available at the measured clock',
           'fp32': summarize(sys.argv[1]), 'bf16': summarize(sys.argv[2])}
    out = sys.argv[3]
    if len(sys.argv) > 4:      # (round 5: a third pass, precision 'bf16x3':  <fp32 dir> <bf16 dir> <bf16x3 dir> <out>)
        res['bf16x3'] = summarize(sys.argv[3])
        out = sys.argv[4]
    with open(out, 'w') as fh:
        json.dump(res, fh, indent=1)
    for mode in [m_ for m_ in ('fp32', 'bf16', 'bf16x3') if m_ in res]:
        for k, v in list(res[mode].items())[:6]:
            print(mode, '%-78s util %5.1f %%  clock %.2f GHz  %8.1f us x %d' % (k[:78], 100 * v['mfma_utilisation'], v['shader_clock_ghz'],
                                                                                  v['avg_duration_us'], v['launches']))
