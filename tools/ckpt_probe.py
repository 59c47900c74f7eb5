"""Dump the key list of a TensorFlow checkpoint (`model.ckpt.index` + `.data-*`) and diff it against the variable names / shapes
`tf_bundle.load_model_weights` expects for a given network -- for the day a pretrained LSTM-UNet model appears (reference
README.md:95-97; train2D.py:235 writes `model.ckpt`, Inference2D.py:34 loads it).  No TensorFlow, no GPU: the index is an
SSTable of BundleEntryProto records, read by lstm-unet_amd/tf_bundle.py.

    python tools/ckpt_probe.py /path/to/model.ckpt [--params /path/to/model_params.pickle] [--json out.json]

Exit code 0: every model variable was found (directly, through the checkpoint's object graph, or through the
`layer_with_weights-N` aliases) with the right shape; 1: something is missing or mis-shaped (the report says what)."""
import argparse
import json
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'lstm-unet_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np      # noqa: E402


def expected_variables(net_params, in_channels=1):
    """{object-graph path: shape} of every variable of ULSTMnet2D(net_params) -- from the static plan, no device needed."""
    import tf_bundle as tb
    from lu_native.plan import make_plan, param_specs, bn_stat_specs

    class _Names(object):      # what tf_bundle.checkpoint_names reads of an engine
        pass
    plan = make_plan(net_params, in_channels)
    shapes = {n: tuple(s) for n, s, _ in param_specs(plan)}
    stats = {n: tuple(s) for n, s, _ in bn_stat_specs(plan)}
    e = _Names()
    e.P, e.S = shapes, stats
    names = tb.checkpoint_names(e)
    return {names[k]: (shapes.get(k) or stats[k]) for k in names}, plan


def probe(prefix, net_params, in_channels=1):
    import re
    import tf_bundle as tb
    listed = tb.list_bundle(prefix)
    want, plan = expected_variables(net_params, in_channels)
    n_down = len(plan['down'])
    have, how = {}, {}
    try:
        for path, key in tb.resolve_through_object_graph(prefix, want.keys()).items():
            if key in listed:
                have[path], how[path] = key, 'object graph'
    except Exception as exc:      # noqa: BLE001 -- a probe reports, it does not stop
        print('object graph not usable: %s: %s' % (type(exc).__name__, exc))
    alias = re.compile(r'^layer_with_weights-(\d+)/')
    for key in listed:
        if not key.endswith(tb.SUFFIX):
            continue
        path = key[:-len(tb.SUFFIX)]
        m = alias.match(path)
        kind = 'direct key'
        if m:
            j = int(m.group(1))
            path = ('DownLayers/%d/' % j if j < n_down else 'UpLayers/%d/' % (j - n_down)) + path[m.end():]
            kind = 'layer_with_weights alias'
        if path not in have:
            have[path], how[path] = key, kind
    report = {'prefix': prefix, 'entries': len(listed), 'model_variables': len(want), 'found': 0, 'missing': [], 'shape_mismatch': [],
              'found_by': {}, 'other_entries': []}
    used = set()
    for path, shape in sorted(want.items()):
        key = have.get(path)
        if key is None:
            report['missing'].append(path)
            continue
        used.add(key)
        got = tuple(listed[key]['shape'])
        if got != tuple(shape):
            report['shape_mismatch'].append({'variable': path, 'key': key, 'checkpoint': got, 'model': tuple(shape)})
            continue
        report['found'] += 1
        report['found_by'][how[path]] = report['found_by'].get(how[path], 0) + 1
    for key in sorted(listed):
        if key not in used:
            e = listed[key]
            report['other_entries'].append({'key': key, 'dtype': int(e['dtype']), 'shape': list(e['shape'])})
    return report, listed


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n\n')[0])
    ap.add_argument('prefix', help='checkpoint prefix, e.g. .../model.ckpt (the .index / .data-* files beside it)')
    ap.add_argument('--params', default=None, help="model_params.pickle written beside the checkpoint (train2D.py:236-239): its "
                    "'params' tuple holds net_kernel_params; default: Params.CTCParams.net_kernel_params")
    ap.add_argument('--in-channels', type=int, default=1)
    ap.add_argument('--json', default=None)
    ap.add_argument('--list', action='store_true', help='print every key with dtype code and shape')
    a = ap.parse_args(argv)
    prefix = a.prefix[:-len('.index')] if a.prefix.endswith('.index') else a.prefix
    if a.params:
        with open(a.params, 'rb') as fh:
            net = pickle.load(fh)['params'][0]
    else:
        import Params
        net = Params.CTCParams.net_kernel_params
    report, listed = probe(prefix, net, a.in_channels)
    if a.list:
        for key in sorted(listed):
            print('%-90s dtype %2d shape %s' % (key, listed[key]['dtype'], list(listed[key]['shape'])))
    print('%s: %d entries; model variables %d: found %d %s, missing %d, shape mismatches %d, other entries %d' % (
        prefix, report['entries'], report['model_variables'], report['found'], report['found_by'], len(report['missing']),
        len(report['shape_mismatch']), len(report['other_entries'])))
    for pth in report['missing'][:12]:
        print('   missing: %s' % pth)
    for m in report['shape_mismatch'][:12]:
        print('   shape:   %(variable)s  checkpoint %(checkpoint)s  model %(model)s  (%(key)s)' % m)
    for o in report['other_entries'][:12]:
        print('   other:   %(key)s  dtype %(dtype)d  shape %(shape)s' % o)
    if a.json:
        with open(a.json, 'w') as fh:
            json.dump(report, fh, indent=1, default=lambda x: list(x) if isinstance(x, (tuple, np.ndarray)) else str(x))
    return 0 if not report['missing'] and not report['shape_mismatch'] else 1


if __name__ == '__main__':
    sys.exit(main())
