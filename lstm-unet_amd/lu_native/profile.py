"""Per-kernel-class timing of one step from HIP events on the launch stream (ops.EVENT_LOG) against the gfx950 rooflines --
shared by bench.py's `roofline` block and train2D's --profile (the reference's --profile wraps train_step in
tf.summary.trace_on / trace_export, train2D.py:152-160; here the step is bracketed with HIP events per kernel class, and the
hardware counters -- HBM bytes, MFMA busy -- come from running the same command under rocprofv3, see profiles/README.md)."""
PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
PEAK_HBM_GBS = 8000.0             # same guide: HBM3E spec (about 6.3 TB/s achievable)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide: dense bf16 (v_mfma_f32_32x32x16_bf16), no sparsity


def summarize_events(events):
    """events: ops.EVENT_LOG entries (kind, work, start, end) after a synchronize.  -> (mfma rows, hbm rows), each sorted by
    time; work is FLOPs for the MFMA classes and bytes for the 'hbm:' classes."""
    classes = {}
    for kind, work, e0, e1 in events:
        c = classes.setdefault(kind, {'work': 0.0, 'ms': 0.0, 'n': 0})
        c['work'] += work
        c['ms'] += e0.elapsed_time(e1)
        c['n'] += 1
    rows, hbm_rows = [], []
    for kind, c in classes.items():
        if c['ms'] <= 0:
            continue
        if kind.startswith('hbm:'):
            gbs = c['work'] / (c['ms'] * 1e-3) / 1e9
            hbm_rows.append({'kernel': kind[4:], 'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                             'frac': round(gbs / PEAK_HBM_GBS, 4), 'launches_per_step': c['n'],
                             'avg_launch_ms': round(c['ms'] / c['n'], 4), 'ms_per_step': round(c['ms'], 2)})
            continue
        ach = c['work'] / (c['ms'] * 1e-3) / 1e12
        peak = PEAK_BF16_MFMA_TFLOPS if 'bf16' in kind else PEAK_FP32_MFMA_TFLOPS
        rows.append({'kernel': kind, 'bound': 'mfma', 'achieved': round(ach, 2), 'peak': peak, 'unit': 'TFLOP/s',
                     'frac': round(ach / peak, 4), 'launches_per_step': c['n'], 'avg_launch_ms': round(c['ms'] / c['n'], 4),
                     'ms_per_step': round(c['ms'], 2), 'flops_per_launch_avg': c['work'] / c['n']})
    rows.sort(key=lambda r_: -r_['ms_per_step'])
    hbm_rows.sort(key=lambda r_: -r_['ms_per_step'])
    return rows, hbm_rows


def by_shape(events):
    """The same events grouped by (kernel class, algorithmic work per launch) -- one row per layer shape, so that the levels of
    the U-Net (and the tile-starved coarse ones among them) can be read apart inside a class: bench.py --by-shape."""
    shapes = {}
    for kind, work, e0, e1 in events:
        c = shapes.setdefault((kind, float(work)), {'ms': 0.0, 'n': 0})
        c['ms'] += e0.elapsed_time(e1)
        c['n'] += 1
    rows = []
    for (kind, work), c in shapes.items():
        if c['ms'] <= 0:
            continue
        hbm = kind.startswith('hbm:')
        rate = work * c['n'] / (c['ms'] * 1e-3) / (1e9 if hbm else 1e12)
        peak = PEAK_HBM_GBS if hbm else PEAK_BF16_MFMA_TFLOPS if 'bf16' in kind else PEAK_FP32_MFMA_TFLOPS
        rows.append({'kernel': kind.split(' ')[0], 'work_per_launch': work, 'launches': c['n'], 'avg_launch_ms': round(c['ms'] / c['n'], 4),
                     'ms_per_step': round(c['ms'], 3), 'achieved': round(rate, 1), 'frac': round(rate / peak, 4)})
    rows.sort(key=lambda r_: -r_['ms_per_step'])
    return rows
