"""Small helpers with the names the reference's utils.py exposes (utils.py:33-69)."""
from datetime import datetime

import numpy as np


def log_print(*args):
    print('{}:'.format(datetime.now().strftime('%Y-%m-%d %H:%M:%S')), *args, flush=True)


def get_model(model_name: str):
    """Resolve a model class by name on Networks (utils.py:38-40) -- part of the saved-model contract."""
    import Networks as Nets
    return getattr(Nets, model_name)


def load_model(model_name: str, *args, **kwargs):
    model = get_model(model_name)
    return model(*args, **kwargs) if (args or kwargs) else model


def bbox_crop(img, margin=10):
    """Tight bounding box of the non-zero region, grown by `margin` (clipped)."""
    ys = np.flatnonzero(np.any(img, axis=1))
    xs = np.flatnonzero(np.any(img, axis=0))
    rmin, rmax = max(0, ys[0] - margin), min(img.shape[0], ys[-1] + margin)
    cmin, cmax = max(0, xs[0] - margin), min(img.shape[1], xs[-1] + margin)
    return img[rmin:rmax, cmin:cmax], (rmin, rmax, cmin, cmax)


def bbox_fill(img, crop, loc):
    rmin, rmax, cmin, cmax = loc
    out = img.copy()
    out[rmin:rmax, cmin:cmax] = crop
    return out
