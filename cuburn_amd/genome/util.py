"""Small genome helpers: dotted access, spec resolution, palette codec
(role of cuburn/genome/util.py:8-97)."""
import base64
import numpy as np
from . import spectypes


def get(dct, default, *keys):
    if len(keys) == 1:
        keys = keys[0].split('.')
    for k in keys:
        if not isinstance(dct, dict) or k not in dct:
            return default
        dct = dct[k]
    return dct


def flatten(src):
    out = {}
    def walk(d, ctx):
        for k, v in d.items():
            if isinstance(v, dict):
                walk(v, ctx + (str(k),))
            else:
                out['.'.join(ctx + (str(k),))] = v
    walk(src, ())
    return out


def unflatten(dct):
    out = {}
    for k, v in dct.items():
        parts = k.split('.')
        d = out
        for p in parts[:-1]:
            d = d.setdefault(p, {})
        d[parts[-1]] = v
    return out


def resolve_spec(sp, path):
    for name in path:
        sp = sp.type if isinstance(sp, spectypes.Map) else sp[name]
    return sp


def palette_decode(datastrs):
    """['rgb8', b64 chunk, ...] -> (256, 4) float32 RGBA in [0, 1] (cuburn/genome/util.py:75-87)."""
    if datastrs[0] != 'rgb8':
        raise NotImplementedError(datastrs[0])
    raw = base64.b64decode(''.join(datastrs[1:]))
    pal = np.frombuffer(raw, np.uint8).reshape(256, 3)
    data = np.ones((256, 4), np.float32)
    data[:, :3] = pal / 255.0
    return data


def palette_encode(data, format='rgb8'):
    """(256, >=3) floats -> ['rgb8', 64-char b64 chunks...] (cuburn/genome/util.py:89-97)."""
    if format != 'rgb8':
        raise NotImplementedError(format)
    clamp = np.clip(np.round(np.asarray(data)[:, :3] * 255.0), 0, 255).astype(np.uint8)
    enc = base64.b64encode(clamp.tobytes()).decode('ascii')
    return ['rgb8'] + [enc[i:i + 64] for i in range(0, len(enc), 64)]
