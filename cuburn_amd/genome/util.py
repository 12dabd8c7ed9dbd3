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


def hash(gnm):
    """
    Structural hash of a genome: only WHICH keys are present matters (that is what decides
    the shape of the xform program), not their values (cuburn/genome/util.py:54-64).
    """
    from hashlib import sha1
    return sha1('\n'.join(flatten(gnm).keys()).encode('utf-8')).hexdigest()


# ------------------------------------------------------------------ genome-flavoured JSON text
def _quote(s):
    return '"%s"' % str(s).encode('unicode_escape').decode('ascii').replace('"', '\\"')


def _is_num(v):
    return isinstance(v, (int, float, np.number)) and not isinstance(v, bool)


def _layout(parts, brackets, indent):
    """One line if it fits in 70 columns, else one item per line with leading commas."""
    opening, closing = brackets
    line = opening + ', '.join(parts) + closing
    if '\n' not in line and len(line) + indent < 70:
        return line
    pad = ' ' * indent
    return '\n' + pad + opening + ' ' + ('\n' + pad + ', ').join(parts) + '\n' + pad + closing


def _encode(obj, indent):
    if isinstance(obj, dict):
        if not obj:
            return '{}'
        # numeric keys in numeric order first, then the rest alphabetically; colours as r, g, b
        keys = sorted(obj, key=lambda k: (0, int(k), '') if str(k).isdigit() else (1, 0, str(k)))
        if keys == ['b', 'g', 'r']:
            keys.reverse()
        return _layout(['%s: %s' % (_quote('%.6g' % k if _is_num(k) else k), _encode(obj[k], indent + 2))
                        for k in keys], '{}', indent)
    if isinstance(obj, (list, tuple)):
        parts = [_encode(v, indent + 2) for v in obj]
        if parts and len(parts) % 2 == 0 and _is_num(obj[1]):      # knot lists read as "t, v" pairs
            parts = [a + ', ' + b for a, b in zip(parts[::2], parts[1::2])]
        return _layout(parts, '[]', indent)
    if isinstance(obj, str):
        return _quote(obj)
    if _is_num(obj):
        return '%.6g' % obj
    raise TypeError("Don't know how to serialize %s of type %s" % (obj, type(obj)))


def json_encode(obj):
    """
    JSON text of a genome in the reference's compact house style
    (cuburn/genome/util.py:99-144): short containers on one line, long ones broken with
    leading commas, spline knot lists paired, numbers as %.6g.
    """
    text = _encode(obj, 0).lstrip()
    return '\n'.join(l.rstrip() for l in text.split('\n')) + '\n'
