#!/usr/bin/env python3
"""
Generate the golden fixtures of tests/golden/ by IMPORTING THE REFERENCE'S HOST CODE.

Runs only in the authoring container (needs /root/reference); the fixtures it writes are
plain data (JSON) and are committed, so nothing at test time reads the reference.

The reference is Python 2 + PyCUDA.  Recipe: copy it to a scratch
directory, run lib2to3 over it, alias Cython's vendored Tempita as `tempita`, stub the
pycuda modules (no device code is executed — only host-side functions are called), and
apply four small py3 patches.  Nothing of the converted sources is kept.

    python tests/golden/make_golden.py
"""
import base64
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'


def prepare_reference():
    tmp = tempfile.mkdtemp(prefix='cuburn_ref_py3_')
    dst = os.path.join(tmp, 'ref')
    shutil.copytree(REF, dst)
    subprocess.run(['chmod', '-R', 'u+w', dst], check=True)
    subprocess.run([sys.executable, '-m', 'lib2to3', '-w', '-n', 'cuburn', 'helpers/shuf.py'],
                   cwd=dst, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    def patch(rel, old, new):
        p = os.path.join(dst, rel)
        s = open(p).read()
        assert old in s, (rel, old)
        open(p, 'w').write(s.replace(old, new))
    patch('cuburn/filters.py', 'from . import code.filters',
          'from . import code\nfrom .code import filters as _cf\ncode.filters = _cf')
    patch('cuburn/code/util.py', "    if isinstance(s, str):\n        s = s.encode('utf-8')\n"
          "    return '\"%s\"' % s.encode(\"string_escape\")",
          "    return '\"%s\"' % s.encode('unicode_escape').decode('ascii')")
    patch('cuburn/code/mwc.py', 'with open(pfpath) as fp', "with open(pfpath, 'rb') as fp")
    patch('cuburn/code/iter.py', 'NTHREADS / 32', 'NTHREADS // 32')

    import Cython.Tempita as tempita
    sys.modules['tempita'] = tempita
    for n in ['pycuda', 'pycuda.driver', 'pycuda.compiler', 'pycuda.tools', 'pycuda.gpuarray', 'pycuda.autoinit']:
        sys.modules[n] = types.ModuleType(n)
    sys.modules['pycuda.driver'].Event = object
    sys.modules['pycuda.gpuarray'].vec = None
    sys.path.insert(0, dst)
    return tmp, dst


def genome_xml_cases():
    """flam3 XML inputs for the conversion fixtures (the first is the reference's own test flame,
    genome/tests/test_convert.py:20-36)."""
    return {
        'ref_test': """
<flame time="0" size="1280 960" center="0.01 0.02" scale="40" oversample="2"
    filter="1" quality="500" batches="50" brightness="4" gamma="4"
    url="test.com" nick="strobe" >
    <color index="0" rgb="1 2 3"/>
    <xform weight="0.1" color="0" hyperbolic="0.1"
        coefs="01 0.2 -0.3 0.4 -0.5 0.6"/>
</flame>""",
        'rich': """
<flames name="pack">
<flame name="rich" size="640 480" center="-0.25 0.5" scale="120" rotate="30" filter="0.8"
    brightness="3.5" gamma="3" gamma_threshold="0.02" highlight_power="1.5" vibrancy="0.9"
    estimator_radius="9" estimator_minimum="0.5" estimator_curve="0.4">
    <xform weight="0.5" color="0.25 0.1" symmetry="0.5" linear="0.6" spherical="0.4"
        coefs="0.7 0.1 -0.2 0.9 0.3 -0.4" post="1.1 0 0 0.9 0.05 0" chaos="1 0.5 2"/>
    <xform weight="0.25" color="0.75" color_speed="0.3" opacity="0.5" animate="1"
        julian="1" julian_power="3" julian_dist="0.8" curl="0.2" curl_c1="0.1" curl_c2="-0.3"
        coefs="-0.5 0.5 0.5 0.5 0 0"/>
    <xform weight="0.25" color="1" symmetry="1" blob="1" blob_low="0.2" blob_high="1.2" blob_waves="5"
        coefs="1 0 0 1 0 0" post="1 0 0 1 0 0"/>
    <finalxform color="0" symmetry="1" linear="1" coefs="1.2 0 0 1.2 0.1 0.1"/>
    <color index="3" rgb="255 128 0"/>
    <color index="255" rgb="10 20 30"/>
</flame>
<flame name="symm" size="100 100" scale="25" brightness="4" gamma="4">
    <symmetry kind="-3"/>
    <xform weight="1" color="0" swirl="1" coefs="0.5 0 0 0.5 0.5 0"/>
</flame>
</flames>""",
    }


def genome_db_case(nodes):
    """A one-file genome database: converted nodes, hand-written nodes (velocities, wide spreads,
    hole / identity variations, final xforms, a base chain) and edges with edits."""
    A = {'type': 'node', 'name': 'A',
         'camera': {'scale': 0.4, 'rotation': [10, 45], 'center': {'x': 0.1, 'y': -0.2}},
         'filters': {'logscale': {'brightness': 5}},
         'time': {'frame_width': 1.5},
         'palette': ['rgb8', 'AAAA' * 256],
         'xforms': {
             '0': {'weight': 0.5, 'color': 0.1, 'pre_affine': {'angle': [30, -360], 'spread': 40, 'magnitude': {'x': 0.7, 'y': 0.6}},
                   'variations': {'linear': {'weight': 1}}},
             '1': {'weight': 0.3, 'color': 0.9, 'color_speed': 0.25, 'pre_affine': {'angle': 200, 'spread': 120, 'offset': {'x': 0.3, 'y': 0.1}},
                   'post_affine': {'angle': 50, 'spread': 100},
                   'variations': {'spherical': {'weight': 0.8}, 'linear': {'weight': 0.2}}},
             '2': {'weight': 0.2, 'color': 0.5, 'pre_affine': {'angle': [90, 180], 'spread': 45},
                   'variations': {'blob': {'weight': 1, 'low': 0.3, 'high': 1.1, 'waves': 4}, 'fan2': {'weight': 0.5, 'x': 0.2, 'y': 0.4}}},
         }}
    B = {'type': 'node', 'name': 'B',
         'camera': {'scale': 0.5, 'rotation': [350, -90]},
         'filters': {'colorclip': {'gamma': 3.5}},
         'blend': {'xform_sort': 'weight'},
         'palette': ['rgb8', '////' * 256],
         'xforms': {
             '0': {'weight': 0.6, 'color': 0.0, 'pre_affine': {'angle': [75, 360], 'spread': 45},
                   'variations': {'swirl': {'weight': 1}}},
             '1': {'weight': 0.4, 'color': 1.0, 'pre_affine': {'angle': 10, 'spread': 130},
                   'post_affine': {'angle': 20, 'spread': 95},
                   'variations': {'julian': {'weight': 1, 'power': 3, 'dist': 0.9}}},
         },
         'final_xform': {'color': 0.3, 'color_speed': 0.1, 'pre_affine': {'angle': 45, 'spread': 45, 'magnitude': {'x': 1.1, 'y': 1.1}},
                         'variations': {'linear': {'weight': 1}}}}
    C = {'type': 'node', 'name': 'C', 'base': 'A',
         'camera': {'scale': 0.8},
         'xforms': {'3': {'weight': 0.1, 'color': 0.2, 'variations': {'perspective': {'weight': 1, 'angle': 0.4, 'dist': 2}}}}}
    edge1 = {'type': 'edge', 'link': {'src': 'A@0.25', 'dst': 'B@0'},
             'blend': {'duration': 3, 'xform_sort': 'weightflip'},
             'camera': {'scale': [0.5, 0.45]},
             'xforms': {'src': {'0': {'weight': [0.5, 0.7]}}, 'dst': {'1': {'color': [0.25, 0.6], 'weight': [0, 0.1, 1, 0.5]}}}}
    edge2 = {'type': 'edge', 'link': {'src': 'B@0.5', 'dst': 'C@-0.25'},
             'blend': {'duration': 2, 'xform_sort': 'natural', 'xform_map': [['0', '2'], ['1', 'dup'], ['pad', '0']]},
             'final_xform': {'color': [0.5, 0.9]}}
    edge3 = {'type': 'edge', 'base': 'edge1', 'link': {'src': 'A@0.25', 'dst': 'B@0'},
             'blend': {'xform_sort': 'color'},
             'xforms': {'src': {'0': {'weight': [0.25, 0.1]}}}}
    out = {'type': 'onefiledb', 'A': A, 'B': B, 'C': C, 'edge1': edge1, 'edge2': edge2, 'edge3': edge3}
    out['X_ref_test'] = nodes['ref_test'][0]
    rich = json.loads(json.dumps(nodes['rich'][0]))
    for xf in rich['xforms'].values():
        xf.pop('chaos', None)        # converted but not in the node schema: resolve() raises KeyError on it
    out['X_rich'] = rich
    out['X_symm'] = nodes['rich'][1]
    return out


def jsonable(obj):
    """bytes -> str (py3 base64), numpy scalars -> python, integer dict keys -> str (what json does)."""
    if isinstance(obj, dict):
        return dict((str(k), jsonable(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return [jsonable(v) for v in obj]
    if isinstance(obj, bytes):
        return obj.decode('ascii')
    if isinstance(obj, np.generic):
        return obj.item()
    return obj


def dump(name, obj):
    with open(os.path.join(HERE, name), 'w') as fp:
        json.dump(obj, fp, indent=1, sort_keys=True)
    print('wrote', name)


def main():
    tmp, dst = prepare_reference()
    sys.path.insert(0, REPO)
    from cuburn.code import mwc, iter as ref_iter, util, interp
    from cuburn import render, profile, filters as ref_filters
    from cuburn.genome import use, specs, variations as ref_vars
    from cuburn.genome.util import palette_decode
    from cuburn_amd import configs

    # ---- 1. MWC: seed rows, raw streams and per-thread sums from the numpy model (mwc.py:101-113)
    seeds = mwc.make_seeds(64, host_seed=42)
    mults = seeds[:, 0].astype(np.uint64)
    states, carries = seeds[:, 1].copy(), seeds[:, 2].copy()
    sums = np.zeros(64, dtype=np.uint64)
    stream0 = []
    for i in range(200):
        step = np.frombuffer((mults * states + carries).data, dtype=np.uint32).reshape((64, 2))
        states[:] = step[:, 0]
        carries[:] = step[:, 1]
        sums += states
        if i < 16:
            stream0.append(int(states[0]))
    prim = open(os.path.join(REF, 'cuburn/code/primes.bin'), 'rb').read()
    dump('mwc.json', {
        'host_seed': 42, 'seeds': seeds.tolist(), 'rounds': 200, 'sums': [int(s) for s in sums],
        'stream0': stream0, 'final_states': states.tolist(), 'final_carries': carries.tolist(),
        'mults_sha256': hashlib.sha256(prim).hexdigest(), 'mults_len': len(prim) // 4,
        'mults_head': np.frombuffer(prim[:64], '<u4').tolist(),
    })

    # ---- 2. calc_dim (render.py:79-89)
    sizes = [(512, 512), (640, 360), (1280, 720), (1920, 1080), (3840, 2160), (7680, 4320), (1, 1), (33, 17), (1000, 999)]
    dump('calc_dim.json', [[w, h] + list(render.Framebuffers.calc_dim(w, h)) for w, h in sizes])

    # ---- 3. Spline normalisation + host evaluation (use.py:121-185)
    cases = [1.5, [2, 3], [45, -360, -315, -360, 0.3, 100], [1, 2, 3, 4, 0.5, 7, 0.2, 9], [0.5, 0, 0.5, 0]]
    sp = []
    for k in cases:
        for scale in (1, 2.5):
            ev = use.SplineEval(k, scale)
            ts = [0, 0.1, 0.25, 0.5, 0.77, 1]
            sp.append({'knots': k, 'scale': scale, 'normalized': ev.knots.tolist(),
                       't': ts, 'val': [ev(t) for t in ts], 'deriv': [ev(t, 1) for t in ts]})
    dump('splines.json', sp)

    # ---- 4. Packer: per-path knot rows of GenomePacker.pack for the config genomes (interp.py:207-232)
    packs = {}
    for name in ('cfg1', 'cfg2', 'cfg3', 'cfg5', 'allvars'):           # allvars: every variation, every parameter (configs.allvars)
        gnm, prof = configs.allvars() if name == 'allvars' else configs.CONFIGS[name]()
        packer, lib = ref_iter.mkiterlib(gnm)
        util.assemble_code(lib)
        times, knots = packer.pack(gnm)
        rows = {}
        for i, path in enumerate(packer.genome):
            n = int(np.sum(times[i] < 1e8))
            rows['.'.join(path)] = {'times': times[i, :n].tolist(), 'knots': knots[i, :n].tolist()}
        packs[name] = {'rows': rows, 'n_genome': len(packer.genome), 'n_packed': len(packer.packed),
                       'packed': ['.'.join(p) for p in packer.packed]}
    dump('packer.json', packs)

    # ---- 5. Profile: frame times, still quirk, wrapped scalars (profile.py:97-127)
    prof_out = {}
    import copy
    builtin0 = copy.deepcopy(profile.BUILTIN)
    def getp(args):
        # the reference mutates BUILTIN[...] in place (profile.py:84-92); start each case fresh
        profile.BUILTIN.clear()
        profile.BUILTIN.update(copy.deepcopy(builtin0))
        return profile.get_from_args(profile.add_args().parse_args(args))
    name, prof = getp([])
    gprof = profile.wrap(prof, {'type': 'edge'})
    fr = profile.enumerate_times(gprof)
    prof_out['default'] = {'n': len(fr), 'first': [fr[0][0], list(map(float, fr[0][1]))],
                           'last': [fr[-1][0], list(map(float, fr[-1][1]))]}
    name, prof = getp(['-P', '720p', '--fps=1', '--duration=5', '--shard=5'])
    fr = profile.enumerate_times(profile.wrap(prof, {'type': 'edge'}))
    prof_out['shard'] = [[f[0], list(map(float, f[1]))] for f in fr]
    name, prof = getp(['--still'])
    gprof = profile.wrap(prof, {'type': 'animation'})
    fr = profile.enumerate_times(gprof)
    prof_out['still'] = [[f[0], list(map(float, f[1]))] for f in fr]
    name, prof = getp(['-P', 'preview'])
    fr = profile.enumerate_times(profile.wrap(prof, {'type': 'animation'}))
    prof_out['preview'] = {'n': len(fr), 'ids': [f[0] for f in fr[:4]]}
    dump('profile.json', prof_out)

    # ---- 6. Filter scalars (filters.py:11-16,74-76,100-106,132-136) on the cfg3 genome / profile
    gnm, prof = configs.cfg3()
    gprof = profile.wrap(prof, gnm)
    tc = 0.3
    dim = render.Framebuffers.calc_dim(gprof.width, gprof.height)
    f32 = np.float32
    def gauss(stdev):
        coefs = np.exp(np.float32(np.arange(-3, 4)) ** 2 / (-2 * stdev ** 2)).astype(np.float32)
        return (coefs / np.sum(coefs)).tolist()
    pb = gprof.filters.bilateral
    pl = gprof.filters.logscale
    gam, lin, lingam = ref_filters.calc_lingam(gprof.filters.colorclip, tc)
    area = dim.h / (pl.scale(tc) ** 2 * dim.w)
    dump('filter_scalars.json', {
        'tc': tc, 'gauss_1': gauss(1), 'gauss_07': gauss(0.7),
        'bilateral': [float(f32(pb.spatial_std(tc) * dim.w / 1920.)), float(f32(pb.color_std(tc))),
                      float(f32(pb.density_std(tc))), float(f32(pb.density_pow(tc))), float(f32(pb.gradient(tc)))],
        'logscale': [float(f32(pl.brightness(tc) * 268 / 256)), float(f32(1.0 / (area * gprof.spp(tc))))],
        'lingam': [float(gam), float(lin), float(lingam)],
        'colorclip': [float(f32(gprof.filters.colorclip.vibrance(tc))), float(f32(gprof.filters.colorclip.highlight_power(tc)))],
        'smearclip_width': float(f32(gprof.filters.smearclip.width(tc))),
        'spp': float(gprof.spp(tc)), 'frame_width': float(gprof.frame_width(tc)),
    })

    # ---- 7. Point shuffle: helpers/shuf.py:75-80 (only the function body is executed)
    src = open(os.path.join(dst, 'helpers/shuf.py')).read()
    start = src.index('def shuf_simple(a):')
    end = src.index('print(', start)
    ns = {'np': np, 'w': 32, 't': 256}
    exec(src[start:end].replace('t/w', 't//w'), ns)
    a = np.arange(256, dtype=np.int32)
    rounds = [a.tolist()]
    for _ in range(3):
        a = ns['shuf_simple'](a)
        rounds.append(a.tolist())
    dump('shuffle.json', {'w': 32, 't': 256, 'rounds': rounds})

    # ---- 8. Variation spec (genome/variations.py:28-127) and schema defaults (specs.py)
    vs = {}
    for num, name in ref_vars.var_names.items():
        vs[name] = {'num': num, 'params': dict((k, [float(s.default), s.interp])
                                               for k, s in ref_vars.var_params[name].items())}
    dump('var_spec.json', vs)
    def spec_defaults(spec, out, prefix=()):
        for k, v in spec.items():
            if isinstance(v, dict):
                spec_defaults(v, out, prefix + (k,))
            elif hasattr(v, 'interp'):
                out['.'.join(prefix + (k,))] = [float(v.default), v.interp]
    sd = {}
    spec_defaults(specs.xform, sd, ('xform',))
    spec_defaults(specs.camera, sd, ('camera',))
    spec_defaults(specs.filters, sd, ('filters',))
    sd = dict((k, v) for k, v in sd.items() if '.variations.' not in k)
    dump('spec_defaults.json', sd)

    # ---- 9. Palette codec (genome/util.py:75-87)
    pal = (np.arange(768) % 251).astype(np.uint8)
    enc = base64.b64encode(pal.tobytes()).decode()
    dec = palette_decode(['rgb8'] + [enc[i:i + 64] for i in range(0, len(enc), 64)])
    dump('palette.json', {'b64': enc, 'decoded_head': dec[:4].tolist(), 'decoded_sum': float(dec.sum())})

    # ---- 10. Genome front-end: flam3 XML -> node (genome/convert.py), node/edge -> animation
    #          (genome/blend.py), house-style JSON text (genome/util.py:99-144).  The reference is
    #          py2: its round() (halves away from zero) is put back into the converted module.
    import math
    from cuburn.genome import convert as ref_convert, blend as ref_blend, db as ref_db
    from cuburn.genome import util as ref_gutil
    ref_blend.round = lambda x: math.copysign(math.floor(abs(x) + 0.5), x)
    ref_gutil.basestring = str
    front = {'xml': {}, 'nodes': {}, 'anims': {}, 'json_text': {}}
    for name, xml in genome_xml_cases().items():
        flames = ref_convert.XMLGenomeParser.parse(xml)
        front['xml'][name] = xml
        front['nodes'][name] = [jsonable(ref_convert.flam3_to_node(f)) for f in flames]
    dbdoc = genome_db_case(front['nodes'])
    front['db'] = dbdoc
    gdb = ref_db.OneFileDB(json.loads(json.dumps(dbdoc)))
    for key, doc in sorted(dbdoc.items()):
        if not isinstance(doc, dict):
            continue
        if doc['type'] == 'node':
            for half in (False, True):
                a = ref_blend.node_to_anim(gdb, json.loads(json.dumps(doc)), half)
                front['anims']['%s/%s' % (key, 'half' if half else 'full')] = a
        elif doc['type'] == 'edge':
            front['anims'][key] = ref_blend.edge_to_anim(gdb, json.loads(json.dumps(doc)))
    for k, a in front['anims'].items():
        front['json_text'][k] = ref_gutil.json_encode(json.loads(json.dumps(a)))
    front['tospline'] = []
    from cuburn.genome import spectypes as ref_st
    for spl_args in (dict(), dict(period=360), dict(var=True), dict(default=45, period=360)):
        spl = ref_st.spline(spl_args.get('default', 0), period=spl_args.get('period'))._replace(var=spl_args.get('var', False))
        for src_v, dst_v, edit, dur in [
                (None, None, None, 1), (1, 1, None, 1), (1, 2, None, 1), (None, 3, None, 2),
                ([10, 180], [20, 180], None, 1), ([10, 180], [20, 180], None, 2), ([10, -360], [10, -360], None, 1),
                ([350, 90], [5, 0], None, 1), ([0, 720], [90, 0], None, 0.5), (30, [60, -45], None, 3),
                (1, 2, [0, 5, 1, 7], 1), (1, 2, [0.5, 9], 1), ([0, 360], [0, 360], [0, 725, 1, -10, 0.25, 3], 1),
                (0.5, 0.5, [0.3, None], 1)]:
            front['tospline'].append({'spl': spl_args, 'src': src_v, 'dst': dst_v, 'edit': edit, 'duration': dur,
                                      'out': ref_blend.tospline(spl, src_v, dst_v, edit, dur)})
    dump('genome_front.json', front)

    shutil.rmtree(tmp)


if __name__ == '__main__':
    main()
