"""
Genome front-end: flam3 XML -> node, node / edge -> animation, JSON text.

Known answers: the reference's own conversion test (cuburn/genome/tests/test_convert.py:48-68)
restated below, and tests/golden/genome_front.json, produced by running the reference's
convert / blend / json_encode on the same documents (tests/golden/make_golden.py section 10).
"""
import binascii
import copy
import json
import os

import numpy as np
import pytest

from common import REPO
from cuburn_amd.genome import blend, convert, store, specs, spectypes, util

GOLD = json.load(open(os.path.join(REPO, 'tests', 'golden', 'genome_front.json')))


def approx_equal(a, b, path=''):
    """Same structure; numbers equal to 1e-12 relative (both sides are float64 arithmetic)."""
    if isinstance(b, dict):
        assert isinstance(a, dict), path
        assert sorted(map(str, a.keys())) == sorted(b.keys()), (path, sorted(map(str, a.keys())), sorted(b.keys()))
        for k in a:
            approx_equal(a[k], b[str(k)], path + '.' + str(k))
    elif isinstance(b, list):
        assert isinstance(a, (list, tuple)) and len(a) == len(b), (path, a, b)
        for i, (x, y) in enumerate(zip(a, b)):
            approx_equal(x, y, '%s[%d]' % (path, i))
    elif isinstance(b, float) or isinstance(a, float):
        assert a == pytest.approx(b, rel=1e-12, abs=1e-12), (path, a, b)
    else:
        assert a == b, (path, a, b)


# ---------------------------------------------------------------- the reference's own test, restated
def _palette_src():
    values = np.zeros((256, 4), 'u1')
    values[:, 0] = range(256)
    values[:, 1], values[:, 2], values[:, 3] = 1, 2, 3
    return '<palettes><palette number="0" name="synthetic" data="%s\n"/></palettes>' % (
        binascii.b2a_hex(values.tobytes()).decode())


def test_xml_palette_parser_known_answer():
    parser = convert.XMLPaletteParser(_palette_src())
    assert 'synthetic' in parser.names and 0 in parser.numbers
    assert list(parser.numbers[0][0]) == [0, 1 / 255., 2 / 255., 3 / 255.]
    assert list(parser.numbers[0][255]) == [1, 1 / 255., 2 / 255., 3 / 255.]


def test_flam3_to_node_known_answer():
    parsed = convert.XMLGenomeParser.parse(GOLD['xml']['ref_test'])
    node = convert.flam3_to_node(parsed[0])
    palette = node.pop('palette')
    assert node == dict(
        type='node',
        author=dict(url='http://test.com', name='strobe'),
        camera=dict(dither_width=1.0, scale=0.03125, center=dict(x=0.01, y=0.02)),
        filters=dict(logscale=dict(brightness=4.0), colorclip=dict(gamma=4.0)),
        xforms={'0': dict(
            color=0.0, variations=dict(hyperbolic=dict(weight=0.1)),
            pre_affine=dict(spread=32.220017414088105, angle=[20.91008494006789, -360],
                            magnitude=dict(x=1.019803902718557, y=0.5), offset=dict(x=-0.5, y=-0.6)),
            weight=0.1)})
    assert palette[0] == 'rgb8' and palette[1][:8] == 'AQID////'


def test_stock_palette_lookup(tmp_path, monkeypatch):
    """palette="N" resolves through the flam3 palette file (not installed here: use a synthetic one)."""
    p = tmp_path / 'flam3-palettes.xml'
    p.write_text(_palette_src())
    monkeypatch.setattr(convert.XMLPaletteParser, '_locations', [str(p)])
    monkeypatch.setattr(convert.XMLPaletteParser, '_names', None)
    monkeypatch.setattr(convert.XMLPaletteParser, '_numbers', None)
    src = GOLD['xml']['ref_test'].replace('nick="strobe" >', 'nick="strobe" palette="0">').replace(
        '<color index="0" rgb="1 2 3"/>', '')
    node = convert.flam3_to_node(convert.XMLGenomeParser.parse(src)[0])
    pal = util.palette_decode(node['palette'])
    assert pal[5, 0] == pytest.approx(5 / 255.) and pal[5, 1] == pytest.approx(1 / 255.)
    monkeypatch.setattr(convert.XMLPaletteParser, '_locations', [str(tmp_path / 'missing.xml')])
    monkeypatch.setattr(convert.XMLPaletteParser, '_names', None)
    with pytest.raises(IOError):
        convert.XMLPaletteParser.lookup(0)


# ---------------------------------------------------------------- golden vectors from the reference
@pytest.mark.parametrize('name', sorted(GOLD['xml']))
def test_convert_matches_reference(name):
    flames = convert.XMLGenomeParser.parse(GOLD['xml'][name])
    assert len(flames) == len(GOLD['nodes'][name])
    for f, want in zip(flames, GOLD['nodes'][name]):
        approx_equal(convert.flam3_to_node(f), want, name)


def test_convert_affine_identity_and_flip():
    assert convert.convert_affine('1 0 0 1 0 0') is None
    a = convert.convert_affine('0 1 -1 0 2 3')          # 90 degree rotation in flam3 space
    assert a['offset'] == {'x': 2.0, 'y': -3.0}
    assert a['magnitude'] == {'x': 1.0, 'y': 1.0}
    assert a['spread'] == pytest.approx(45.0)


@pytest.mark.parametrize('key', sorted(GOLD['anims']))
def test_node_and_edge_to_anim_match_reference(key):
    gdb = store.GenomeStore.__new__(store.GenomeStore)
    gdb.docs, gdb.directory = copy.deepcopy(GOLD['db']), None
    ident, _, mode = key.partition('/')
    doc = gdb.get(ident)
    before = copy.deepcopy(doc)
    if doc['type'] == 'node':
        anim = blend.node_to_anim(gdb, doc, mode == 'half')
    else:
        anim = blend.edge_to_anim(gdb, doc)
    assert doc == before                                  # inputs are not modified
    assert anim['type'] == 'animation'
    approx_equal(anim, GOLD['anims'][key], key)
    # the result is a valid animation for the renderer's packer
    from cuburn_amd.packer import GenomePacker
    if 'chaos' not in json.dumps(anim):
        GenomePacker(json.loads(json.dumps(anim)))


@pytest.mark.parametrize('key', sorted(GOLD['json_text']))
def test_json_encode_matches_reference(key):
    text = util.json_encode(GOLD['anims'][key])
    assert text == GOLD['json_text'][key]
    approx_equal(json.loads(text), json.loads(GOLD['json_text'][key]))


def test_tospline_matches_reference():
    for case in GOLD['tospline']:
        a = case['spl']
        spl = spectypes.spline(a.get('default', 0), period=a.get('period'))._replace(var=a.get('var', False))
        got = blend.tospline(spl, case['src'], case['dst'], case['edit'], case['duration'])
        approx_equal(got, case['out'], str(case))


def test_tospline_rounds_half_turns_like_python2():
    """movement 0.5 turns: py2 round() gives one whole turn, py3's banker's rounding none."""
    spl = spectypes.spline(45, period=360)
    assert blend.tospline(spl, [0, 180], [0, 180], None, 1) == [0, 180, 360.0, 180]
    assert blend.tospline(spl, [0, -180], [0, -180], None, 1) == [0, -180, -360.0, -180]


def test_padding_xform_rules():
    assert blend.padding_xform({'variations': {'spherical': {'weight': 1}}}, False) == {
        'variations': {'linear': {'weight': -1}}, 'pre_affine': {'angle': 225}}
    p = blend.padding_xform({'pre_affine': {'spread': 120}, 'variations': {'blob': {'weight': 1}, 'fan2': {'weight': 1}}}, True)
    assert p['pre_affine'] == {'angle': 135, 'spread': 135} and p['weight'] == 0 and p['color_speed'] == 0
    assert sorted(p['variations']) == ['blob', 'fan2']
    assert p['variations']['blob'] == {'low': 1.0, 'high': 1.0, 'waves': 1.0, 'weight': 0.5}
    assert blend.padding_xform({}, False)['variations'] == {'linear': {'weight': 1}}


def test_sort_xforms_explicit_pairs_and_padding():
    sx = {'0': {'weight': 0.5}, '1': {'weight': 0.2}, '2': {'weight': 0.3}}
    dx = {'a': {'weight': 0.9}, 'b': {'weight': 0.1}}
    pairs = list(blend.sort_xforms(sx, dx, 'weight', [['0', 'b']]))
    assert pairs[0] == ('0', 'b')
    assert sorted(pairs[1:], key=str) == sorted([('1', 'a'), ('2', None)], key=str)
    flip = list(blend.sort_xforms(sx, dx, 'weightflip'))
    assert ('1', 'a') in flip and ('2', 'b') in flip and ('0', None) in flip


def test_resolve_unknown_key_is_an_error():
    gdb = store.GenomeStore.__new__(store.GenomeStore)
    gdb.docs, gdb.directory = {}, None
    with pytest.raises(KeyError):
        blend.resolve(gdb, {'type': 'node', 'xforms': {'0': {'chaos': {'0': 1}}}})


def test_db_get_anim_from_files(tmp_path):
    (tmp_path / 'A.json').write_text(json.dumps(GOLD['db']['A']))
    (tmp_path / 'B.json').write_text(json.dumps(GOLD['db']['B']))
    (tmp_path / 'edge1.json').write_text(json.dumps(GOLD['db']['edge1']))
    gdb = store.connect(str(tmp_path))
    assert gdb.directory == str(tmp_path)
    anim, base = gdb.animation('edge1')
    assert base == 'edge1'
    approx_equal(anim, GOLD['anims']['edge1'])
    anim, base = gdb.animation('A.json', half=True)
    assert base == 'A'
    approx_equal(anim, GOLD['anims']['A/half'])
    flame = tmp_path / 'x.flam3'
    flame.write_text(GOLD['xml']['ref_test'])
    anim, base = gdb.animation(str(flame))
    assert base == 'x' and anim['type'] == 'animation'
    approx_equal(anim, GOLD['anims']['X_ref_test/full'])
    one = tmp_path / 'all.json'
    one.write_text(json.dumps(GOLD['db']))
    onefile = store.connect(str(one))
    assert onefile.directory is None and onefile.get('B')['name'] == 'B'
    with pytest.raises(KeyError):
        onefile.get('nope')
    gdb.stash('tmp', GOLD['db']['B'])
    assert gdb.get('tmp')['name'] == 'B'


def test_genome_hash_depends_on_structure_only():
    a = copy.deepcopy(GOLD['anims']['A/full'])
    b = copy.deepcopy(a)
    b['camera']['scale'] = 123
    assert util.hash(a) == util.hash(b)
    b['camera']['spp'] = 1
    assert util.hash(a) != util.hash(b)
