"""
Variation table: flam3 number, genome parameters and their interpolation domain.

Same information as cuburn/genome/variations.py:28-127 (names, numbers, parameter
names/defaults/domains) plus what the device side needs: the order of a variation's
record in the parameter block (weight, genome parameters in sorted-name order, then
precalculated values — include/flame_hip.h (5)) and how each precalculated value is
derived (FL_OP_* of include/flame_hip.h (6)).
"""
import math

# Spline parameter descriptor: (default, interp) with interp in {'linear', 'mag'}
def _s(default=0.0): return (default, 'linear')
def _ss(default=1.0): return (default, 'mag')

# name -> (flam3 number, {param: (default, interp)})
_TABLE = [
    (0, 'linear', {}), (1, 'sinusoidal', {}), (2, 'spherical', {}), (3, 'swirl', {}),
    (4, 'horseshoe', {}), (5, 'polar', {}), (6, 'handkerchief', {}), (7, 'heart', {}),
    (8, 'disc', {}), (9, 'spiral', {}), (10, 'hyperbolic', {}), (11, 'diamond', {}),
    (12, 'ex', {}), (13, 'julia', {}), (14, 'bent', {}), (15, 'waves', {}),
    (16, 'fisheye', {}), (17, 'popcorn', {}), (18, 'exponential', {}), (19, 'power', {}),
    (20, 'cosine', {}), (21, 'rings', {}), (22, 'fan', {}),
    (23, 'blob', dict(low=_ss(), high=_ss(), waves=_ss())),
    (24, 'pdj', dict(a=_s(), b=_s(), c=_s(), d=_s())),
    (25, 'fan2', dict(x=_s(), y=_s())),
    (26, 'rings2', dict(val=_s())),
    (27, 'eyefish', {}), (28, 'bubble', {}), (29, 'cylinder', {}),
    (30, 'perspective', dict(angle=_s(), dist=_ss())),
    (31, 'noise', {}),
    (32, 'julian', dict(power=_ss(), dist=_ss())),
    (33, 'juliascope', dict(power=_ss(), dist=_ss())),
    (34, 'blur', {}), (35, 'gaussian_blur', {}),
    (36, 'radial_blur', dict(angle=_s())),
    (37, 'pie', dict(slices=_s(6), rotation=_s(), thickness=_s(0.5))),
    (38, 'ngon', dict(sides=_s(5), power=_s(3), circle=_s(1), corners=_s(2))),
    (39, 'curl', dict(c1=_s(1), c2=_s())),
    (40, 'rectangles', dict(x=_s(), y=_s())),
    (41, 'arch', {}), (42, 'tangent', {}), (43, 'square', {}), (44, 'rays', {}),
    (45, 'blade', {}), (46, 'secant2', {}), (48, 'cross', {}),
    (49, 'disc2', dict(rot=_s(), twist=_s())),
    (50, 'super_shape', dict(rnd=_s(), m=_s(), n1=_ss(), n2=_s(1), n3=_s(1), holes=_s())),
    (51, 'flower', dict(holes=_s(), petals=_s())),
    (52, 'conic', dict(holes=_s(), eccentricity=_s(1))),
    (53, 'parabola', dict(height=_ss(), width=_ss())),
    (54, 'bent2', dict(x=_ss(), y=_ss())),
    (55, 'bipolar', dict(shift=_s())),
    (56, 'boarders', {}), (57, 'butterfly', {}),
    (58, 'cell', dict(size=_ss())),
    (59, 'cpow', dict(r=_ss(), i=_s(), power=_ss())),
    (60, 'curve', dict(xamp=_s(), yamp=_s(), xlength=_ss(), ylength=_ss())),
    (61, 'edisc', {}), (62, 'elliptic', {}),
    (63, 'escher', dict(beta=_s())),
    (64, 'foci', {}),
    (65, 'lazysusan', dict(x=_s(), y=_s(), twist=_s(), space=_s(), spin=_s())),
    (66, 'loonie', {}), (67, 'pre_blur', {}),
    (68, 'modulus', dict(x=_s(), y=_s())),
    (69, 'oscope', dict(separation=_s(1), frequency=_ss(math.pi), amplitude=_ss(), damping=_s())),
    (70, 'polar2', {}),
    (71, 'popcorn2', dict(x=_s(), y=_s(), c=_s())),
    (72, 'scry', {}),
    (73, 'separation', dict(x=_s(), xinside=_s(), y=_s(), yinside=_s())),
    (74, 'split', dict(xsize=_s(), ysize=_s())),
    (75, 'splits', dict(x=_s(), y=_s())),
    (76, 'stripes', dict(space=_s(), warp=_s())),
    (77, 'wedge', dict(angle=_s(), hole=_s(), count=_ss(), swirl=_s())),
    (80, 'whorl', dict(inside=_s(), outside=_s())),
    (81, 'waves2', dict(scalex=_ss(), scaley=_ss(), freqx=_ss(math.pi), freqy=_ss(math.pi))),
    (82, 'exp', {}), (83, 'log', {}), (84, 'sin', {}), (85, 'cos', {}), (86, 'tan', {}),
    (87, 'sec', {}), (88, 'csc', {}), (89, 'cot', {}), (90, 'sinh', {}), (91, 'cosh', {}),
    (92, 'tanh', {}), (93, 'sech', {}), (94, 'csch', {}), (95, 'coth', {}),
    (97, 'flux', dict(spread=_s())),
    (98, 'mobius', dict(re_a=_s(), im_a=_s(), re_b=_s(), im_b=_s(),
                        re_c=_s(), im_c=_s(), re_d=_s(), im_d=_s())),
]

var_names = dict((num, name) for num, name, _ in _TABLE)
var_ids = dict((name, num) for num, name, _ in _TABLE)
# name -> {param: (default, interp)}; 'weight' is implicit (default 0, linear)
var_params = dict((name, dict(p, weight=_s())) for _, name, p in _TABLE)

# Precalculated values appended to a variation's record, in order:
# (name, op kind, source spec) — cuburn/code/variations.py precalc blocks
#   waves        :136-140   dx2, dy2 from the xform's pre_affine offset
#   perspective  :267-273   mdist, sin, cos
#   julian(scope):292-294   cn = dist / (2 power)
#   curve        :630-634   x2, y2
var_precalc = {
    'waves': [('dx2', 'invsq', 'pre_affine.offset.x'), ('dy2', 'invsq', 'pre_affine.offset.y')],
    'perspective': [('mdist,sin,cos', 'persp', ('angle', 'dist'))],
    'julian': [('cn', 'ratio2', ('dist', 'power'))],
    'juliascope': [('cn', 'ratio2', ('dist', 'power'))],
    'curve': [('x2', 'invsq_max', 'xlength'), ('y2', 'invsq_max', 'ylength')],
}

def record_layout(name):
    """Names of the floats of variation ``name``'s record after the weight."""
    names = sorted(k for k in var_params[name] if k != 'weight')
    for pname, kind, _ in var_precalc.get(name, ()):
        names.extend(pname.split(','))
    return names
