"""
Schema node types for the genome / profile documents (role of cuburn/genome/spectypes.py:279-326).

A spec is a nested dict whose leaves are these descriptors.  ``Spline`` is the animated
parameter type; its JSON encodings (number, 2-list, >=4-list) are decoded by
``use.SplineEval.normalize``.
"""
from collections import namedtuple

Spline = namedtuple('Spline', 'default min max interp period doc var')
Scalar = namedtuple('Scalar', 'default doc')
RefScalar = namedtuple('RefScalar', 'default ref doc')
String = namedtuple('String', 'doc')
Enum = namedtuple('Enum', 'choices default doc')
Map = namedtuple('Map', 'type doc')
List = namedtuple('List', 'type default doc')
Palette = namedtuple('Palette', '')


def spline(default=0, min=None, max=None, interp='linear', period=None, d=None, var=False):
    return Spline(default, min, max, interp, period, d, var)


def scalespline(default=1, min=0, max=None, d=None, var=False):
    """A magnitude-domain spline (interpolated in the lin-log domain on device)."""
    return Spline(default, min, max, 'mag', None, d, var)


def scalar(default, d=None): return Scalar(default, d)
def refscalar(default, ref, d=None): return RefScalar(default, ref, d)
def map_(type, d=None): return Map(type, d)
def list_(type, default=(), d=None): return List(type, default, d)


def enum(choices, default=None, d=None):
    if isinstance(choices, str):
        choices = choices.split()
    return Enum(list(choices), default, d)


class XYPair(dict):
    """An {x, y} pair of one spline type."""
    def __init__(self, type):
        super().__init__(x=type, y=type)
        self.type = type
