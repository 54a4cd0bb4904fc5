"""
Unit-less stand-in for `unyt`, used ONLY by tests/golden/make_golden.py in the build
container to run the reference's pure functions (the real unyt is not installed and there is
no network).  Every quantity is a bare number in ONE fixed unit system (Mpc, 1e10 Msun,
km/s-like code velocity), so unit conversions are the identity and `.to()` is a no-op: the
numerical operations the reference performs on the underlying arrays are untouched.
This is test tooling of this repository, not part of the product and not a copy of unyt.
"""
import numpy as np

# values of named units / constants in the golden unit system
_UNIT_VALUES = {"newton_G": 43.00917552, "Mpc": 1.0, "kpc": 1.0e-3, "dimensionless": 1.0}


class Unit(float):
    registry = None

    def __new__(cls, name=1.0, registry=None, **kw):
        if isinstance(name, str):
            val = _UNIT_VALUES.get(name, 1.0)
        else:
            val = float(name)
        obj = float.__new__(cls, val)
        return obj

    @property
    def units(self):
        return Unit(1.0)

    # arithmetic between units stays a Unit (so `.registry` keeps working)
    def __mul__(self, o):
        r = float.__mul__(self, o) if isinstance(o, (int, float)) else NotImplemented
        return Unit(r) if r is not NotImplemented and not isinstance(o, np.ndarray) else np.multiply(float(self), o)

    __rmul__ = __mul__

    def __truediv__(self, o):
        if isinstance(o, np.ndarray):
            return np.divide(float(self), o)
        return Unit(float(self) / float(o))

    def __rtruediv__(self, o):
        if isinstance(o, np.ndarray):
            return np.divide(o, float(self))
        return Unit(float(o) / float(self))

    def __pow__(self, p):
        return Unit(float(self) ** p)

    def to(self, *a, **k):
        # number * unit is a quantity in unyt; here it is still this float: converting it hands back a quantity
        return unyt_array(float(self))


class unyt_array(np.ndarray):
    def __new__(cls, value, units=None, dtype=None, registry=None, **kw):
        return np.array(value, dtype=dtype).view(cls)

    def __array_function__(self, func, types, args, kwargs):
        # run NumPy's own implementation on the bare arrays, hand back unit-less "quantities"
        def strip(x):
            if isinstance(x, unyt_array):
                return np.asarray(x)
            if isinstance(x, (list, tuple)):
                return type(x)(strip(y) for y in x)
            return x

        res = func(*strip(args), **{k: strip(v) for k, v in kwargs.items()})
        if isinstance(res, np.ndarray):
            return res.view(unyt_array)
        if isinstance(res, tuple):
            return tuple(r.view(unyt_array) if isinstance(r, np.ndarray) else r for r in res)
        if isinstance(res, np.floating):
            # scalar results of functions applied to quantities stay quantities (np.linalg.norm(L).units)
            return np.asarray(res).view(unyt_array)
        return res

    @property
    def units(self):
        return Unit(1.0)

    @property
    def value(self):
        return np.asarray(self)

    v = value

    def __getitem__(self, item):
        # like unyt: one element of a quantity array is still a quantity (it has .to(), .units, ...)
        res = super().__getitem__(item)
        if isinstance(res, np.generic):
            return np.asarray(res).view(unyt_array)
        return res

    def to(self, *a, **k):
        return self

    def to_value(self, *a, **k):
        return np.asarray(self)

    def convert_to_units(self, *a, **k):
        return None

    def in_units(self, *a, **k):
        return self


unyt_quantity = unyt_array
Mpc = Unit("Mpc")
kpc = Unit("kpc")
dimensionless = Unit(1.0)
Msun = Unit(1.0)


class UnitRegistry:
    def __init__(self, *a, **k):
        pass

    def add(self, *a, **k):
        pass


from . import dimensions  # noqa: E402,F401


def __getattr__(name):
    # any other unit / constant symbol (unyt.c, unyt.sigma_thompson, ...) is the number 1
    if name.startswith("__"):
        raise AttributeError(name)
    return Unit(1.0)
