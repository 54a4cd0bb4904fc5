"""
Unit-AWARE stand-in for `unyt`, used ONLY by tests/golden/make_control_flow_golden.py (variant "units") in the build
container: the real unyt is not installed and there is no network.  Quantities carry a unit = (factor to the base
system, powers of length / mass / time / temperature); `a` is a unit whose factor is the expansion factor, as in the
reference's registry, so `snap_length * a` (comoving) and `snap_length` (physical) differ by exactly that factor and
every `.to()`, `+=`, comparison or `unyt_array(..., units=...)` of the reference converts the way unyt would: by the
ratio of the two factors.  What this stand-in CANNOT reproduce is unyt's own rounding of a conversion factor (a ratio
of base values built by sympy): vectors made with it pin which conversions happen and with which powers of `a`, to a
few ulp -- not to the last bit.  Base system: snap_length, snap_mass, snap_time of the golden snapshot.
This is test tooling of this repository, not part of the product and not a copy of unyt.
"""
import re

import numpy as np

_REG = {}


def define_golden_system(a, kpc_in_length, G_value):
    """(re)define the registry: snapshot units are the base; a = expansion factor; kpc, Mpc, newton_G in base units."""
    _REG.clear()
    Z = (0, 0, 0, 0)
    _REG.update({
        "dimensionless": Unit(1.0, Z), "snap_length": Unit(1.0, (1, 0, 0, 0)), "snap_mass": Unit(1.0, (0, 1, 0, 0)),
        "snap_time": Unit(1.0, (0, 0, 1, 0)), "snap_temperature": Unit(1.0, (0, 0, 0, 1)), "K": Unit(1.0, (0, 0, 0, 1)),
        "a": Unit(float(a), Z), "h": Unit(1.0, Z),
        "kpc": Unit(float(kpc_in_length), (1, 0, 0, 0)), "Mpc": Unit(1000.0 * float(kpc_in_length), (1, 0, 0, 0)),
        "swift_mpc": Unit(1000.0 * float(kpc_in_length), (1, 0, 0, 0)),
        "newton_G": Unit(float(G_value), (3, -1, -2, 0)),
        "code_time": Unit(1.0, (0, 0, 1, 0)),
    })  # fmt: skip


class Unit:
    __array_ufunc__ = None   # ndarray * Unit: NumPy steps aside and Unit.__rmul__ makes the quantity

    def __init__(self, factor=1.0, dims=(0, 0, 0, 0), registry=None, **kw):
        if isinstance(factor, str):
            u = _parse(factor)
            factor, dims = u.factor, u.dims
        elif isinstance(factor, Unit):
            factor, dims = factor.factor, factor.dims
        self.factor, self.dims = float(factor), tuple(dims)
        self.registry = registry

    # ---- algebra
    def __mul__(self, o):
        if isinstance(o, Unit):
            return Unit(self.factor * o.factor, tuple(x + y for x, y in zip(self.dims, o.dims)))
        if isinstance(o, unyt_array):   # quantity * unit: the units multiply
            out = np.array(o.view(np.ndarray)).view(unyt_array)
            out.units = o.units * self
            return out
        return unyt_array(o, self)   # number * unit = quantity

    __rmul__ = __mul__

    def __truediv__(self, o):
        if isinstance(o, Unit):
            return Unit(self.factor / o.factor, tuple(x - y for x, y in zip(self.dims, o.dims)))
        if isinstance(o, unyt_array):
            out = np.asarray(1.0 / o.view(np.ndarray)).view(unyt_array)
            out.units = self / o.units
            return out
        return unyt_array(1.0 / np.asarray(o, dtype=float), self)

    def __rtruediv__(self, o):
        inv = Unit(1.0 / self.factor, tuple(-x for x in self.dims))
        if isinstance(o, unyt_array):
            out = np.array(o.view(np.ndarray)).view(unyt_array)
            out.units = o.units * inv
            return out
        return unyt_array(o, inv)

    def __pow__(self, p):
        return Unit(self.factor**p, tuple(x * p for x in self.dims))

    def __eq__(self, o):
        return isinstance(o, Unit) and self.dims == o.dims and self.factor == o.factor

    def __hash__(self):
        return hash((self.factor, self.dims))

    def same_dimensions_as(self, o):
        return all(abs(x - y) < 1e-12 for x, y in zip(self.dims, o.dims))

    @property
    def units(self):
        return self

    @property
    def is_dimensionless(self):
        return all(abs(x) < 1e-12 for x in self.dims)

    def __repr__(self):
        return f"Unit({self.factor!r}, {self.dims})"


def _parse(expr):
    """unit expression: names, numbers, * / ** and parentheses (what the property table and the classes use)"""
    expr = expr.strip()
    if expr in ("", "1"):
        return _REG["dimensionless"]
    tokens = re.findall(r"\*\*|[*/()]|[A-Za-z_][A-Za-z_0-9]*|[0-9.]+(?:[eE][-+]?[0-9]+)?|-", expr)
    pos = [0]

    def peek():
        return tokens[pos[0]] if pos[0] < len(tokens) else None

    def take():
        pos[0] += 1
        return tokens[pos[0] - 1]

    def atom():
        t = take()
        if t == "(":
            u = term()
            assert take() == ")"
        elif re.match(r"[0-9.]", t):
            u = Unit(float(t))
        else:
            if t not in _REG:
                raise KeyError(f"unit {t!r} is not in the golden registry")
            u = _REG[t]
        if peek() == "**":
            take()
            sign = -1.0 if peek() == "-" else 1.0
            if peek() == "-":
                take()
            t2 = take()
            if t2 == "(":
                inner = []
                while peek() != ")":
                    inner.append(take())
                take()
                p = sign * eval("".join(inner))   # noqa: S307 (digits and / only: fractional powers like (1/2))
            else:
                p = sign * float(t2)
            u = u**p
        return u

    def term():
        u = atom()
        while peek() in ("*", "/"):
            op = take()
            v = atom()
            u = u * v if op == "*" else u / v
        return u

    u = term()
    assert pos[0] == len(tokens), (expr, tokens)
    return u


def _as_unit(u):
    if u is None:
        return _REG["dimensionless"]
    if isinstance(u, Unit):
        return u
    if isinstance(u, str):
        return _parse(u)
    if isinstance(u, unyt_array):
        return u.units
    raise TypeError(f"cannot make a unit of {u!r}")


_SAME_UNITS_UFUNCS = {np.add, np.subtract, np.maximum, np.minimum, np.fmax, np.fmin, np.hypot}
_COMPARISONS = {np.less, np.less_equal, np.greater, np.greater_equal, np.equal, np.not_equal}
_KEEP = {np.negative, np.absolute, np.fabs, np.positive, np.floor, np.ceil, np.rint, np.trunc}
_BARE = {np.isfinite, np.isnan, np.isinf, np.sign, np.logical_not, np.logical_and, np.logical_or, np.signbit}


class unyt_array(np.ndarray):
    def __new__(cls, value, units=None, dtype=None, registry=None, **kw):
        if isinstance(value, unyt_array) and units is None:
            units = value.units
        elif isinstance(value, unyt_array):
            value = value.to(units)
        elif isinstance(value, (list, tuple)) and any(isinstance(v, unyt_array) for v in value):
            u0 = next(v.units for v in value if isinstance(v, unyt_array))
            value = [v.to(u0).view(np.ndarray) if isinstance(v, unyt_array) else v for v in value]
            units = units or u0
        obj = np.array(np.asarray(value), dtype=dtype).view(cls)
        obj.units = _as_unit(units)
        return obj

    def __array_finalize__(self, obj):
        if not hasattr(self, "units"):
            self.units = getattr(obj, "units", None) or _REG.get("dimensionless", Unit())

    # ---- conversions
    def to(self, units=None, *a, **k):
        u = self.units if units is None else _as_unit(units)
        if not self.units.same_dimensions_as(u):
            raise ValueError(f"cannot convert {self.units} to {u}")
        f = self.units.factor / u.factor
        # unyt keeps the dtype of a float array when it converts it
        out = np.asarray(self.view(np.ndarray) * self.dtype.type(f) if self.dtype.kind == "f" else self.view(np.ndarray) * f).view(unyt_array)
        out.units = u
        return out

    in_units = to

    def to_value(self, units=None, *a, **k):
        return np.asarray(self.to(units))

    def convert_to_units(self, units, *a, **k):
        new = self.to(units)
        self[...] = new.view(np.ndarray)
        self.units = new.units

    @property
    def value(self):
        return np.array(self.view(np.ndarray))

    v = value
    d = value

    def copy(self, *a, **k):
        out = np.ndarray.copy(self, *a, **k)
        out.units = self.units
        return out

    def __getitem__(self, item):
        res = np.ndarray.__getitem__(self, item)
        if isinstance(res, np.generic):
            res = np.asarray(res).view(unyt_array)
        if isinstance(res, unyt_array):
            res.units = self.units
        return res

    def __repr__(self):
        return f"unyt({np.asarray(self)!r}, {self.units})"

    def __reduce__(self):
        return (unyt_array, (np.asarray(self), self.units))

    # ---- ufuncs
    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        def unit_of(x):
            return x.units if isinstance(x, unyt_array) else _REG["dimensionless"]

        def bare(x):
            return x.view(np.ndarray) if isinstance(x, unyt_array) else x

        ins = list(inputs)
        res_unit = None
        if method in ("reduce", "accumulate"):
            res_unit = unit_of(ins[0]) if ufunc in (np.add, np.maximum, np.minimum) else None
            if ufunc is np.multiply:
                raise NotImplementedError("product reduction of quantities")
        elif ufunc in _SAME_UNITS_UFUNCS or ufunc in _COMPARISONS:
            u0 = unit_of(ins[0]) if isinstance(ins[0], unyt_array) else unit_of(ins[1])
            conv = []
            for x in ins:
                if isinstance(x, unyt_array):
                    conv.append(bare(x.to(u0)) if x.units != u0 else bare(x))
                else:
                    # a bare number meets a quantity: only zero (any unit) or a dimensionless partner make sense
                    if not (u0.is_dimensionless or np.all(np.asarray(x) == 0) or np.all(~np.isfinite(np.asarray(x, dtype=float)))):
                        raise ValueError(f"bare number {x!r} combined with a quantity in {u0}")
                    conv.append(np.asarray(x) / u0.factor if u0.is_dimensionless and u0.factor != 1.0 else x)
            ins = conv
            res_unit = None if ufunc in _COMPARISONS else u0
        elif ufunc is np.multiply:
            res_unit = unit_of(ins[0]) * unit_of(ins[1])
        elif ufunc in (np.divide, np.true_divide, np.floor_divide):
            res_unit = unit_of(ins[0]) / unit_of(ins[1])
        elif ufunc is np.remainder or ufunc is np.fmod:
            u0 = unit_of(ins[0])
            ins = [bare(ins[0]), bare(ins[1].to(u0)) if isinstance(ins[1], unyt_array) else ins[1]]
            res_unit = u0
        elif ufunc is np.sqrt:
            res_unit = unit_of(ins[0]) ** 0.5
        elif ufunc is np.cbrt:
            res_unit = unit_of(ins[0]) ** (1.0 / 3.0)
        elif ufunc is np.square:
            res_unit = unit_of(ins[0]) ** 2
        elif ufunc is np.reciprocal:
            res_unit = unit_of(ins[0]) ** -1
        elif ufunc is np.power:
            p = float(np.asarray(bare(ins[1])).reshape(-1)[0])
            res_unit = unit_of(ins[0]) ** p
        elif ufunc.__name__ == "_ones_like":   # x ** 0
            res_unit = _REG["dimensionless"]
        elif ufunc in _KEEP:
            res_unit = unit_of(ins[0])
        elif ufunc in _BARE:
            res_unit = None
        elif ufunc in (np.exp, np.log, np.log10, np.log2, np.arccos, np.arcsin, np.arctan, np.cos, np.sin, np.tan, np.arctan2, np.tanh):
            for x in ins:
                if isinstance(x, unyt_array) and not x.units.is_dimensionless:
                    raise ValueError(f"{ufunc.__name__} of a dimensional quantity")
            ins = [bare(x.to("dimensionless")) if isinstance(x, unyt_array) else x for x in ins]
            res_unit = _REG["dimensionless"]
        else:
            raise NotImplementedError(f"ufunc {ufunc.__name__} on quantities")
        # unyt's rule for products / ratios of quantities of the SAME dimensions whose units differ by a pure factor
        # (km / m, snap_mass / (a**3 snap_mass)): the factor goes into the numbers and the result is plainly dimensionless
        post_factor = None
        if ufunc in (np.multiply, np.divide, np.true_divide) and method == "__call__" and res_unit is not None:
            u0, u1 = unit_of(inputs[0]), unit_of(inputs[1])
            if res_unit.is_dimensionless and res_unit.factor != 1.0 and not u0.is_dimensionless and u0.same_dimensions_as(u1):
                post_factor = res_unit.factor
                res_unit = _REG["dimensionless"]
        ins = [bare(x) for x in ins]
        if post_factor is not None:
            if out is not None:
                kwargs["out"] = tuple(bare(x) for x in out)
                getattr(ufunc, method)(*ins, **kwargs)
                np.multiply(bare(out[0]), post_factor, out=bare(out[0]))
                if isinstance(out[0], unyt_array):
                    out[0].units = res_unit
                return out[0]
            res = np.asarray(np.multiply(getattr(ufunc, method)(*ins, **kwargs), post_factor)).view(unyt_array)
            res.units = res_unit
            return res
        if out is not None:
            # in-place: the result is converted into the units of the output array (what `+=` on a quantity does)
            o = out[0]
            if isinstance(o, unyt_array) and res_unit is not None and ufunc in (np.multiply, np.divide, np.true_divide, np.power, np.sqrt, np.square):
                # `x *= y`, `x /= y`: the data change in place and the quantity takes the new units
                kwargs["out"] = tuple(bare(x) for x in out)
                getattr(ufunc, method)(*ins, **kwargs)
                o.units = res_unit
                return o
            if isinstance(o, unyt_array) and res_unit is not None and o.units != res_unit:
                if not o.units.same_dimensions_as(res_unit):
                    raise ValueError(f"cannot store {res_unit} into {o.units}")
                tmp = getattr(ufunc, method)(*ins, **kwargs) * (res_unit.factor / o.units.factor)
                np.copyto(bare(o), tmp, casting="same_kind")
                return o
            kwargs["out"] = tuple(bare(x) for x in out)
            getattr(ufunc, method)(*ins, **kwargs)
            return out[0]
        res = getattr(ufunc, method)(*ins, **kwargs)
        if res_unit is None or not isinstance(res, (np.ndarray, np.generic)):
            return res
        res = np.asarray(res).view(unyt_array)
        res.units = res_unit
        return res

    # ---- functions
    def __array_function__(self, func, types, args, kwargs):
        def first_unit(xs):
            for x in xs:
                if isinstance(x, unyt_array):
                    return x.units
                if isinstance(x, (list, tuple)):
                    u = first_unit(x)
                    if u is not None:
                        return u
            return None

        def strip(x, u=None):
            if isinstance(x, unyt_array):
                return (x.to(u) if (u is not None and x.units != u) else x).view(np.ndarray)
            if isinstance(x, (list, tuple)):
                return type(x)(strip(y, u) for y in x)
            return x

        name = func.__name__
        u0 = first_unit(args)
        keep = {"sum", "cumsum", "max", "min", "amax", "amin", "median", "mean", "concatenate", "stack", "clip", "copy", "diag",
                "zeros_like", "ones_like", "empty_like", "sort", "abs", "where", "atleast_1d", "squeeze", "ravel", "reshape",
                "take", "nan_to_num", "transpose", "vstack", "hstack", "full_like", "ptp", "trace", "around", "round", "partition"}
        bare_out = {"argsort", "argmax", "argmin", "any", "all", "isclose", "allclose", "array_equal", "searchsorted", "nonzero",
                    "count_nonzero", "shape", "size", "ndim", "bincount", "flatnonzero", "lexsort", "unique", "result_type", "can_cast",
                    "may_share_memory", "shares_memory", "iscomplexobj", "isrealobj", "array_repr", "array_str", "array2string"}
        if name in keep:
            same = name in ("concatenate", "stack", "clip", "where", "vstack", "hstack")
            res = func(*strip(args, u0 if same else None), **{k: strip(v, u0 if same else None) for k, v in kwargs.items()})
            unit = u0
        elif name in bare_out:
            same = name in ("isclose", "allclose", "array_equal", "searchsorted")
            return func(*strip(args, u0 if same else None), **{k: strip(v) for k, v in kwargs.items()})
        elif name in ("cross", "dot", "matmul", "inner", "outer"):
            a, b = args[0], args[1]
            ua = a.units if isinstance(a, unyt_array) else _REG["dimensionless"]
            ub = b.units if isinstance(b, unyt_array) else _REG["dimensionless"]
            res = func(strip(a), strip(b), *args[2:], **kwargs)
            unit = ua * ub
        elif name == "norm":
            res = func(*strip(args), **kwargs)
            unit = u0
        elif name in ("eigh", "eig", "eigvalsh", "inv", "det"):
            return func(*strip(args), **kwargs)
        elif name in ("copyto",):
            dst, src = args[0], args[1]
            func(strip(dst), strip(src, dst.units if isinstance(dst, unyt_array) else None), *args[2:], **kwargs)
            return None
        else:
            raise NotImplementedError(f"numpy.{name} on quantities (tests/golden/_shims_units)")
        if isinstance(res, np.ndarray):
            res = res.view(unyt_array)
            res.units = unit
            return res
        if isinstance(res, np.generic):
            res = np.asarray(res).view(unyt_array)
            res.units = unit
            return res
        if isinstance(res, tuple):
            return tuple((r.view(unyt_array) if isinstance(r, np.ndarray) else r) for r in res)
        return res


class unyt_quantity(unyt_array):
    def __new__(cls, value, units=None, dtype=None, registry=None, **kw):
        return unyt_array.__new__(unyt_array, value, units, dtype, registry)


class UnitRegistry:
    def __init__(self, *a, **k):
        pass

    def add(self, *a, **k):
        pass


from . import dimensions  # noqa: E402,F401


def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)
    if name in _REG:
        return _REG[name]
    raise AttributeError(f"unit {name!r} is not in the golden registry (tests/golden/_shims_units)")
