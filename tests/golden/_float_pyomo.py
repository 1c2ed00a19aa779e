"""Float-valued stand-ins for the handful of Pyomo classes the reference's FTE model text uses
(ConcreteModel, RangeSet, Param, Var, Constraint, Objective).  Build-container tooling for
make_golden.py: with these in the exec namespace, the reference's OWN model-construction lines
(src/all_optimizations.py:283-500) run on plain floats, so its constraints become residuals and its
objective a number.  Written for this repo (Pyomo is absent); it contains no reference code.
"""
import itertools
import math


def _val(a):
    if isinstance(a, VarData):
        return a._v()
    return float(a)


class Residual:
    """lhs == rhs evaluated on floats: keeps lhs - rhs."""
    def __init__(self, r):
        self.r = float(r)


class Ineq:
    """lhs <= rhs evaluated on floats."""
    def __init__(self, lhs, rhs):
        self.lhs, self.rhs = float(lhs), float(rhs)


class _Num:
    __array_ufunc__ = None          # numpy scalars defer to our reflected operators

    def _v(self):
        raise NotImplementedError

    def __add__(self, o): return F(self._v() + _val(o))
    def __radd__(self, o): return F(_val(o) + self._v())
    def __sub__(self, o): return F(self._v() - _val(o))
    def __rsub__(self, o): return F(_val(o) - self._v())
    def __mul__(self, o): return F(self._v() * _val(o))
    def __rmul__(self, o): return F(_val(o) * self._v())
    def __truediv__(self, o): return F(self._v() / _val(o))
    def __rtruediv__(self, o): return F(_val(o) / self._v())
    def __pow__(self, o): return F(self._v() ** _val(o))
    def __rpow__(self, o): return F(_val(o) ** self._v())
    def __neg__(self): return F(-self._v())
    def __pos__(self): return F(self._v())
    def __abs__(self): return F(abs(self._v()))
    def __eq__(self, o): return Residual(self._v() - _val(o))
    def __le__(self, o): return Ineq(self._v(), _val(o))
    def __ge__(self, o): return Ineq(_val(o), self._v())
    def __float__(self): return float(self._v())
    __hash__ = None


class F(_Num, float):
    def _v(self):
        return float.__float__(self)

    def __float__(self):
        return float.__float__(self)

    __hash__ = float.__hash__


class VarData(_Num):
    def __init__(self, value=None):
        self.value = value
        self.reads = 0

    def _v(self):
        self.reads += 1
        if READ_LOG is not None:
            READ_LOG.append(self)
        return 0.0 if self.value is None else self.value      # Pyomo builds symbolically; an unset Var reads as 0 here

    __hash__ = object.__hash__


READ_LOG = None          # set to a list to record every VarData that is read (ConstraintList uses it)
Any = object()           # the `within=Any` domain marker of src/build.py's Params


def sin(a): return F(math.sin(_val(a)))
def cos(a): return F(math.cos(_val(a)))
def atan(a): return F(math.atan(_val(a)))


def RangeSet(n):
    return range(1, int(n) + 1)


class _Component:
    def construct(self, model):
        pass


class Param(_Component):
    def __init__(self, *sets, initialize=None, mutable=False, within=None):
        self.sets, self.init, self.data = sets, initialize, {}

    def construct(self, model):
        for idx in itertools.product(*self.sets):
            self.data[idx] = self.init(model, *idx) if callable(self.init) else self.init

    def __getitem__(self, idx):
        return self.data[idx if isinstance(idx, tuple) else (idx,)]


class Var(_Component):
    def __init__(self, *sets, initialize=None):
        self.sets, self.init, self.data = sets, initialize, {}

    def construct(self, model):
        for idx in itertools.product(*self.sets):
            self.data[idx] = VarData(self.init)

    def __getitem__(self, idx):
        return self.data[idx if isinstance(idx, tuple) else (idx,)]


class Constraint(_Component):
    Skip = "skip"

    def __init__(self, *sets, rule=None):
        self.sets, self.rule, self.data = sets, rule, {}

    def construct(self, model):
        self.evaluate(model)

    def evaluate(self, model):
        self.data = {idx: self.rule(model, *idx) for idx in itertools.product(*self.sets)}
        return self.data


class ConstraintList(_Component):
    """src/build.py:270-273 adds already-evaluated inequalities one by one: each entry keeps the inequality and the
    variables that were read while its expression was formed (with READ_LOG enabled)."""
    def __init__(self):
        self.entries = []

    def add(self, expr=None):
        global READ_LOG
        touched = tuple(READ_LOG) if READ_LOG is not None else ()
        if READ_LOG is not None:
            del READ_LOG[:]
        self.entries.append((expr, touched))


class Objective(_Component):
    def __init__(self, rule=None):
        self.rule = rule

    def value(self, model):
        return float(self.rule(model))


class ConcreteModel:
    def __init__(self, name=None):
        object.__setattr__(self, "_components", {})

    def __setattr__(self, k, v):
        object.__setattr__(self, k, v)
        if isinstance(v, _Component):
            self._components[k] = v
            v.construct(self)
