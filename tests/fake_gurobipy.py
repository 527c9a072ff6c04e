"""A tiny stand-in for the gurobipy API surface run_same touches -- TEST DOUBLE ONLY.

gurobipy is proprietary and absent from the image.  This mock lets the tests drive
same_amd.run_same end to end (model assembly -> callback -> post-solve tables) without a
solver: `optimize(cb)` takes the MIP start as the incumbent, invokes the lazy callback once
with it, and reports OPTIMAL.  It makes no attempt to solve anything."""
import sys
import types


class LinExpr:
    def __init__(self, terms=None, const=0.0):
        self.terms = dict(terms or {})
        self.const = const

    def _add(self, other, sign=1.0):
        out = LinExpr(self.terms, self.const)
        if isinstance(other, Var):
            out.terms[other] = out.terms.get(other, 0.0) + sign
        elif isinstance(other, LinExpr):
            for v, c in other.terms.items():
                out.terms[v] = out.terms.get(v, 0.0) + sign * c
            out.const += sign * other.const
        else:
            out.const += sign * float(other)
        return out

    def __add__(self, o): return self._add(o)
    __radd__ = __add__
    def __sub__(self, o): return self._add(o, -1.0)
    def __rsub__(self, o): return (self * -1.0)._add(o)
    def __mul__(self, k): return LinExpr({v: c * float(k) for v, c in self.terms.items()}, self.const * float(k))
    __rmul__ = __mul__
    def __le__(self, o): return TempConstr(self - o, "<=")
    def __ge__(self, o): return TempConstr(self - o, ">=")
    def __eq__(self, o): return TempConstr(self - o, "==")
    __hash__ = object.__hash__

    def __neg__(self): return self * -1.0


class Var:
    def __init__(self, name, vtype, lb, ub):
        self.VarName, self.vtype, self.lb, self.ub = name, vtype, lb, ub
        self.Start = None
        self.x = 0.0

    def _e(self): return LinExpr({self: 1.0})
    def __neg__(self): return self._e() * -1.0
    def __add__(self, o): return self._e() + o
    __radd__ = __add__
    def __sub__(self, o): return self._e() - o
    def __rsub__(self, o): return o - self._e()
    def __mul__(self, k): return self._e() * k
    __rmul__ = __mul__
    def __le__(self, o): return self._e() <= o
    def __ge__(self, o): return self._e() >= o
    def __eq__(self, o): return self._e() == o
    __hash__ = object.__hash__


class TempConstr:
    def __init__(self, expr, sense):
        self.expr, self.sense = expr, sense


def quicksum(it):
    out = LinExpr()
    for t in it:
        out = out + t
    return out


class _Callback:
    MIPSOL = 4


class GRB:
    BINARY, CONTINUOUS, MINIMIZE = "B", "C", 1
    OPTIMAL, TIME_LIMIT = 2, 9
    METHOD_PDHG = 6
    Callback = _Callback


class _Params:
    pass


class Env:
    def __init__(self, params=None):
        self.params = params


class Model:
    last = None

    def __init__(self, name="", env=None):
        self.name, self.vars, self.constrs, self.lazy = name, [], [], []
        self.Params = _Params()
        self.objective = None
        self.status = None
        self.Runtime = 0.0
        Model.last = self

    def addVars(self, n, vtype=None, lb=0, ub=None, name="v"):
        d = {i: Var(f"{name}[{i}]", vtype, lb, ub) for i in range(n)}
        self.vars.extend(d.values())
        return d

    def addVar(self, vtype=None, lb=0, ub=None, name="v"):
        v = Var(name, vtype, lb, ub)
        self.vars.append(v)
        return v

    def addConstrs(self, gen, name=None):
        for c in gen:
            self.addConstr(c, name=name)

    def addConstr(self, c, name=None):
        assert isinstance(c, TempConstr)
        self.constrs.append((name, c))
        return c

    def update(self): pass
    def setObjective(self, expr, sense): self.objective = expr

    def write(self, path):
        with open(path, "w") as f:
            f.write(f"\\ mock model: {len(self.vars)} vars, {len(self.constrs)} constraints\n")
            for name, _ in self.constrs:
                if name is not None:
                    f.write(f" {name}\n")

    # callback API
    def cbGetSolution(self, vars_):
        return {k: (v.Start or 0.0) for k, v in vars_.items()} if isinstance(vars_, dict) else [v.Start or 0.0 for v in vars_]

    def cbLazy(self, c):
        assert isinstance(c, TempConstr)
        self.lazy.append(c)

    def optimize(self, cb=None):
        for v in self.vars:
            v.x = float(v.Start) if v.Start is not None else 0.0
        if cb is not None:
            cb(self, GRB.Callback.MIPSOL)
            cb(self, 0)  # a non-MIPSOL event must be ignored
        # cuts that were added force their q_tri to 1 in a real solve; mimic for the post-solve report
        for c in self.lazy:
            for v, coef in c.expr.terms.items():
                if v.VarName.startswith("q_tri") and coef < 0:
                    v.x = 1.0
        self.status = GRB.OPTIMAL
        self.Runtime = 0.01


def install():
    m = types.ModuleType("gurobipy")
    for k, v in dict(Model=Model, GRB=GRB, quicksum=quicksum, Env=Env, LinExpr=LinExpr, Var=Var).items():
        setattr(m, k, v)
    sys.modules["gurobipy"] = m
    return m


def canonical_constraints(constrs):
    """[(name, TempConstr)] -> list of (sense, const, ((var name, coef), ...)) with zero coefficients dropped: the
    comparable form of a model section (used to compare an assembled model with a recorded one)."""
    out = []
    for _, c in constrs:
        terms = tuple(sorted((v.VarName, float(k)) for v, k in c.expr.terms.items() if k != 0.0))
        out.append((c.sense, float(c.expr.const), terms))
    return out
