"""A stand-in for the `gurobipy` surface run_same touches, for TIMING the reference signature on a box without a licence
(bench.py's `api_path` record).  It solves nothing: `optimize(cb)` takes the MIP start as the incumbent, calls the lazy callback once
with it and reports OPTIMAL -- the behaviour of the recording double the parity tests use (tests/fake_gurobipy.py), without its
bookkeeping: expressions are not recorded (every arithmetic operation returns the one shared expression object), so what is left is the
cost of run_same's own Python: one object per variable, its loops over pairs and triangles, the post-solve tables.  NOT a solver, never
imported by the product (run_same imports `gurobipy` by name: `install()` puts this module there, `uninstall()` restores what was)."""
import sys
import types


class _Expr:
    __slots__ = ()

    def _same(self, *_a):
        return self

    __add__ = __radd__ = __sub__ = __rsub__ = __mul__ = __rmul__ = __neg__ = __le__ = __ge__ = _same
    __eq__ = _same
    __hash__ = object.__hash__


_E = _Expr()


class Var(_Expr):
    __slots__ = ("VarName", "Start", "x")

    def __init__(self, name):
        self.VarName, self.Start, self.x = name, None, 0.0


def quicksum(it):
    for _ in it:
        pass
    return _E


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
    def __init__(self, name="", env=None):
        self.vars, self.n_constrs, self.n_lazy = [], 0, 0
        self.Params = _Params()
        self.status, self.Runtime = None, 0.0

    def addVars(self, n, vtype=None, lb=0, ub=None, name="v"):
        d = {i: Var(name) for i in range(n)}
        self.vars.extend(d.values())
        return d

    def addVar(self, vtype=None, lb=0, ub=None, name="v"):
        v = Var(name)
        self.vars.append(v)
        return v

    def addConstrs(self, gen, name=None):
        for _ in gen:
            self.n_constrs += 1

    def addConstr(self, c, name=None):
        self.n_constrs += 1
        return c

    def update(self):
        pass

    def setObjective(self, expr, sense):
        pass

    def write(self, path):
        with open(path, "w") as f:
            f.write(f"\\ timing double: {len(self.vars)} vars, {self.n_constrs} constraints\n")

    def cbGetSolution(self, vars_):
        return {k: (v.Start or 0.0) for k, v in vars_.items()} if isinstance(vars_, dict) else [v.Start or 0.0 for v in vars_]

    def cbLazy(self, c):
        self.n_lazy += 1

    def optimize(self, cb=None):
        for v in self.vars:
            v.x = float(v.Start) if v.Start is not None else 0.0
        if cb is not None:
            cb(self, GRB.Callback.MIPSOL)
        self.status = GRB.OPTIMAL


_saved = []


def install():
    m = types.ModuleType("gurobipy")
    for k, v in dict(Model=Model, GRB=GRB, quicksum=quicksum, Env=Env, Var=Var).items():
        setattr(m, k, v)
    _saved.append(sys.modules.get("gurobipy"))
    sys.modules["gurobipy"] = m
    return m


def uninstall():
    old = _saved.pop() if _saved else None
    if old is None:
        sys.modules.pop("gurobipy", None)
    else:
        sys.modules["gurobipy"] = old
