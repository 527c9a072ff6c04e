"""Poor man's pyflakes (none is installed here): names that are read somewhere in a module but bound nowhere in it.
Deliberately coarse -- every name bound ANYWHERE in the file (any scope) or a builtin counts as defined -- so it has no
false positives on this code base and still catches a forgotten import, which a CPU-only box cannot catch by running
GPU paths.  Usage: python tools/undefined_names.py file.py ..."""
import ast, builtins, sys

bad = 0
for path in sys.argv[1:]:
    tree = ast.parse(open(path).read(), path)
    bound = set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            bound.add(n.name)
            if not isinstance(n, ast.ClassDef):
                a = n.args
                for arg in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                    bound.add(arg.arg)
        elif isinstance(n, ast.Lambda):
            a = n.args
            for arg in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                bound.add(arg.arg)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for al in n.names:
                bound.add((al.asname or al.name).split(".")[0])
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            bound.add(n.id)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            bound.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            bound.update(n.names)
    for n in ast.walk(tree):
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in bound:
            print(f"{path}:{n.lineno}: name {n.id!r} is never bound in this file")
            bad += 1
sys.exit(1 if bad else 0)
