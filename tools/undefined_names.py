"""Poor man's pyflakes (none is installed here): names a module loads at function or module level that nothing in the module binds -- what a mechanical split of a
big file leaves behind.   python tools/undefined_names.py bench.py bench_blocks/*.py"""
import ast
import builtins
import sys


def bound_names(tree):
    out = set(dir(builtins)) | {"__file__", "__name__"}
    star = []
    for n in ast.walk(tree):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            out.add(n.name)
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            a = n.args
            for x in a.args + a.kwonlyargs + a.posonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                out.add(x.arg)
        elif isinstance(n, ast.Import):
            for al in n.names:
                out.add((al.asname or al.name).split(".")[0])
        elif isinstance(n, ast.ImportFrom):
            for al in n.names:
                if al.name == "*":
                    star.append((n.module, n.level))
                else:
                    out.add(al.asname or al.name)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
        elif isinstance(n, ast.arg):
            out.add(n.arg)
    return out, star


def main():
    bad = 0
    for path in sys.argv[1:]:
        tree = ast.parse(open(path).read(), path)
        names, star = bound_names(tree)
        for mod, level in star:  # `from .common import *`: whatever that module binds at top level
            import os
            base = os.path.dirname(path) if level else "."
            p = os.path.join(base, *(mod or "").split(".")) + ".py"
            if os.path.exists(p):
                t2 = ast.parse(open(p).read(), p)
                n2, _ = bound_names(t2)
                names |= n2
        for n in ast.walk(tree):
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in names:
                print("%s:%d: undefined name %s" % (path, n.lineno, n.id))
                bad += 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
