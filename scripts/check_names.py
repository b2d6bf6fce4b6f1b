"""Poor man's pyflakes (no linter is installed offline): names loaded in a module that are neither bound in an enclosing
function scope, module globals, nor builtins.   python scripts/check_names.py file.py [...]"""
import ast
import builtins
import sys

SCOPES = (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda, ast.ClassDef)


def bound_in(nodes):
    """Names bound directly in a scope whose statements are `nodes` (nested scopes contribute only their own name)."""
    names = set()

    def walk(n):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            names.add(n.name)
            return
        if isinstance(n, ast.Lambda):
            return
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            names.add(n.id)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for a in n.names:
                names.add((a.asname or a.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            names.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            names.update(n.names)
        for c in ast.iter_child_nodes(n):
            walk(c)
    for n in nodes:
        walk(n)
    return names


def check(path):
    tree = ast.parse(open(path).read())
    bad = []

    def scope(nodes, visible):
        visible = visible | bound_in(nodes)

        def walk(n):
            if isinstance(n, SCOPES):
                inner = set()
                if not isinstance(n, ast.ClassDef):
                    a = n.args
                    for x in a.args + a.kwonlyargs + a.posonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                        inner.add(x.arg)
                    for d in a.defaults + [k for k in a.kw_defaults if k is not None]:
                        walk(d)
                body = n.body if isinstance(n.body, list) else [n.body]
                for d in getattr(n, "decorator_list", []):
                    walk(d)
                scope(body, visible | inner)
                return
            if isinstance(n, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
                inner = set()
                for g in n.generators:
                    inner |= bound_in([g.target])
                scope(list(ast.iter_child_nodes(n)), visible | inner)
                return
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in visible:
                bad.append((n.lineno, n.id))
            for c in ast.iter_child_nodes(n):
                walk(c)
        for n in nodes:
            walk(n)
    scope(tree.body, set(dir(builtins)) | {"__file__", "__name__", "__doc__"})
    return sorted(set(bad))


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        for line, name in check(p):
            print(f"{p}:{line}: undefined name {name}")
            rc = 1
    sys.exit(rc)
