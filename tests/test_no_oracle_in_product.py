"""The oracle (oracle/, oracle/_ref/) is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import, call or execute it.  AST walk of the shipped package, the build / staging scripts, and of
bench.py outside ``cpu_baseline`` (the file scripts/stage_reference.py's docstring refers to)."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "transformer-explainability_amd")
FORBIDDEN_ROOTS = ("oracle",)


def _imports(tree):
    """(module name, node) of every import statement in the tree."""
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            for a in node.names:
                yield a.name, node
        elif isinstance(node, ast.ImportFrom):
            yield ("." * node.level) + (node.module or ""), node
        elif isinstance(node, ast.Call):
            f = node.func
            name = f.id if isinstance(f, ast.Name) else f.attr if isinstance(f, ast.Attribute) else ""
            if name in ("import_module", "__import__") and node.args and isinstance(node.args[0], ast.Constant):
                yield str(node.args[0].value), node


def _touches_oracle(mod):
    head = mod.lstrip(".").split(".")[0]
    return head in FORBIDDEN_ROOTS


def _py_files(top):
    for d, _, fs in os.walk(top):
        if "__pycache__" in d:
            continue
        for f in fs:
            if f.endswith(".py"):
                yield os.path.join(d, f)


def test_package_never_imports_the_oracle():
    bad = []
    for path in list(_py_files(PKG)) + [os.path.join(ROOT, "transformer_explainability_amd.py")]:
        with open(path) as f:
            src = f.read()
        tree = ast.parse(src, path)
        for mod, node in _imports(tree):
            if _touches_oracle(mod):
                bad.append((os.path.relpath(path, ROOT), node.lineno, mod))
        # no string reference to the oracle directory either (sys.path games, spec_from_file_location, subprocess)
        for node in ast.walk(tree):
            if isinstance(node, ast.Constant) and isinstance(node.value, str) and not isinstance(getattr(node, "parent", None), ast.Expr):
                v = node.value
                if ("relprop_oracle" in v or "oracle/_ref" in v or "ref_harness" in v) and len(v) < 200:
                    bad.append((os.path.relpath(path, ROOT), node.lineno, "string: " + v))
    assert not bad, bad


def test_csrc_has_no_host_fallback():
    """The C ABI has no CPU path: no source of the library mentions the oracle or includes a host BLAS."""
    bad = []
    for d, _, fs in os.walk(os.path.join(PKG, "csrc")):
        for f in fs:
            with open(os.path.join(d, f), errors="replace") as fh:
                src = fh.read()
            for needle in ("oracle", "cblas", "openblas", "mkl.h"):
                if needle in src.lower():
                    bad.append((f, needle))
    assert not bad, bad


def test_bench_uses_the_oracle_only_in_its_cpu_baseline_leg():
    path = os.path.join(ROOT, "bench.py")
    with open(path) as f:
        tree = ast.parse(f.read(), path)
    allowed = {"cpu_baseline", "cpu_baseline_parity"}      # the leg's timing half and its parity half (the checker, checking)
    bad = []

    def visit(node, fn):
        for child in ast.iter_child_nodes(node):
            name = child.name if isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef)) else fn
            if isinstance(child, (ast.Import, ast.ImportFrom)):
                mods = [a.name for a in child.names] if isinstance(child, ast.Import) else [child.module or ""]
                for m in mods:
                    if _touches_oracle(m) and fn not in allowed:
                        bad.append((fn, child.lineno, m))
            visit(child, name)

    visit(tree, "<module>")
    assert not bad, bad
    # and the timed region (main) never calls cpu_baseline before the line's throughput is computed
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    calls = [n.lineno for n in ast.walk(main) if isinstance(n, ast.Call) and isinstance(n.func, ast.Name)
             and n.func.id in ("cpu_baseline", "cpu_baseline_parity")]
    value_line = min(n.lineno for n in ast.walk(main) if isinstance(n, ast.Assign) and any(
        isinstance(t, ast.Name) and t.id == "value" for t in n.targets))
    assert calls and min(calls) > value_line, (calls, value_line)


def test_graft_entry_uses_the_oracle_only_in_smoke_and_build_check():
    path = os.path.join(ROOT, "__graft_entry__.py")
    with open(path) as f:
        tree = ast.parse(f.read(), path)
    for fn in tree.body:
        if not isinstance(fn, ast.FunctionDef):
            for mod, node in _imports(fn):
                assert not _touches_oracle(mod), (node.lineno, mod)
            continue
        for mod, node in _imports(fn):
            if _touches_oracle(mod):
                assert fn.name in ("smoke", "build"), (fn.name, node.lineno, mod)
