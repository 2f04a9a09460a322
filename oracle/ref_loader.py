"""TEST INFRASTRUCTURE ONLY -- loads the *real* reference implementation for fixture generation.

This module works only in the build container, where the upstream reference is mounted
read-only at /root/reference.  It never ships reference code: the reference's Python 2
sources are read as text at run time, translated in memory (lib2to3 + an AST pass that
turns integer `/` into `//`) and exec'd into throw-away module objects.  Nothing is written
to disk.  On the GPU box /root/reference does not exist and `load()` raises.

Used by tests/golden/make_golden.py (fixture generator) and by the optional
`tests/test_oracle_vs_reference.py` (skipped when the reference is absent).

Recipe: SURVEY.md section 8(c).
"""
import ast
import os
import sys
import types

REF_DIR = os.environ.get("FOURQ_REFERENCE_DIR", "/root/reference/impl")


def available():
    return os.path.isfile(os.path.join(REF_DIR, "curve4q.py"))


def _translate(path):
    from lib2to3 import refactor

    with open(path, "r") as fh:
        src = fh.read()
    if not src.endswith("\n"):
        src += "\n"
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    tree = ast.parse(str(tool.refactor_string(src, path)), filename=path)

    class _IntDiv(ast.NodeTransformer):
        # every `/` on the hot path has int operands (curve4q.py:222,224-226)
        def visit_BinOp(self, node):
            self.generic_visit(node)
            if isinstance(node.op, ast.Div):
                node.op = ast.FloorDiv()
            return node

    tree = ast.fix_missing_locations(_IntDiv().visit(tree))
    return compile(tree, path, "exec")


_cache = {}


def load():
    """Return (fields, curve4q) reference modules.  Raises RuntimeError when absent."""
    if "mods" in _cache:
        return _cache["mods"]
    if not available():
        raise RuntimeError("reference not mounted at %s" % REF_DIR)
    saved = {k: sys.modules.get(k) for k in ("fields", "test", "curve4q")}
    try:
        mods = {}
        for name in ("fields", "test", "curve4q"):
            mod = types.ModuleType(name)
            mod.__file__ = os.path.join(REF_DIR, name + ".py")
            sys.modules[name] = mod
            exec(_translate(mod.__file__), mod.__dict__)
            mods[name] = mod
        mods["curve4q"].test = mods["test"]
        mods["fields"].test = mods["test"]
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    _cache["mods"] = (mods["fields"], mods["curve4q"])
    return _cache["mods"]


if __name__ == "__main__":
    f, c = load()
    for fn in ("test_definitions", "test_reps", "test_core", "test_endo", "test_recoding"):
        getattr(c, fn)()
