"""Name resolution of the code that only ever EXECUTES on a GPU box (ops.HipOps, the torch binding, bench.py,
__graft_entry__.smoke): every global a function loads must exist in its module.  A typo there would pass every CPU test
and fail the first GPU call; no linter is installed in the image, so this is a forty-line one."""
import ast
import builtins
import importlib
import importlib.util
import os

import pytest

from conftest import ROOT

FILES = ["tooncrafter_amd/ops.py", "tooncrafter_amd/torch_ops.py", "tooncrafter_amd/_lib.py", "tooncrafter_amd/dist.py",
         "tooncrafter_amd/output.py", "bench.py", "__graft_entry__.py", "scripts/conv_halo_bench.py", "scripts/conv_halo_debug.py"]


def _module(path):
    rel = os.path.relpath(path, ROOT)
    if rel.startswith("tooncrafter_amd"):
        return importlib.import_module(rel[:-3].replace(os.sep, "."))
    if rel.startswith("scripts"):
        return None                         # scripts run their body on import: resolve against their own top-level names
    spec = importlib.util.spec_from_file_location("_static_" + os.path.basename(rel)[:-3], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _stores(node):
    out = set()
    for n in ast.walk(node):
        if isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        elif isinstance(n, ast.arg):
            out.add(n.arg)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            out.update((a.asname or a.name).split(".")[0] for a in n.names)
        elif isinstance(n, (ast.FunctionDef, ast.ClassDef)):
            out.add(n.name)
        elif isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
    return out


@pytest.mark.parametrize("rel", FILES)
def test_every_loaded_global_exists(rel):
    path = os.path.join(ROOT, rel)
    tree = ast.parse(open(path).read())
    mod = _module(path)
    top = _stores(tree) if mod is None else set(dir(mod)) | _stores(tree)
    missing = set()
    for fn in ast.walk(tree):
        if isinstance(fn, (ast.FunctionDef, ast.Lambda)):
            local = _stores(fn)
            for n in ast.walk(fn):
                if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in local and n.id not in top and \
                        not hasattr(builtins, n.id):
                    missing.add((getattr(fn, "name", "<lambda>"), n.id))
    # names a nested function takes from its enclosing function are "stores" of that function: collect them per file
    enclosing = _stores(tree)
    missing = {m for m in missing if m[1] not in enclosing}
    assert not missing, sorted(missing)
