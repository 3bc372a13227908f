"""Load the reference's own functions for this path, by file path, in the BUILD CONTAINER ONLY.

TEST INFRASTRUCTURE.  Used by ``tests/golden/gen_golden.py`` to freeze reference outputs into
``tests/golden/*.npz`` (``tests/test_oracle_golden.py`` then checks the oracle against those fixtures bit for bit,
here and on the GPU box, where ``/root/reference`` does not exist).  Nothing from the reference tree is
copied into this repository: the functions are executed from where they lie.

How:
  * ``s2_emit/synth.py`` imports rasterio at module level but the two hot functions never touch
    it -> register an empty stand-in module object for the *import statement only* and load the
    file with importlib.
  * ``s2_emit/color.py`` imports POT (``ot``) at module level -> same; the OT function itself is
    never called (POT absent => Sinkhorn parity unpinned).
  * ``s2_emit/poly_regression.py`` is a script (module level code with /content paths); its two
    top-level function definitions (lines 16-62 and 65-84) are selected from the parsed AST and
    executed with ``np`` in scope.
  * notebook helpers are executed from the cell source stored in the .ipynb JSON.
  * ``tiles_helpers/utils.py:save_tile_pair`` does rasterio I/O around seven statements of array
    arithmetic (lines 362-374, the uint16 quantisation of an EMIT tile); those statements are
    selected from the function's AST by the names they assign and executed on in-memory arrays.
"""
from __future__ import annotations

import ast
import importlib.util
import json
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get("HSR_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "s2_emit", "synth.py"))


def _ensure_placeholder(name: str, **attrs):
    if name in sys.modules:
        return
    try:
        importlib.import_module(name)
        return
    except Exception:
        pass
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__hsr_placeholder__ = True
    sys.modules[name] = m


def _load_file(modname: str, relpath: str):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REFERENCE_ROOT, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_synth():
    _ensure_placeholder("rasterio")
    _ensure_placeholder("rasterio.windows", from_bounds=None, transform=None)
    return _load_file("_hsr_ref_synth", "s2_emit/synth.py")


def load_color():
    _ensure_placeholder("ot")
    return _load_file("_hsr_ref_color", "s2_emit/color.py")


def load_poly_functions():
    """Returns dict with the reference's fit_ot_poly_rgb / apply_poly_rgb function objects."""
    path = os.path.join(REFERENCE_ROOT, "s2_emit", "poly_regression.py")
    tree = ast.parse(open(path).read(), filename=path)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef)]
    ns = {"np": np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return {k: v for k, v in ns.items() if callable(v) and k in ("fit_ot_poly_rgb", "apply_poly_rgb")}


def _notebook_functions(relpath: str, names):
    nb = json.load(open(os.path.join(REFERENCE_ROOT, relpath)))
    ns = {"np": np}
    for cell in nb["cells"]:
        if cell["cell_type"] != "code":
            continue
        src = "".join(cell["source"])
        try:
            tree = ast.parse(src)
        except SyntaxError:      # cells with shell magics
            continue
        defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
        if defs:
            exec(compile(ast.Module(body=defs, type_ignores=[]), relpath, "exec"), ns)
    return {k: ns[k] for k in names if k in ns}


def load_pairs_notebook_functions():
    return _notebook_functions("Pairs_EMIT_S2_demo-2.ipynb", ["calibrate_pseudo_to_real_linear"])


def load_spectral_matching_functions():
    return _notebook_functions(
        "legacy_notebooks/Spectral_matching.ipynb",
        ["subsample_bands_evenly", "flatten_pixels", "logit", "sigmoid", "predict_cube_logit"],
    )


def load_tile_quantiser():
    """The uint16 tile quantisation of tiles_helpers/utils.py:362-374 as a callable
    ``q(emit_tile, src_nodata=None, emit_scale=10000.0, emit_nodata_u16=65535) -> uint16 array``.
    The statements run from the reference file itself (none of its text is stored here)."""
    path = os.path.join(REFERENCE_ROOT, "tiles_helpers", "utils.py")
    tree = ast.parse(open(path).read(), filename=path)
    fn = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == "save_tile_pair")
    wanted = {"emit", "valid", "scaled_i32", "emit_u16"}

    def assigned(node):
        out = set()
        tgts = node.targets if isinstance(node, ast.Assign) else [node.target] if isinstance(node, ast.AugAssign) else []
        for t in tgts:
            while isinstance(t, ast.Subscript):
                t = t.value
            if isinstance(t, ast.Name):
                out.add(t.id)
        return out

    stmts = []
    for w in (n for n in ast.walk(fn) if isinstance(n, ast.With)):
        for st in w.body:
            if isinstance(st, (ast.Assign, ast.AugAssign)) and assigned(st) and assigned(st) <= wanted:
                stmts.append(st)
            elif isinstance(st, ast.If) and all(isinstance(b, ast.AugAssign) and assigned(b) <= {"valid"} for b in st.body):
                stmts.append(st)
        if stmts:
            break
    if len(stmts) < 6:
        raise RuntimeError("save_tile_pair: quantisation statements not found (reference changed?)")
    code = compile(ast.Module(body=stmts, type_ignores=[]), path, "exec")

    def quantise(emit_tile, src_nodata=None, emit_scale=10000.0, emit_nodata_u16=65535):
        ns = {"np": np, "emit_tile": emit_tile, "emit_ds": types.SimpleNamespace(nodata=src_nodata),
              "emit_scale": emit_scale, "emit_nodata_u16": emit_nodata_u16}
        exec(code, ns)
        return ns["emit_u16"]

    return quantise
