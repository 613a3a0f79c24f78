"""Import the reference's own model code read-only from /root/reference (TEST INFRASTRUCTURE).

Only usable in the build container (the GPU box has no /root/reference).  The reference
package is called ``model`` -- the same name as this repo's drop-in package -- so it is
imported with a temporarily rewritten sys.path / sys.modules and then detached: the
returned module objects keep working, and ``import model`` afterwards resolves to this
repo's package again.
"""
import importlib
import os
import sys

REFERENCE_ROOT = os.environ.get("E2FGVI_REFERENCE_ROOT", "/root/reference")
_SHIM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mmcv_shim")
_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_CACHE = {}


def available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "model", "e2fgvi.py"))


def load_reference(name="e2fgvi"):
    """Return the reference module ``model.<name>`` (e2fgvi | e2fgvi_hq)."""
    if name in _CACHE:
        return _CACHE[name]
    if not available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    if _REPO not in sys.path:
        sys.path.insert(0, _REPO)
    import oracle.dcn  # noqa: F401  (the shim imports it; load it while the repo root is on sys.path)
    saved = {k: v for k, v in sys.modules.items() if k == "model" or k.startswith("model.")}
    saved_mmcv = {k: v for k, v in sys.modules.items() if k == "mmcv" or k.startswith("mmcv.")}
    for k in list(saved) + list(saved_mmcv):
        del sys.modules[k]
    # The reference's ``model`` directory is a namespace package (no __init__.py); this repo's
    # ``model`` is a regular package and would win regardless of order, so the repo root (and the
    # implicit cwd entry) must be off sys.path while the reference is imported.
    saved_path = list(sys.path)
    here = {os.path.realpath(_REPO), os.path.realpath(os.getcwd())}
    sys.path[:] = [_SHIM, REFERENCE_ROOT] + [q for q in saved_path
                                             if q not in ("", ".") and os.path.realpath(q) not in here]
    try:
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            mod = importlib.import_module("model." + name)
        ref_mods = {k: v for k, v in sys.modules.items() if k == "model" or k.startswith("model.")}
        assert os.path.realpath(mod.__file__).startswith(os.path.realpath(REFERENCE_ROOT)), mod.__file__
    finally:
        sys.path[:] = saved_path
        for k in [k for k in sys.modules if k == "model" or k.startswith("model.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    mod.__ref_modules__ = ref_mods
    _CACHE[name] = mod
    return mod


def build_reference_model(name="e2fgvi", state_dict=None):
    """Construct the reference InpaintGenerator (eval mode, CPU) and load ``state_dict``."""
    import io, contextlib
    mod = load_reference(name)
    with contextlib.redirect_stdout(io.StringIO()):
        net = mod.InpaintGenerator()
    if state_dict is not None:
        net.load_state_dict(state_dict, strict=True)
    return net.eval()
