"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/smilfit.h declares, and
rejects bad arguments without touching a GPU."""
import ctypes
import os
import re

from conftest import REPO


def test_library_exports_every_declared_symbol():
    from smilify_amd import _lib

    lib = _lib.load()
    header = open(os.path.join(REPO, "include", "smilfit.h")).read()
    declared = set(re.findall(r"\b(smil_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in smilfit.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    assert b"gfx950" in lib.smil_version()


def test_the_product_library_is_not_an_instrumented_build():
    """libsmilfit.so holds no experiment code: the timers / counters / cut-off and wrap experiments of tools/dbg exist only in libraries
    built by `make variant` (which define SMIL_INSTRUMENTED and say so in smil_version()), and the kernels' translation units
    contain no ablation switch at all."""
    from smilify_amd import _lib

    lib = _lib.load()
    assert b"instrumented" not in lib.smil_version(), lib.smil_version()
    assert os.path.basename(_lib.LIB_PATH) == "libsmilfit.so" or os.environ.get("SMILFIT_LIB")
    csrc = os.path.join(REPO, "smilify_amd", "csrc")
    banned = re.compile(r"\b(ABL_[A-Z0-9_]+|NDC_ABL_[A-Z0-9_]+|FWD_ABL_[A-Z0-9_]+|RASTER_EXPERIMENT|STREAM_NT|DBG_TIMERS|DBG_STATS)\b")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")) and f != "raster_hooks.h":
            hits = banned.findall(open(os.path.join(csrc, f)).read())
            assert not hits, (f, sorted(set(hits)))
    mk = open(os.path.join(csrc, "Makefile")).read()
    assert "-DSMIL_INSTRUMENTED" in mk.split("variant:")[1] and "-DSMIL_INSTRUMENTED" not in mk.split("variant:")[0]


def test_argument_validation_without_gpu():
    from smilify_amd import _lib

    lib = _lib.load()
    handle = ctypes.c_void_p()
    d = _lib.ModelDesc()
    d.V = d.F = d.J = 0
    assert lib.smil_model_create(ctypes.byref(d), ctypes.byref(handle)) == -1
    assert b"bad sizes" in lib.smil_last_error()
    assert lib.smil_model_dims(None, None) == -1
    assert lib.smil_adam_step(None, None, None, None, 0, 0.1, 0.5, 0.999, 1e-8, 1, None) == -1
    assert lib.smil_raster_workspace_bytes(None, 1, 64) == 0


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "smilify_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src, f
