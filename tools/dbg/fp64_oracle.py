"""A float64 build of oracle/raster_oracle.c (same source, `float` -> `double`, under /tmp): the yardstick that says how far plain fp32
arithmetic - the oracle's own included - lands from the exact value in a badly conditioned scene.  Used by fp64_check.py,
clip_depth_one.py and fuzz_raster.py (clip mode); CPU only."""
import ctypes, os, re, subprocess
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DP = ctypes.POINTER(ctypes.c_double); IP = ctypes.POINTER(ctypes.c_int32)
_lib = None


def load():
    global _lib
    if _lib is None:
        src = open(os.path.join(REPO, "oracle", "raster_oracle.c")).read()
        src = re.sub(r"\bfloat\b", "double", src)
        for fn in ("fmaxf", "fminf", "fabsf", "expf", "sqrtf", "floorf", "ceilf"):
            src = re.sub(r"\b" + fn + r"\b", fn[:-1], src)
        src = re.sub(r"(\d)f\b", r"\1", src)  # 1e-8f -> 1e-8
        d = f"/tmp/fp64_{os.getuid()}"
        os.makedirs(d, exist_ok=True)
        open(f"{d}/raster_oracle64.c", "w").write(src)
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", f"{d}/lib64.so", f"{d}/raster_oracle64.c", "-lm"])
        _lib = ctypes.CDLL(f"{d}/lib64.so")
        _lib.oracle_silhouette_forward.argtypes = [DP, IP] + [ctypes.c_int] * 4 + [ctypes.c_double] * 2 + [ctypes.c_int, DP, IP, IP, DP, DP]
        _lib.oracle_silhouette_backward.argtypes = [DP, IP] + [ctypes.c_int] * 4 + [ctypes.c_double] * 2 + [ctypes.c_int, DP, DP]
        _lib.oracle_set_z_clip.argtypes = [ctypes.c_double]
        _lib.oracle_set_select_mode(1)
    return _lib


def depth_gradient(render_ref, ndc_n, faces, S, K, grad_sil_n, select_mode=1):
    """(V,) float64: the depth channel of ONE image - the clipped mesh exactly as the fp32 oracle builds it (fp32 crossing points),
    its backward pass in double, `render_ref.clip_depth_gradient` on the new vertices' float64 xy gradients."""
    lib = load()
    lib.oracle_set_select_mode(int(select_mode))
    V = ndc_n.shape[0]
    z = np.zeros(V)
    plan = render_ref._clip_plan(np.ascontiguousarray(ndc_n[None], np.float32), faces.astype(np.int32))[0]
    if plan is None:
        return z
    va, fa, src, _ = plan
    va64 = np.ascontiguousarray(va, np.float64); fa32 = np.ascontiguousarray(fa, np.int32)
    gs64 = np.ascontiguousarray(grad_sil_n[None], np.float64)
    g64 = np.empty((1, va.shape[0], 3))
    rc = lib.oracle_silhouette_backward(va64.ctypes.data_as(DP), fa32.ctypes.data_as(IP), 1, va.shape[0], fa.shape[0], S,
                                        render_ref.BLUR_RADIUS, render_ref.SIGMA, K, gs64.ctypes.data_as(DP), g64.ctypes.data_as(DP))
    assert rc == 0
    for j in range(len(src)):
        a, b = int(src[j, 0]), int(src[j, 1])
        dza, dzb = render_ref.clip_depth_gradient(va[a], va[b], g64[0, V + j, :2], render_ref._Z_CLIP)
        z[a] += dza; z[b] += dzb
    return z
