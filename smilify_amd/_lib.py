"""ctypes binding of ``libsmilfit.so`` (C ABI: include/smilfit.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C smilify_amd/csrc``.  There is
NO fallback: if the shared object is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMILFIT_LIB: load another build of the same library (instrumented builds under tools/dbg); never a CPU path
LIB_PATH = os.environ.get("SMILFIT_LIB") or os.path.join(_HERE, "lib", "libsmilfit.so")

EXPORTS = [
    "smil_model_create", "smil_model_destroy", "smil_model_dims", "smil_last_error", "smil_version",
    "smil_lbs_forward", "smil_lbs_forward_project", "smil_lbs_backward", "smil_lbs_backward_ndc", "smil_lbs_backward_ndc_supported", "smil_clip_depth_backward", "smil_project", "smil_project2", "smil_project_backward", "smil_project_backward2",
    "smil_fov_reduce", "smil_fit_epilogue",
    "smil_raster_workspace_bytes", "smil_raster_stats", "smil_silhouette_forward", "smil_silhouette_backward",
    "smil_silhouette_l1_fused", "smil_prior_losses", "smil_mask_rows", "smil_joint_loss", "smil_pix_scale",
    "smil_image_abs_sum", "smil_sil_objective", "smil_window_terms", "smil_adam_step", "smil_adam_step_multi", "smil_adam_step_dev", "smil_profile_enable",
    "smil_profile_read",
]

N_OBJS = 10


class AdamTensor(Structure):
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("exp_avg", c_void_p), ("exp_avg_sq", c_void_p), ("n", c_int64),
                ("lr", c_float), ("step", c_int32)]


ADAM_MAX_TENSORS = 8


class SmilError(RuntimeError):
    pass


class ModelDesc(Structure):
    _fields_ = [("V", c_int32), ("F", c_int32), ("J", c_int32), ("nB", c_int32),
                ("v_template", c_void_p), ("shapedirs", c_void_p), ("faces", c_void_p), ("parents", c_void_p),
                ("skin_idx", c_void_p), ("skin_w", c_void_p), ("jreg_rowptr", c_void_p), ("jreg_col", c_void_p),
                ("jreg_val", c_void_p), ("static_joints", c_int32), ("J_static", c_void_p), ("posedirs", c_void_p)]


class LbsInputs(Structure):
    _fields_ = [("B", c_int32), ("shared_beta", c_int32), ("nB_used", c_int32), ("beta", c_void_p),
                ("theta", c_void_p), ("Rs_in", c_void_p), ("logscale", c_void_p), ("logscale_shared", c_int32),
                ("btrans", c_void_p), ("btrans_shared", c_int32), ("trans", c_void_p), ("trans_after_joints", c_int32),
                ("del_v", c_void_p),
                ("v_template", c_void_p), ("propagate_scaling", c_int32), ("allow_limb_scaling", c_int32), ("theta_mask", c_void_p)]


class LbsOutputs(Structure):
    _fields_ = [(n, c_void_p) for n in ("v_shaped", "J_rest", "Rs", "G", "A", "new_J", "verts", "joints", "v_posed")]


class LbsGrads(Structure):
    _fields_ = [(n, c_void_p) for n in ("d_verts", "d_joints", "d_beta", "d_theta", "d_logscale", "d_btrans",
                                        "d_trans", "d_A", "d_Jrest", "d_Rs", "d_vposed", "d_posefeat", "d_del_v", "d_Rs_in")] + [
        ("accumulate_shared_beta", c_int32), ("up_Rs", c_void_p), ("up_v_shaped", c_void_p), ("clip_depth", c_void_p),
        ("beta_rows", c_void_p)]


class ClipDepth(Structure):
    _fields_ = [("vertex", c_void_p), ("dz", c_void_p), ("range", c_void_p), ("counter", c_void_p), ("capacity", c_int32)]


class Cameras(Structure):
    _fields_ = [("N", c_int32), ("views", c_int32), ("S", c_int32), ("R", c_void_p), ("nR", c_int32),
                ("T", c_void_p), ("nT", c_int32), ("fov", c_void_p), ("nFov", c_int32), ("aspect", c_void_p),
                ("nAspect", c_int32)]


class RasterSettings(Structure):
    _fields_ = [("blur_radius", c_float), ("sigma", c_float), ("faces_per_pixel", c_int32), ("z_clip", c_float),
                ("tie_rule", c_int32), ("clip_depth", c_void_p), ("image0", c_int32)]


class FitConfig(Structure):
    _fields_ = [("N", c_int32), ("J", c_int32), ("nB", c_int32), ("window", c_int32), ("frame0", c_int32),
                ("N_total", c_int32), ("w_j2d", c_float), ("w_reproj", c_float), ("w_betas", c_float),
                ("w_pose", c_float), ("w_limit", c_float), ("w_splay", c_float), ("w_temp", c_float),
                ("limit", c_float), ("train_global", c_int32), ("train_joints", c_int32), ("train_trans", c_int32)]


_lib = None


def load():
    """Load the shared library once; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SmilError(
            f"{LIB_PATH} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C smilify_amd/csrc). There is no CPU fallback.")
    import torch  # noqa: F401  the library must bind to the HIP runtime torch has loaded, not bring up a second one
    lib = ctypes.CDLL(LIB_PATH)
    lib.smil_last_error.restype = c_char_p
    lib.smil_version.restype = c_char_p
    lib.smil_model_create.argtypes = [POINTER(ModelDesc), POINTER(c_void_p)]
    lib.smil_model_destroy.argtypes = [c_void_p]
    lib.smil_model_destroy.restype = None
    lib.smil_model_dims.argtypes = [c_void_p, POINTER(c_int32)]
    lib.smil_lbs_forward.argtypes = [c_void_p, POINTER(LbsInputs), POINTER(LbsOutputs), c_void_p]
    lib.smil_lbs_forward_project.argtypes = [c_void_p, POINTER(LbsInputs), POINTER(LbsOutputs), POINTER(Cameras), c_void_p, c_void_p, c_void_p]
    lib.smil_lbs_backward.argtypes = [c_void_p, POINTER(LbsInputs), POINTER(LbsOutputs), POINTER(LbsGrads), c_void_p]
    lib.smil_lbs_backward_ndc.argtypes = [c_void_p, POINTER(LbsInputs), POINTER(LbsOutputs), POINTER(LbsGrads), POINTER(Cameras),
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.smil_lbs_backward_ndc_supported.argtypes = [c_void_p, c_int32, c_int32]
    lib.smil_clip_depth_backward.argtypes = [POINTER(Cameras), POINTER(ClipDepth), c_int32, c_int32, c_void_p, c_void_p]
    lib.smil_project.argtypes = [POINTER(Cameras), c_void_p, c_int32, c_void_p, c_void_p, c_void_p]
    lib.smil_project_backward.argtypes = [POINTER(Cameras), c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int32, c_void_p, c_void_p]
    lib.smil_fov_reduce.argtypes = [POINTER(Cameras), c_void_p, c_void_p, c_void_p]
    lib.smil_project2.argtypes = [POINTER(Cameras), c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]
    lib.smil_project_backward2.argtypes = [POINTER(Cameras), c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.smil_fit_epilogue.argtypes = [POINTER(FitConfig)] + [c_void_p] * 12 + [c_int32, c_void_p, c_void_p, c_int32, POINTER(Cameras),
                                                                                 c_void_p, c_void_p, c_void_p]
    lib.smil_raster_workspace_bytes.argtypes = [c_void_p, c_int32, c_int32]
    lib.smil_raster_workspace_bytes.restype = c_size_t
    lib.smil_raster_stats.argtypes = [c_void_p, c_int32, c_void_p, c_void_p, c_void_p]
    lib.smil_silhouette_forward.argtypes = [c_void_p, c_void_p, c_int32, c_int32, POINTER(RasterSettings), c_void_p,
                                            c_void_p, c_void_p]
    lib.smil_silhouette_backward.argtypes = [c_void_p, c_void_p, c_int32, c_int32, POINTER(RasterSettings), c_void_p,
                                             c_void_p, c_void_p, c_void_p]
    lib.smil_silhouette_l1_fused.argtypes = [c_void_p, c_void_p, c_int32, c_int32, POINTER(RasterSettings), c_void_p, c_int32,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.smil_prior_losses.argtypes = [POINTER(FitConfig)] + [c_void_p] * 12 + [c_int32, c_void_p]
    lib.smil_mask_rows.argtypes = [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]
    lib.smil_joint_loss.argtypes = [POINTER(FitConfig), c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p]
    lib.smil_pix_scale.argtypes = [POINTER(FitConfig), c_int32, c_int32, c_void_p, c_void_p]
    lib.smil_image_abs_sum.argtypes = [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]
    lib.smil_sil_objective.argtypes = [c_void_p, c_void_p, c_int32, c_void_p, c_void_p]
    lib.smil_window_terms.argtypes = [POINTER(FitConfig), c_int32, c_int32] + [c_void_p] * 10 + [c_int32, c_void_p]
    lib.smil_adam_step.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                                   c_int32, c_void_p]
    lib.smil_adam_step_multi.argtypes = [POINTER(AdamTensor), c_int32, c_float, c_float, c_float, c_void_p]
    lib.smil_adam_step_dev.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float,
                                       c_void_p, c_int32, c_void_p]
    lib.smil_profile_enable.argtypes = [c_int32]
    lib.smil_profile_read.argtypes = [POINTER(c_float), POINTER(c_int32)]
    for name in EXPORTS:
        fn = getattr(lib, name)  # raises AttributeError if the symbol is missing
        if fn.restype is ctypes.c_int:
            fn.restype = c_int32
    _lib = lib
    return lib


def check(rc: int, what: str = "libsmilfit call") -> None:
    if rc != 0:
        msg = load().smil_last_error()
        raise SmilError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
