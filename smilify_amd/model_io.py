"""Model-file ingest: SMAL/SMIL ``.pkl`` -> flat device tables.

The reference reads the model pickle inside ``SMAL.__init__``
(reference: smal_model/smal_torch.py:21-73 loader, :104-196 buffers) and keeps
dense ``weights (V,J)`` / ``J_regressor (V,J)`` matrices.  The HIP path wants
compact tables instead:

* ``skin_idx/skin_w (V,4)``  - the <=4 non-zero bone weights of every vertex
  (the Blender exporter limits SMIL models to 2 bones per vertex, reference:
  3D_model_prep/SMIL_processing_addon.py:214-217; SMPL/SMAL use <=4),
* the joint regressor in CSR (by joint, forward gather) and CSC (by vertex,
  backward gather) form,
* joints in their stored order with ``parent[i] < i`` (validated) plus the depth
  of every joint for the level-synchronous kinematic chain.

Nothing here touches a GPU; the result is a bag of numpy arrays that can be
saved as ``.npz`` (``save_npz``) so the pickle itself never has to travel.
"""
from __future__ import annotations

import importlib
import io
import os
import pickle
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

MAX_BONES_PER_VERTEX = 4

# Only these globals may be resolved while unpickling a model file.  The SMIL
# pickles reference numpy array reconstruction only; legacy SMAL/SMPL pickles
# also carry chumpy arrays and scipy sparse matrices.
_NUMPY_GLOBALS = {
    ("numpy", "dtype"),
    ("numpy", "ndarray"),
    ("numpy.core.numeric", "_frombuffer"),
    ("numpy._core.numeric", "_frombuffer"),
    ("numpy.core.multiarray", "_reconstruct"),
    ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "scalar"),
}
# protocol-2 pickles written under Python 3 wrap array bytes in _codecs.encode(str, 'latin1'): a pure function
_BENIGN_GLOBALS = {("_codecs", "encode"), ("builtins", "bytearray"), ("builtins", "bytes"), ("__builtin__", "bytes"),
                   ("__builtin__", "bytearray"), ("collections", "OrderedDict")}
_SCIPY_GLOBALS = {
    ("scipy.sparse.csc", "csc_matrix"),
    ("scipy.sparse._csc", "csc_matrix"),
    ("scipy.sparse.csr", "csr_matrix"),
    ("scipy.sparse._csr", "csr_matrix"),
    ("scipy.sparse.coo", "coo_matrix"),
    ("scipy.sparse._coo", "coo_matrix"),
}


class _ChArray:
    """Stand-in for ``chumpy.ch.Ch`` found in legacy SMAL pickles: keeps the numbers only."""

    def __init__(self, *args, **kwargs):
        self.data = np.array(args[0]) if args else np.array([])

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.data = np.array(state.get("x", []))
        elif isinstance(state, (tuple, list)):
            self.data = np.array(state[0])
        else:
            self.data = np.array(state)

    def __array__(self, dtype=None, copy=None):
        return self.data if dtype is None else self.data.astype(dtype)


class _ModelUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) == ("chumpy.ch", "Ch"):
            return _ChArray
        if (module, name) in _NUMPY_GLOBALS:
            module = module.replace("numpy.core", "numpy._core") if np.__version__ >= "2" else module
            return getattr(importlib.import_module(module), name)
        if (module, name) in _BENIGN_GLOBALS:
            return getattr(importlib.import_module("builtins" if module == "__builtin__" else module), name)
        if (module, name) in _SCIPY_GLOBALS:
            import scipy.sparse as sp

            return getattr(sp, name)
        raise pickle.UnpicklingError(f"model file references a disallowed global {module}.{name}")


def read_model_pickle(path: str) -> Dict[str, object]:
    """Read a SMAL/SMPL/SMIL model pickle into a plain ``dict`` of numpy data."""
    with open(path, "rb") as fh:
        dd = _ModelUnpickler(io.BytesIO(fh.read()), encoding="latin1").load()
    out = {}
    for k, v in dd.items():
        out[k] = np.array(v.data) if isinstance(v, _ChArray) else v
    return out


def _dense(a) -> np.ndarray:
    if hasattr(a, "todense"):
        return np.asarray(a.todense())
    return np.asarray(a)


@dataclass
class SmilModelTables:
    """Flat fp32/int32 tables of one articulated mesh model."""

    name: str
    v_template: np.ndarray  # (V,3) f32
    shapedirs: np.ndarray  # (nB,3V) f32, inner index v*3+c (reference smal_torch.py:124)
    faces: np.ndarray  # (F,3) i32
    parents: np.ndarray  # (J,) i32, parents[0] == -1
    depth: np.ndarray  # (J,) i32 distance to the root
    skin_idx: np.ndarray  # (V,4) i32 bone ids, padded with 0
    skin_w: np.ndarray  # (V,4) f32 weights, padded with 0
    jreg_rowptr: np.ndarray  # (J+1,) i32   CSR by joint
    jreg_col: np.ndarray  # (nnz,) i32   vertex ids
    jreg_val: np.ndarray  # (nnz,) f32
    static_joints: bool
    J_static: Optional[np.ndarray]  # (J,3) f32 or None
    joint_names: List[str] = field(default_factory=list)
    shape_cov: Optional[np.ndarray] = None  # (>=nB,>=nB) f64
    shape_mean_betas: Optional[np.ndarray] = None
    posedirs: Optional[np.ndarray] = None  # (9(J-1),3V) f32 or None when empty

    # ---- sizes -------------------------------------------------------
    @property
    def V(self) -> int:
        return int(self.v_template.shape[0])

    @property
    def F(self) -> int:
        return int(self.faces.shape[0])

    @property
    def J(self) -> int:
        return int(self.parents.shape[0])

    @property
    def nB(self) -> int:
        return int(self.shapedirs.shape[0])

    # ---- dense views the reference API exposes -------------------------
    def dense_weights(self) -> np.ndarray:
        """(V,J) skinning weights as stored by the reference (smal_torch.py:196)."""
        W = np.zeros((self.V, self.J), np.float32)
        rows = np.repeat(np.arange(self.V), MAX_BONES_PER_VERTEX)
        np.add.at(W, (rows, self.skin_idx.reshape(-1)), self.skin_w.reshape(-1))
        return W

    def dense_J_regressor(self) -> np.ndarray:
        """(V,J) joint regressor, transposed like the reference buffer (smal_torch.py:169-172)."""
        R = np.zeros((self.V, self.J), np.float32)
        for j in range(self.J):
            s, e = self.jreg_rowptr[j], self.jreg_rowptr[j + 1]
            R[self.jreg_col[s:e], j] = self.jreg_val[s:e]
        return R

    def jreg_csc(self):
        """Regressor by vertex: (colptr (V+1,), joint ids, values), used by the backward gather."""
        nnz = len(self.jreg_col)
        joint_of = np.repeat(np.arange(self.J, dtype=np.int32), np.diff(self.jreg_rowptr))
        order = np.argsort(self.jreg_col, kind="stable")
        colptr = np.zeros(self.V + 1, np.int32)
        np.add.at(colptr, self.jreg_col + 1, 1)
        colptr = np.cumsum(colptr).astype(np.int32)
        assert colptr[-1] == nnz
        return colptr, joint_of[order].astype(np.int32), self.jreg_val[order].astype(np.float32)

    def bone_vertex_lists(self):
        """Skin weights by bone: (ptr (J+1,), vertex ids, weights); backward reduction order."""
        idx = self.skin_idx.reshape(-1)
        w = self.skin_w.reshape(-1)
        v = np.repeat(np.arange(self.V, dtype=np.int32), MAX_BONES_PER_VERTEX)
        keep = w != 0
        idx, w, v = idx[keep], w[keep], v[keep]
        order = np.argsort(idx, kind="stable")
        ptr = np.zeros(self.J + 1, np.int32)
        np.add.at(ptr, idx + 1, 1)
        return np.cumsum(ptr).astype(np.int32), v[order].astype(np.int32), w[order].astype(np.float32)

    # ---- persistence ----------------------------------------------------
    def save_npz(self, path: str) -> None:
        d = dict(
            name=np.array(self.name),
            v_template=self.v_template,
            shapedirs=self.shapedirs,
            faces=self.faces,
            parents=self.parents,
            depth=self.depth,
            skin_idx=self.skin_idx.astype(np.uint8 if self.J <= 256 else np.int32),
            skin_w=self.skin_w,
            jreg_rowptr=self.jreg_rowptr,
            jreg_col=self.jreg_col,
            jreg_val=self.jreg_val,
            static_joints=np.array(self.static_joints),
            joint_names=np.array(self.joint_names),
        )
        if self.J_static is not None:
            d["J_static"] = self.J_static
        if self.shape_cov is not None:
            d["shape_cov"] = self.shape_cov
        if self.shape_mean_betas is not None:
            d["shape_mean_betas"] = self.shape_mean_betas
        if self.posedirs is not None:
            d["posedirs"] = self.posedirs
        np.savez_compressed(path, **d)

    @staticmethod
    def load_npz(path: str) -> "SmilModelTables":
        z = np.load(path, allow_pickle=False)
        g = lambda k: z[k] if k in z.files else None  # noqa: E731
        t = SmilModelTables(
            name=str(z["name"]),
            v_template=z["v_template"].astype(np.float32),
            shapedirs=z["shapedirs"].astype(np.float32),
            faces=z["faces"].astype(np.int32),
            parents=z["parents"].astype(np.int32),
            depth=z["depth"].astype(np.int32),
            skin_idx=z["skin_idx"].astype(np.int32),
            skin_w=z["skin_w"].astype(np.float32),
            jreg_rowptr=z["jreg_rowptr"].astype(np.int32),
            jreg_col=z["jreg_col"].astype(np.int32),
            jreg_val=z["jreg_val"].astype(np.float32),
            static_joints=bool(z["static_joints"]),
            J_static=g("J_static"),
            joint_names=[str(s) for s in z["joint_names"]],
            shape_cov=g("shape_cov"),
            shape_mean_betas=g("shape_mean_betas"),
            posedirs=g("posedirs"),
        )
        validate_tables(t)
        return t


def validate_tables(t: SmilModelTables) -> None:
    """Shape/ordering checks every kernel relies on; raises ``ValueError``."""
    V, F, J, nB = t.V, t.F, t.J, t.nB
    if t.v_template.shape != (V, 3) or t.shapedirs.shape != (nB, 3 * V):
        raise ValueError("v_template/shapedirs shapes disagree")
    if t.faces.shape != (F, 3) or t.faces.min() < 0 or t.faces.max() >= V:
        raise ValueError("face indices out of range")
    if t.parents[0] != -1:
        raise ValueError("joint 0 must be the root (parent -1)")
    for i in range(1, J):
        if not (0 <= t.parents[i] < i):
            raise ValueError(f"joint {i}: parent {t.parents[i]} does not precede it")
    if t.skin_idx.shape != (V, MAX_BONES_PER_VERTEX) or t.skin_idx.min() < 0 or t.skin_idx.max() >= J:
        raise ValueError("skin index table malformed")
    if t.jreg_rowptr.shape != (J + 1,) or t.jreg_rowptr[-1] != len(t.jreg_col):
        raise ValueError("joint-regressor CSR malformed")
    if len(t.jreg_col) and (t.jreg_col.min() < 0 or t.jreg_col.max() >= V):
        raise ValueError("joint-regressor vertex ids out of range")
    if t.static_joints and (t.J_static is None or t.J_static.shape != (J, 3)):
        raise ValueError("static-joint model without J table")


def tables_from_dict(dd: Dict[str, object], name: str = "model") -> SmilModelTables:
    """Build tables from an unpickled model dict (schema: SMIL_processing_addon.py:1590-1603)."""
    v_template = np.asarray(dd["v_template"], np.float64).astype(np.float32)
    V = v_template.shape[0]
    sd = np.asarray(dd["shapedirs"], np.float64)
    nB = sd.shape[-1]
    shapedirs = np.reshape(sd, [-1, nB]).T.astype(np.float32).copy()  # (nB,3V) reference :124
    faces = np.asarray(dd["f"]).astype(np.int32)
    parents = np.asarray(dd["kintree_table"])[0].astype(np.int64)
    parents = np.where(parents > 2**31 - 2, -1, parents).astype(np.int32)  # legacy files store 2^32-1
    parents[0] = -1
    J = parents.shape[0]
    depth = np.zeros(J, np.int32)
    for i in range(1, J):
        depth[i] = depth[parents[i]] + 1

    W = np.asarray(_dense(dd["weights"]), np.float32)
    if W.shape != (V, J):
        raise ValueError(f"weights shape {W.shape} != ({V},{J})")
    nnz = (W != 0).sum(1)
    if nnz.max() > MAX_BONES_PER_VERTEX:
        raise ValueError(f"a vertex is bound to {nnz.max()} bones; this build supports <= {MAX_BONES_PER_VERTEX}")
    # keep the bones of each vertex in ascending bone order: same summation order as a dense row walk
    order = np.argsort(-(W != 0).astype(np.int8), axis=1, kind="stable")[:, :MAX_BONES_PER_VERTEX]
    skin_w = np.take_along_axis(W, order, axis=1).astype(np.float32)
    skin_idx = np.where(skin_w != 0, order, 0).astype(np.int32)

    JR = np.asarray(_dense(dd["J_regressor"]), np.float32)  # (J,V)
    if JR.shape != (J, V):
        raise ValueError(f"J_regressor shape {JR.shape} != ({J},{V})")
    rowptr = np.zeros(J + 1, np.int32)
    cols, vals = [], []
    for j in range(J):
        nz = np.nonzero(JR[j])[0]
        cols.append(nz.astype(np.int32))
        vals.append(JR[j, nz].astype(np.float32))
        rowptr[j + 1] = rowptr[j] + len(nz)
    jreg_col = np.concatenate(cols) if cols else np.zeros(0, np.int32)
    jreg_val = np.concatenate(vals) if vals else np.zeros(0, np.float32)

    static = bool(dd.get("static_joint_locs", False))
    J_static = np.asarray(dd["J"], np.float32) if static else None

    pd = np.asarray(dd.get("posedirs", np.zeros(0)))
    posedirs = None
    if pd.size != 0:
        posedirs = np.reshape(pd, [-1, pd.shape[-1]]).T.astype(np.float32).copy()
        if not np.any(posedirs):
            posedirs = None

    t = SmilModelTables(
        name=name,
        v_template=v_template,
        shapedirs=shapedirs,
        faces=faces,
        parents=parents,
        depth=depth,
        skin_idx=skin_idx,
        skin_w=skin_w,
        jreg_rowptr=rowptr,
        jreg_col=jreg_col.astype(np.int32),
        jreg_val=jreg_val.astype(np.float32),
        static_joints=static,
        J_static=J_static,
        joint_names=[str(s) for s in dd.get("J_names", [f"j{i}" for i in range(J)])],
        shape_cov=np.asarray(dd["shape_cov"], np.float64) if "shape_cov" in dd else None,
        shape_mean_betas=np.asarray(dd["shape_mean_betas"], np.float64) if "shape_mean_betas" in dd else None,
        posedirs=posedirs,
    )
    validate_tables(t)
    return t


def load_model(path: str) -> SmilModelTables:
    """Load ``.npz`` tables or convert a model ``.pkl`` on the fly."""
    if path.endswith(".npz"):
        return SmilModelTables.load_npz(path)
    name = os.path.splitext(os.path.basename(path))[0]
    return tables_from_dict(read_model_pickle(path), name=name)


def synthetic_model(V_side: int = 12, J: int = 9, nB: int = 3, seed: int = 0, static_joints: bool = False):
    """Small procedurally generated closed mesh (a bent tube) with a chain skeleton.

    Used by tests and smoke runs that must not depend on any model file.
    """
    rng = np.random.default_rng(seed)
    rings, seg = J + 1, V_side
    verts, faces = [], []
    for r in range(rings):
        x = -1.0 + 2.0 * r / (rings - 1)
        rad = 0.18 + 0.08 * np.sin(3.0 * x)
        for s in range(seg):
            a = 2 * np.pi * s / seg
            verts.append([x, rad * np.cos(a), rad * np.sin(a)])
    verts.append([-1.0, 0, 0])
    verts.append([1.0, 0, 0])
    cap0, cap1 = rings * seg, rings * seg + 1
    for r in range(rings - 1):
        for s in range(seg):
            a, b = r * seg + s, r * seg + (s + 1) % seg
            c, d = a + seg, b + seg
            faces += [[a, b, c], [b, d, c]]
    for s in range(seg):
        faces.append([cap0, (s + 1) % seg, s])
        faces.append([cap1, (rings - 1) * seg + s, (rings - 1) * seg + (s + 1) % seg])
    v_template = np.array(verts, np.float32)
    V = len(verts)
    # skeleton: root at the middle, two chains going left/right plus a side branch
    parents = np.full(J, -1, np.int32)
    jx = np.zeros((J, 3), np.float32)
    half = (J - 1) // 2
    for i in range(1, J):
        if i <= half:
            parents[i] = i - 1
            jx[i, 0] = -i / (half + 0.5)
        else:
            parents[i] = 0 if i == half + 1 else i - 1
            jx[i, 0] = (i - half) / (J - half - 0.5)
    depth = np.zeros(J, np.int32)
    for i in range(1, J):
        depth[i] = depth[parents[i]] + 1
    # weights: two nearest joints along x, linear falloff
    d = np.abs(v_template[:, None, 0] - jx[None, :, 0])
    nearest = np.argsort(d, axis=1)[:, :2]
    dn = np.take_along_axis(d, nearest, 1) + 1e-3
    w = (1 / dn) / (1 / dn).sum(1, keepdims=True)
    skin_idx = np.zeros((V, 4), np.int32)
    skin_w = np.zeros((V, 4), np.float32)
    srt = np.argsort(nearest, axis=1)
    skin_idx[:, :2] = np.take_along_axis(nearest, srt, 1)
    skin_w[:, :2] = np.take_along_axis(w, srt, 1).astype(np.float32)
    # regressor: each joint = mean of its nearest ring
    rowptr = np.zeros(J + 1, np.int32)
    cols, vals = [], []
    for j in range(J):
        ring = int(np.argmin(np.abs(np.linspace(-1, 1, rings) - jx[j, 0])))
        ids = np.arange(ring * seg, (ring + 1) * seg, dtype=np.int32)
        cols.append(ids)
        vals.append(np.full(seg, 1.0 / seg, np.float32))
        rowptr[j + 1] = rowptr[j] + seg
    shapedirs = (0.05 * rng.standard_normal((nB, 3 * V))).astype(np.float32)
    t = SmilModelTables(
        name=f"synthetic_tube_V{V}_J{J}",
        v_template=v_template,
        shapedirs=shapedirs,
        faces=np.array(faces, np.int32),
        parents=parents,
        depth=depth,
        skin_idx=skin_idx,
        skin_w=skin_w,
        jreg_rowptr=rowptr,
        jreg_col=np.concatenate(cols),
        jreg_val=np.concatenate(vals),
        static_joints=static_joints,
        J_static=jx.copy() if static_joints else None,
        joint_names=[f"j{i}" for i in range(J)],
        shape_cov=None,
        shape_mean_betas=None,
    )
    validate_tables(t)
    return t
