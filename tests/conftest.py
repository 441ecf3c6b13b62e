import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")
MODEL_FILES = {
    "stick": os.path.join(REPO, "data", "models", "SMILy_STICK.npz"),
    "mouse": os.path.join(REPO, "data", "models", "SMILy_Mouse_static_joints.npz"),
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_library_built():
    """The HIP library is built in-tree by __graft_entry__.build(); build it here if a fresh checkout lacks it
    (hipcc cross-compiles gfx950 without a GPU).  The product itself never falls back: a missing .so raises."""
    from smilify_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()


def oracle_model(tables):
    """Dense fp32 torch tables for the oracle from the product's flat tables (test-side densify)."""
    m = dict(
        v_template=torch.from_numpy(tables.v_template),
        shapedirs=torch.from_numpy(tables.shapedirs),
        J_regressor=torch.from_numpy(tables.dense_J_regressor()),
        weights=torch.from_numpy(tables.dense_weights()),
        parents=tables.parents.copy(),
        faces=torch.from_numpy(tables.faces.astype(np.int64)),
        J_static=torch.from_numpy(tables.J_static) if tables.static_joints else None,
        posedirs=None,
    )
    return m


@pytest.fixture(scope="session")
def tables():
    from smilify_amd import model_io

    cache = {}

    def get(key):
        if key not in cache:
            if key == "synthetic":
                cache[key] = model_io.synthetic_model()
            elif key == "synthetic_static":
                cache[key] = model_io.synthetic_model(static_joints=True, seed=3)
            else:
                cache[key] = model_io.load_model(MODEL_FILES[key])
        return cache[key]

    return get


@pytest.fixture(scope="session")
def golden():
    def get(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return get


def vertex_probe(shape, k):
    n = int(np.prod(shape))
    return torch.from_numpy(np.cos(0.37 * np.arange(n) * (k + 1) + k).astype(np.float32).reshape(shape))
