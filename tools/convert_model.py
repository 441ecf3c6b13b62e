"""Convert a SMAL / SMIL model ``.pkl`` to the flat ``.npz`` tables used by smilify_amd (and validate it).

    python tools/convert_model.py 3D_model_prep/SMILy_STICK.pkl data/models/SMILy_STICK.npz
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilify_amd import model_io  # noqa: E402


def main(argv):
    if len(argv) != 3:
        print(__doc__)
        return 2
    t = model_io.load_model(argv[1])
    t.save_npz(argv[2])
    bones = int((t.skin_w != 0).sum(1).max())
    print(f"{t.name}: V={t.V} F={t.F} J={t.J} nB={t.nB} static_joints={t.static_joints} max bones/vertex={bones} "
          f"regressor nnz={len(t.jreg_col)} max chain depth={int(t.depth.max())} posedirs={'yes' if t.posedirs is not None else 'empty'}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
