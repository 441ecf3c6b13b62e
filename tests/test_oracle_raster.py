"""Known-answer tests for the oracle's restatement of the naive soft-silhouette rasteriser.
PARITY UNPINNED by reference-owned vectors (pytorch3d is un-vendored and absent): these analytic
cases are what pins oracle/raster_oracle.c."""
import math

import numpy as np
import torch

from oracle import render_ref as rr

S = 32


def pix_ndc(i):  # output index -> NDC coordinate of the pixel centre (x: col, y: row)
    return 1.0 - (2 * i + 1) / S


def sigmoid(x):
    return 1 / (1 + math.exp(-x))


def test_single_triangle_values_and_orientation():
    # triangle in the +x,+y NDC quadrant: must appear top-left (NDC +x = left, +y = up)
    v = np.array([[[0.1, 0.1, 2.0], [0.9, 0.1, 2.0], [0.1, 0.9, 2.0]]], np.float32)
    f = np.array([[0, 1, 2]], np.int32)
    sil, ncand = rr.silhouette_forward_np(v, f, S)
    assert sil[0, : S // 2, : S // 2].sum() > 10 and sil[0, S // 2 :, :].sum() == 0 and sil[0, :, S // 2 :].sum() == 0
    assert ncand.max() == 1
    # analytic value at every pixel
    for yo in range(S):
        for xo in range(S):
            px, py = pix_ndc(xo), pix_ndc(yo)
            d = _tri_d2(px, py, v[0, :, :2])
            inside = px > 0.1 and py > 0.1 and (px - 0.1) + (py - 0.1) < 0.8
            if not inside and d >= rr.BLUR_RADIUS:
                assert sil[0, yo, xo] == 0.0
            else:
                want = sigmoid((d if inside else -d) / rr.SIGMA)
                assert abs(sil[0, yo, xo] - want) < 2e-5, (yo, xo, sil[0, yo, xo], want)


def _seg_d2(px, py, a, b):
    ba = b - a
    t = np.clip(np.dot(ba, np.array([px, py]) - a) / np.dot(ba, ba), 0, 1)
    q = a + t * ba
    return float((q[0] - px) ** 2 + (q[1] - py) ** 2)


def _tri_d2(px, py, tri):
    return min(_seg_d2(px, py, tri[0], tri[1]), _seg_d2(px, py, tri[0], tri[2]), _seg_d2(px, py, tri[1], tri[2]))


def test_backface_rendered_and_behind_camera_skipped():
    front = np.array([[[-0.5, -0.5, 1.0], [0.5, -0.5, 1.0], [0.0, 0.5, 1.0]]], np.float32)
    back = front[:, [0, 2, 1]].copy()  # opposite winding
    f = np.array([[0, 1, 2]], np.int32)
    a, _ = rr.silhouette_forward_np(front, f, S)
    b, _ = rr.silhouette_forward_np(back, f, S)
    np.testing.assert_allclose(a, b, atol=1e-6)
    behind = front.copy()
    behind[..., 2] = -1.0
    c, n = rr.silhouette_forward_np(behind, f, S)
    assert c.sum() == 0 and n.sum() == 0
    # one vertex behind the camera plane: the kernel alone would skip the face (z_invalid), but clip_faces runs first and cuts it
    # at z_clip - the front part (a quadrilateral on the plane's near side, two triangles) is what gets rendered
    strad = front.copy()
    strad[0, 0, 2] = -0.1
    c, n = rr.silhouette_forward_np(strad, f, S)
    va, fa, src, coef = rr.clip_faces_np(strad[0], f, 5e-4)
    assert va.shape == (5, 3) and fa.shape == (3, 3) and (fa[0] == fa[0][0]).all()       # placeholder + two front triangles
    assert np.allclose(va[3:, 2], 5e-4) and src.tolist() == [[0, 1], [0, 2]]
    np.testing.assert_allclose(coef.sum(1), 1.0, rtol=1e-12)                             # c_a z_a... the crossing is an affine combination in view space
    want, _ = rr.silhouette_forward_np(va[None], fa, S, _clipped=True)
    assert c.sum() > 10 and np.array_equal(c, want)
    rr.set_z_clip(0.0)                                                                   # clipping off: the raw kernel rule drops the face
    try:
        c0, _ = rr.silhouette_forward_np(strad, f, S)
    finally:
        rr.set_z_clip(5e-4)
    assert c0.sum() == 0


def test_top_k_keeps_nearest_by_depth():
    # 6 stacked triangles; the far 3 are big, the near 3 are tiny and lie away from the probe pixel.
    K = 3
    big = [[-0.8, -0.8], [0.8, -0.8], [0.0, 0.8]]
    verts, faces = [], []
    for i in range(6):
        z = 1.0 + i
        for p in big:
            verts.append([p[0], p[1], z])
        faces.append([3 * i, 3 * i + 1, 3 * i + 2])
    v = np.array([verts], np.float32)
    f = np.array(faces, np.int32)
    sil, ncand, ff, fd, fz = rr.silhouette_forward_np(v, f, S, K=K, want_fragments=True)
    c = S // 2
    assert ncand[0, c, c] == 6
    np.testing.assert_array_equal(ff[0, c, c], [0, 1, 2])
    np.testing.assert_allclose(fz[0, c, c], [1, 2, 3], atol=1e-6)
    # reversed submission order: same kept set, sorted output
    f2 = f[::-1].copy()
    _, _, ff2, _, fz2 = rr.silhouette_forward_np(v, f2, S, K=K, want_fragments=True)
    np.testing.assert_allclose(fz2[0, c, c], [1, 2, 3], atol=1e-6)
    np.testing.assert_array_equal(ff2[0, c, c], [5, 4, 3])


def test_alpha_product_over_faces():
    # two overlapping triangles, pixel outside both but inside both blur bands
    t1 = [[-0.5, -0.5, 1.0], [0.5, -0.5, 1.0], [0.0, 0.5, 1.0]]
    t2 = [[-0.5, -0.52, 2.0], [0.5, -0.52, 2.0], [0.0, 0.5, 2.0]]
    v = np.array([t1 + t2], np.float32)
    f = np.array([[0, 1, 2], [3, 4, 5]], np.int32)
    sil, ncand = rr.silhouette_forward_np(v, f, S)
    found = 0
    for yo in range(S):
        for xo in range(S):
            if ncand[0, yo, xo] == 2 and 0 < sil[0, yo, xo] < 0.999:
                px, py = pix_ndc(xo), pix_ndc(yo)
                p = []
                for tri in (np.array(t1)[:, :2], np.array(t2)[:, :2]):
                    d = _tri_d2(px, py, tri)
                    ins = _inside(px, py, tri)
                    p.append(sigmoid((d if ins else -d) / rr.SIGMA))
                want = 1 - (1 - p[0]) * (1 - p[1])
                assert abs(sil[0, yo, xo] - want) < 5e-5
                found += 1
    assert found > 0


def _inside(px, py, tri):
    def e(a, b):
        return (px - a[0]) * (b[1] - a[1]) - (py - a[1]) * (b[0] - a[0])

    s = [e(tri[1], tri[2]), e(tri[2], tri[0]), e(tri[0], tri[1])]
    return all(x > 0 for x in s) or all(x < 0 for x in s)


def test_backward_matches_finite_differences():
    rng = np.random.default_rng(5)
    V = 12
    v = np.zeros((1, V, 3), np.float32)
    v[0, :, :2] = rng.uniform(-0.7, 0.7, (V, 2))
    v[0, :, 2] = rng.uniform(1.0, 3.0, V)
    f = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11], [0, 4, 8], [2, 6, 10]], np.int32)
    w = rng.standard_normal((1, S, S)).astype(np.float32)
    vt = torch.from_numpy(v.copy()).requires_grad_()
    sil = rr.SoftSilhouette.apply(vt, torch.from_numpy(f), S, rr.BLUR_RADIUS, rr.SIGMA, 100)
    (sil * torch.from_numpy(w)).sum().backward()
    g = vt.grad.numpy()
    assert np.all(g[..., 2] == 0)
    eps = 2e-4
    worst = 0.0
    for vi in range(V):
        for c in range(2):
            vp, vm = v.copy(), v.copy()
            vp[0, vi, c] += eps
            vm[0, vi, c] -= eps
            lp = (rr.silhouette_forward_np(vp, f, S)[0].astype(np.float64) * w).sum()
            lm = (rr.silhouette_forward_np(vm, f, S)[0].astype(np.float64) * w).sum()
            fd = (lp - lm) / (2 * eps)
            worst = max(worst, abs(fd - g[0, vi, c]) / (abs(fd) + 1.0))
    assert worst < 0.05, worst


def test_depth_gradient_of_cut_edges_matches_finite_differences():
    """pytorch3d differentiates ``clip_faces`` through its interpolation weight ``w = (z_a - z_clip) / (z_a - z_b)`` and the explicit
    depth factors of the perspective-correct crossing (renderer/mesh/clip.py; left on by p3d_renderer.py:36-47): the end points of an
    edge that crosses the plane receive a DEPTH gradient.  The oracle's analytic one (``clip_depth_gradient``) against central
    differences of its own forward, for an end point in front of the plane and one behind it."""
    S_ = 40
    v = np.array([[[-0.55, -0.4, 2.0e-3], [0.6, -0.35, 1.6e-3], [0.05, 0.55, 2.0e-4],      # vertex 2 is nearer than z_clip = 5e-4
                   [-0.3, 0.2, 1.2e-3], [0.35, 0.25, 3.0e-4], [0.0, -0.6, 0.9e-3]]], np.float32)
    f = np.array([[0, 1, 2], [3, 4, 5]], np.int32)
    w = np.cos(0.37 * np.arange(S_ * S_)).reshape(1, S_, S_).astype(np.float32)
    _, _, src, _ = rr.clip_faces_np(v[0], f, 5e-4)
    assert len(src) == 4
    g = rr.silhouette_backward_np(v, f, S_, w)
    assert np.abs(g[0, :, 2]).max() > 0
    loss = lambda x: float((rr.silhouette_forward_np(x, f, S_)[0].astype(np.float64) * w).sum())  # noqa: E731
    assert (np.abs(g[0, :, 2]) > 0).all()  # both faces are cut: all six vertices are end points of a cut edge
    for vi in range(6):
        eps = 5e-4 * abs(float(v[0, vi, 2]))
        vp, vm = v.copy(), v.copy()
        vp[0, vi, 2] += eps
        vm[0, vi, 2] -= eps
        fd = (loss(vp) - loss(vm)) / (float(vp[0, vi, 2]) - float(vm[0, vi, 2]))
        assert abs(fd - g[0, vi, 2]) <= 0.02 * abs(g[0, vi, 2]), (vi, fd, g[0, vi, 2])


def test_depth_gradient_formula_is_what_autograd_gives_through_the_crossing_point():
    """``clip_depth_gradient`` is written in a factored form (the crossing slides along the edge); pytorch3d reaches the same numbers by
    autograd through ``xy = (xy_a z_a (1 - w) + xy_b z_b w) / z_clip``, ``w = (z_a - z_clip) / (z_a - z_b)``.  Checked in float64 on
    random edges, including crossings hundreds of NDC units outside the image."""
    rng = np.random.default_rng(5)
    for _ in range(50):
        zc = 5e-4
        za, zb = float(rng.uniform(-2e-3, 4.9e-4)), float(rng.uniform(5.1e-4, 5e-3))
        if rng.integers(0, 2):
            za, zb = zb, za
        va = np.array([*rng.uniform(-300, 300, 2), za]); vb = np.array([*rng.uniform(-2, 2, 2), zb])
        g = rng.standard_normal(2)
        z = torch.tensor([za, zb], dtype=torch.float64, requires_grad=True)
        xa, xb = torch.tensor(va[:2]), torch.tensor(vb[:2])
        w = (z[0] - zc) / (z[0] - z[1])
        xy = (xa * z[0] * (1 - w) + xb * z[1] * w) / zc
        (xy * torch.tensor(g)).sum().backward()
        got = rr.clip_depth_gradient(va, vb, g, zc)
        np.testing.assert_allclose(got, z.grad.numpy(), rtol=1e-9, atol=1e-9 * float(np.abs(z.grad.numpy()).max()))


def test_default_camera_and_screen_projection():
    R, T = rr.look_at_view_transform(2.7, 0.0, 0.0)
    np.testing.assert_allclose(R[0].numpy(), np.diag([-1.0, 1.0, -1.0]), atol=1e-6)
    np.testing.assert_allclose(T[0].numpy(), [0, 0, 2.7], atol=1e-6)
    # point on the optical axis lands on the image centre; +x world (=-x view) goes right
    pts = torch.tensor([[[0.0, 0.0, 0.0], [0.5, 0.0, 0.0], [0.0, 0.5, 0.0]]])
    yx = rr.project_points_screen(pts, R, T, torch.tensor([60.0]), 256)
    t = math.tan(math.radians(30))
    np.testing.assert_allclose(yx[0, 0].numpy(), [128, 128], atol=1e-4)
    np.testing.assert_allclose(yx[0, 1].numpy(), [128, 128 + 128 * 0.5 / (t * 2.7)], rtol=1e-5)
    np.testing.assert_allclose(yx[0, 2].numpy(), [128 - 128 * 0.5 / (t * 2.7), 128], rtol=1e-5)
