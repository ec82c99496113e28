"""DeformableConvolution v1: the C oracle against an INDEPENDENT fp64 PyTorch statement of the op whose
gradients come from autograd (tests/dcn_torch_ref.py) -- CPU only, no GPU.

What this pins (VERDICT r3 "Missing 1" / "Next 1a"): `oracle/deform_conv.c` restates MXNet 1.6's
hand-written kernels (deformable_im2col / col2im / col2im_coord); upstream's source is not on this
machine, so it cannot be compiled as a twin.  The second statement shares no code and no structure
with the oracle (absolute coordinates + gathers + autograd against patch-relative pointers +
hand-derived scatter / coordinate weights), so agreement on full tensors shows that the oracle's
three backward kernels ARE the gradient of its forward -- i.e. a bug in the restated backward
would have to be mirrored by PyTorch's autograd.

What stays unpinnable here: that upstream's forward has exactly the semantics both statements share
(zero outside [0,H) x [0,W), last-row value without fade-out from H-1 on, `-1` sentinel in
col2im_coord).  Those facts come from the published kernel text, not from a run of it.

Bars: fp32 oracle vs fp64 autograd, relative to max|tensor| -- measured 2e-7 .. 4.4e-6 (fp32
summation over up to 2304 products); bar 1e-5.  Interior samples (>= 2e-3 from every integer
coordinate) in the full-tensor tests; the kinks, borders and the deliberate non-gradient spots are the
named known-answer tests below.
"""
import numpy as np
import pytest
import torch

from . import dcn_torch_ref as R
from .test_deform_conv import _case, _nok


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _oracle_grads(oracle, x, off, w, dy, kw):
    """the oracle's backward chain, as deformable_convolution-inl.h Backward runs it:
    dcol = W^T dY; dX = col2im(dcol); dOffset = col2im_coord(dcol); dW = sum_n dY col^T"""
    N, F, K = x.shape[0], w.shape[0], w[0].size
    dx, do, dw = np.zeros_like(x), np.zeros_like(off), np.zeros((F, K), np.float64)
    cols = []
    for n in range(N):
        col = oracle.deform_im2col(x[n], off[n], **kw)
        cols.append(col)
        dcol = (w.reshape(F, K).T.astype(np.float64) @ dy[n].reshape(F, -1).astype(np.float64)).astype(np.float32)
        dx[n] = oracle.deform_col2im(dcol, off[n], x[n].shape, **kw)
        do[n] = oracle.deform_col2im_coord(dcol, x[n], off[n], **kw)
        dw += dy[n].reshape(F, -1).astype(np.float64) @ col.astype(np.float64).T
    return np.stack(cols), dx, do, dw.reshape(w.shape)


def _rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() /
                 max(1e-30, float(np.abs(b).max())))


CASES = {
    "3x3": dict(),
    "stride2": dict(stride=2, H=14, W=15),
    "dilate2": dict(pad=2, dil=2),
    "one_group": dict(dg=1, C=6),
    "borders": dict(N=2, C=32, H=25, W=42, F=24, dg=4, off_scale=3.0),   # ~1/4 of the samples cross a border
    "wild": dict(N=1, C=8, H=12, W=16, off_scale=12.0),                   # most samples outside
    "5x5": dict(k=5, pad=2, H=12, W=16, dg=2),
    # the layer of models/dcn/builder.py:14-17 at the BASELINE plane: (256, 50, 84), 256 filters, 4 groups
    "layer": dict(N=1, C=256, H=50, W=84, F=256, dg=4, off_scale=2.0),
}


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_backward_is_autograd_of_an_independent_forward(oracle, name):
    x, off, w, kw = _case(31, **CASES[name])
    if name == "layer":
        w *= 0.25
    geo = (kw["kernel"], kw["pad"], kw["stride"], kw["dil"], kw["dgroup"])
    off = R.keep_off_the_kinks(_t(off), x.shape[2], x.shape[3], *geo).numpy()
    yo = oracle.deform_conv_fwd(x, off, w, **_nok(kw))
    dy = np.random.RandomState(5).standard_normal(yo.shape).astype(np.float32)
    y, dx, do, dw = R.dcn_grads(_t(x), _t(off), _t(w), _t(dy), *geo[1:])
    cols, odx, odo, odw = _oracle_grads(oracle, x, off, w, dy, kw)
    col = R.dcn_col(_t(x), _t(off), *geo).numpy().reshape(cols.shape)
    # forward: sample values one fp32 rounding chain apart, the convolution a K-term fp32 sum apart
    assert float(np.abs(cols - col).max()) <= 4e-6 * max(1.0, float(np.abs(col).max()))
    assert _rel(yo, y.numpy()) <= 1e-5
    # MXNet's three backward kernels == autograd of the independent forward
    assert _rel(odx, dx.numpy()) <= 1e-5, "d_data (deformable_col2im)"
    assert _rel(odo, do.numpy()) <= 1e-5, "d_offset (deformable_col2im_coord)"
    assert _rel(odw, dw.numpy()) <= 1e-5, "d_weight"
    assert float(np.abs(odo).max()) > 0 and float(np.abs(odx).max()) > 0


# ---------------------------------------------------------------------- named known-answer tests --
def _one_sample(h, w, H=6, W=7, hout=2, wout=3):
    """1x1 kernel, no padding: output pixel (hout, wout) samples exactly (h, w); every other output
    pixel samples its own centre.  -> x, offset, geometry kwargs"""
    rs = np.random.RandomState(3)
    x = rs.standard_normal((1, 1, H, W)).astype(np.float32)
    off = np.zeros((1, 2, H, W), np.float32)
    off[0, 0, hout, wout] = np.float32(h - hout)
    off[0, 1, hout, wout] = np.float32(w - wout)
    return x, off, dict(kernel=(1, 1), pad=0, stride=1, dil=1, dgroup=1)


def _both(oracle, h, w, **k):
    """value, d value / d x, d value / d (h, w) of ONE sample: (oracle triple, autograd triple)"""
    x, off, kw = _one_sample(h, w, **k)
    H, W = x.shape[2:]
    hout, wout = k.get("hout", 2), k.get("wout", 3)
    p = hout * W + wout
    col = oracle.deform_im2col(x[0], off[0], **kw)
    g = np.zeros_like(col)
    g[0, p] = 1.0
    odx = oracle.deform_col2im(g, off[0], x[0].shape, **kw)
    odo = oracle.deform_col2im_coord(g, x[0], off[0], **kw)
    xt, ot = _t(x).double().requires_grad_(True), _t(off).double().requires_grad_(True)
    c = R.dcn_col(xt, ot, (1, 1), 0, 1, 1, 1).reshape(-1)[p]
    c.backward()
    return ((float(col[0, p]), odx[0], odo[:, hout, wout]),
            (float(c.detach()), xt.grad[0, 0].numpy(), ot.grad[0, :, hout, wout].numpy()), x[0, 0])


def test_interior_sample_is_plain_bilinear(oracle):
    (v, dx, do), (tv, tdx, tdo), x = _both(oracle, 2.25, 3.5)
    want = 0.75 * 0.5 * x[2, 3] + 0.75 * 0.5 * x[2, 4] + 0.25 * 0.5 * x[3, 3] + 0.25 * 0.5 * x[3, 4]
    assert abs(v - want) < 1e-6 and abs(tv - want) < 1e-6
    np.testing.assert_allclose(dx, tdx, atol=1e-6)
    assert sorted(np.flatnonzero(dx).tolist()) == [2 * 7 + 3, 2 * 7 + 4, 3 * 7 + 3, 3 * 7 + 4]
    np.testing.assert_allclose(do, tdo, atol=1e-6)


def test_integer_coordinates_take_the_right_hand_slope(oracle):
    """On an integer row the bilinear surface has a kink.  MXNet's col2im_coord uses low = (int)h,
    high = low + 1: the slope towards the NEXT row -- what autograd gives for frac = h - floor(h) --
    and the data gradient puts the whole weight on (h, w) (the high neighbour's weight is exactly 0
    and falls outside the |h - row| < 1 window of col2im)."""
    (v, dx, do), (tv, tdx, tdo), x = _both(oracle, 2.0, 3.0)
    assert v == x[2, 3] and abs(tv - x[2, 3]) < 1e-7
    assert np.flatnonzero(dx).tolist() == [2 * 7 + 3] and dx[2, 3] == 1.0
    np.testing.assert_allclose(dx, tdx, atol=1e-7)
    np.testing.assert_allclose(do, [x[3, 3] - x[2, 3], x[2, 4] - x[2, 3]], atol=1e-6)
    np.testing.assert_allclose(do, tdo, atol=1e-6)


def test_last_row_band_has_no_fade_out_and_no_row_gradient(oracle):
    """H-1 <= h < H: im2col's bilinear clamps low = high = H-1 with fraction 0 -- the value of the last
    row, not a fade to zero.  The forward is constant in h there, and so is MXNet's backward:
    get_coordinate_weight's two row terms cancel (low == high), get_gradient_weight puts weight
    (1) * column weights on row H-1.  This is a true gradient, not an approximation."""
    H = 6
    (v, dx, do), (tv, tdx, tdo), x = _both(oracle, H - 1 + 0.4, 3.25, hout=5)
    want = 0.75 * x[5, 3] + 0.25 * x[5, 4]
    assert abs(v - want) < 1e-6 and abs(tv - want) < 1e-6
    np.testing.assert_allclose(dx, tdx, atol=1e-6)
    assert np.flatnonzero(dx).tolist() == [5 * 7 + 3, 5 * 7 + 4]
    # d / d h: exactly 0 in autograd; MXNet sums -a*v1 - b*v2 + a*v1 + b*v2 left to right in fp32,
    # which cancels only up to one rounding of the partial sums
    assert abs(do[0]) <= 2.0 ** -23 * float(np.abs(x).max()) and tdo[0] == 0.0
    np.testing.assert_allclose(do[1], x[5, 4] - x[5, 3], atol=1e-6)
    np.testing.assert_allclose(do, tdo, atol=1e-6)
    # the same in the last column band
    (v, dx, do), (tv, tdx, tdo), x = _both(oracle, 2.5, 7 - 1 + 0.7, wout=6)
    assert abs(v - (0.5 * x[2, 6] + 0.5 * x[3, 6])) < 1e-6
    np.testing.assert_allclose(dx, tdx, atol=1e-6)
    assert abs(do[1]) <= 2.0 ** -23 * float(np.abs(x).max()) and tdo[1] == 0.0
    np.testing.assert_allclose(do, tdo, atol=1e-6)


@pytest.mark.parametrize("h,w", [(6.0, 3.3), (2.5, 7.0), (6.0, 7.0)])
def test_coordinate_exactly_on_the_far_edge(oracle, h, w):
    """h == H (or w == W): the forward's test is `h < H` -> 0.  get_gradient_weight and
    get_coordinate_weight only reject `h > H`, so on their own they would let the sample through;
    what keeps the backward at 0 is the caller: col2im's window test |h - row| < 1 has no row left
    (H - 1 is exactly 1 away) and col2im_coord replaces the coordinate by the -1 sentinel before
    calling (`inv_h >= height`).  Net effect: zero gradient, consistent with the zero forward."""
    (v, dx, do), (tv, tdx, tdo), _ = _both(oracle, h, w, hout=5, wout=6)
    assert v == 0.0 and tv == 0.0
    assert not dx.any() and not tdx.any()
    assert not do.any() and not tdo.any()


@pytest.mark.parametrize("h,w", [(-0.25, 3.5), (2.5, -0.5), (-1e-3, -1e-3), (6.5, 3.0), (2.0, 9.0)])
def test_outside_samples_are_zero_with_zero_gradient(oracle, h, w):
    """-1 < h < 0 would interpolate between a virtual zero row and row 0 in a zero-padded bilinear; the
    v1 kernel instead drops the sample altogether (`h_im >= 0`), forward and backward alike."""
    (v, dx, do), (tv, tdx, tdo), _ = _both(oracle, h, w, hout=0, wout=0)
    assert v == 0.0 and tv == 0.0
    assert not dx.any() and not tdx.any() and not do.any() and not tdo.any()


def test_first_row_is_inside(oracle):
    (v, dx, do), (tv, tdx, tdo), x = _both(oracle, 0.0, 0.0, hout=0, wout=0)
    assert v == x[0, 0]
    np.testing.assert_allclose(dx, tdx, atol=1e-7)
    np.testing.assert_allclose(do, tdo, atol=1e-6)


def test_backward_weights_use_the_absolute_frame_not_the_forwards_patch_frame(oracle):
    """The ONE place where MXNet's backward is deliberately not the bit-level adjoint of its forward:
    im2col interpolates in patch-relative coordinates (map_h = i * dil + offset, fraction of THAT
    fp32 number), col2im recomputes the weights from the absolute coordinate h_in + i * dil + offset,
    whose fp32 rounding is ~H times coarser.  The two weights differ in the last bits (<= ulp(H));
    both statements are within fp32 rounding of the exact bilinear weight."""
    H, W = 50, 84
    rs = np.random.RandomState(0)
    x = rs.standard_normal((1, H, W)).astype(np.float32)
    off = np.full((18, H, W), 0.3, np.float32)            # 3x3, pad 1, one group: every tap +0.3 / +0.3
    kw = dict(kernel=(3, 3), pad=1, stride=1, dil=1, dgroup=1)
    # output pixel (47, 60), tap (2, 2): patch frame 2.3, absolute frame 46 + 2 + 0.3 = 48.3
    p, tap = 47 * W + 60, 8
    e = np.zeros((1, H, W), np.float32)
    e[0, 48, 61] = 1.0
    fwd_w = float(oracle.deform_im2col(e, off, **kw)[tap, p])           # weight of pixel (48, 61) in the forward
    g = np.zeros((9, H * W), np.float32)
    g[tap, p] = 1.0
    bwd_w = float(oracle.deform_col2im(g, off, (1, H, W), **kw)[0, 48, 61])
    exact = (1 - (np.float64(np.float32(0.3)))) ** 2
    assert fwd_w != bwd_w                                              # not bit-adjoint ...
    assert abs(fwd_w - exact) < 2e-7 and abs(bwd_w - exact) < 4e-6     # ... both right to fp32 rounding of their frame
    assert abs(fwd_w - bwd_w) < 4e-6


# ---- the boundary rules, written out (VERDICT r5 "Next 2") -----------------------------------------------------
# Every predicate below is upstream apache/incubator-mxnet @ 1.6.0 (docker/Dockerfile:48),
# src/operator/contrib/nn/deformable_im2col.cuh -- the DCN **v1** file.  (MXNet 1.6 also ships
# modulated_deformable_im2col.cuh, the DCNv2 operator the reference never calls: ITS forward accepts
# `h_im > -1 && w_im > -1 && h_im < height && w_im < width` and its coord kernel uses `<= -1` / a -2 sentinel.
# Round 5's DESIGN.md quoted those v2 forms by mistake; the code always had the v1 forms below.)
#   R1  deformable_im2col_gpu_kernel:        a tap contributes iff  h_im >= 0 && w_im >= 0 && h_im < height && w_im < width
#   R2  deformable_im2col_bilinear:          h_low = floor(h); if h_low >= height - 1: h_low = h_high = height - 1, h = h_low
#                                            (the last row's value with fraction 0: no fade-out); same for w
#   R3  deformable_col2im_gpu_kernel:        pixel (y, x) of the 5 x 5 neighbourhood of ((int)h, (int)w) receives weight iff it
#                                            is inside the image and |h - y| < 1 and |w - x| < 1
#   R4  get_gradient_weight:                 0 if h < 0 || h > height || w < 0 || w > width (note `>`: h == height passes
#                                            here and is stopped by R3's |h - y| < 1); then the clamp of R2
#   R5  deformable_col2im_coord_gpu_kernel:  if (inv_h < 0 || inv_w < 0 || inv_h >= height || inv_w >= width) inv_h = inv_w = -1
#       get_coordinate_weight:               0 if h < 0 || h > height || ...; then the clamp of R2; one-sided slope
#                                            between low and high (low == high: the two row terms cancel)
# The expected values are worked out BY HAND from R1-R5 (H = 6, W = 7, x = the helper's seeded plane), not taken
# from either implementation; the oracle, the fp64 statement where it is the same function, and (GPU twin:
# tests/test_deform_conv.py::test_boundary_rules_on_the_gpu) the HIP kernels must all give them.
def boundary_cases(x):
    """(h, w, hout, wout) -> (value, {pixel: dX weight}, (d/dh, d/dw)); x = the (6, 7) plane"""
    H, W = 6, 7
    cw = 3.25  # an interior column: weights 0.75 on column 3, 0.25 on column 4
    ch = 2.5   # an interior row: weights 0.5 / 0.5 on rows 2 and 3
    row = lambda r: 0.75 * x[r, 3] + 0.25 * x[r, 4]
    colv = lambda c: 0.5 * x[2, c] + 0.5 * x[3, c]
    return [
        # ---- rows: h = -0.5, 0, H - 1, H - 0.5, H at an interior column ----
        ((-0.5, cw, 0, 3), (0.0, {}, (0.0, 0.0))),                                            # R1: h_im >= 0 fails; R4 / R5: 0
        ((0.0, cw, 0, 3), (row(0), {(0, 3): 0.75, (0, 4): 0.25},                              # R1 passes at exactly 0
                           (row(1) - row(0), x[0, 4] - x[0, 3]))),                            # slope towards row 1 (low = 0, high = 1)
        ((H - 1.0, cw, 5, 3), (row(5), {(5, 3): 0.75, (5, 4): 0.25}, (0.0, x[5, 4] - x[5, 3]))),   # R2: clamped, no row slope
        ((H - 0.5, cw, 5, 3), (row(5), {(5, 3): 0.75, (5, 4): 0.25}, (0.0, x[5, 4] - x[5, 3]))),   # R2: no fade-out
        ((float(H), cw, 5, 3), (0.0, {}, (0.0, 0.0))),                                        # R1: h_im < H fails; R3 / R5: 0
        # ---- columns: w = -0.5, 0, W - 1, W - 0.5, W at an interior row ----
        ((ch, -0.5, 2, 0), (0.0, {}, (0.0, 0.0))),
        ((ch, 0.0, 2, 0), (colv(0), {(2, 0): 0.5, (3, 0): 0.5}, (x[3, 0] - x[2, 0], colv(1) - colv(0)))),
        ((ch, W - 1.0, 2, 6), (colv(6), {(2, 6): 0.5, (3, 6): 0.5}, (x[3, 6] - x[2, 6], 0.0))),
        ((ch, W - 0.5, 2, 6), (colv(6), {(2, 6): 0.5, (3, 6): 0.5}, (x[3, 6] - x[2, 6], 0.0))),
        ((ch, float(W), 2, 6), (0.0, {}, (0.0, 0.0))),
    ]


def check_boundary_case(case, want, value, dx, do, x, tol=1e-6):
    wv, wdx, wdo = want
    assert abs(value - wv) <= tol, (case, value, wv)
    full = np.zeros_like(x)
    for (r, c), wgt in wdx.items():
        full[r, c] = wgt
    np.testing.assert_allclose(dx, full, atol=tol, err_msg=str(case))
    # (a clamped axis sums -a v1 - b v2 + a v1 + b v2 left to right in fp32: zero up to one rounding)
    np.testing.assert_allclose(do, wdo, atol=tol + 2.0 ** -22 * float(np.abs(x).max()), err_msg=str(case))


def test_boundary_rules_known_answers(oracle):
    x0 = _one_sample(0.0, 0.0)[0][0, 0]
    for (h, w, hout, wout), want in boundary_cases(x0):
        (v, dx, do), (tv, tdx, tdo), x = _both(oracle, h, w, hout=hout, wout=wout)
        check_boundary_case((h, w), want, v, dx, do, x)
        # the independent fp64 statement agrees wherever it is the same function (everywhere but the integer
        # kinks, where autograd's one-sided slope is the same right-hand one: frac = h - floor(h))
        check_boundary_case((h, w, "fp64 autograd"), want, tv, tdx, tdo, x)


# ---- the operator with a bias and num_group > 1 (models/RepPoints/builder.py:215-245, models/sepc/sepc_dconv.py:5-16)
FULL = {
    "bias": dict(bias=True, G=1),
    "bias_one_dgroup": dict(bias=True, G=1, dg=1, C=8),            # RepPoints: num_deformable_group default 1
    "groups2": dict(bias=False, G=2, C=8, F=6, dg=2),
    "groups4_bias": dict(bias=True, G=4, C=16, F=8, dg=2, H=10, W=12),
    "groups_ne_dgroups": dict(bias=True, G=2, C=12, F=4, dg=3),    # filter groups and deformable groups differ
    "stride2_bias": dict(bias=True, G=1, stride=2, H=14, W=15),
}


@pytest.mark.parametrize("name", list(FULL))
def test_full_operator_oracle_against_autograd(oracle, name):
    """orc_deform_convolution_fwd / _bwd (the restated DeformableConvolutionOp::Forward / ::Backward with
    num_group and bias) against the independent fp64 statement + autograd: y, d_x, d_offset, d_weight,
    d_bias on full tensors."""
    cfg = dict(FULL[name])
    has_bias, G = cfg.pop("bias"), cfg.pop("G")
    x, off, w, kw = _case(41, **cfg)
    F = w.shape[0]
    w = np.ascontiguousarray(w[:, : x.shape[1] // G])                 # (F, C / G, kh, kw)
    b = np.random.RandomState(42).standard_normal(F).astype(np.float32) if has_bias else None
    geo = (kw["kernel"], kw["pad"], kw["stride"], kw["dil"], kw["dgroup"])
    off = R.keep_off_the_kinks(_t(off), x.shape[2], x.shape[3], *geo).numpy()
    yo = oracle.deform_convolution_fwd(x, off, w, b, num_group=G, **_nok(kw))
    dy = np.random.RandomState(43).standard_normal(yo.shape).astype(np.float32)
    ref = R.dcn_grads(_t(x), _t(off), _t(w), _t(dy), *geo[1:], bias=_t(b) if has_bias else None, num_group=G)
    got = oracle.deform_convolution_bwd(dy, x, off, w, bias=has_bias, num_group=G, **_nok(kw))
    assert _rel(yo, ref[0].numpy()) <= 1e-5
    for nm, g_, r_ in zip(("d_x", "d_offset", "d_weight", "d_bias"), got, ref[1:]):
        assert _rel(g_, r_.numpy()) <= 1e-5, (nm, _rel(g_, r_.numpy()))
    if G == 1 and not has_bias:
        return
    # and the group / bias plumbing reduces to the plain operator: one group, zero bias == orc_deform_conv_fwd
    if G == 1:
        y0 = oracle.deform_conv_fwd(x, off, w, **_nok(kw))
        np.testing.assert_array_equal(oracle.deform_convolution_fwd(x, off, w, None, num_group=1, **_nok(kw)), y0)
        np.testing.assert_allclose(yo - b.reshape(1, -1, 1, 1), y0, atol=2e-6 * max(1.0, float(np.abs(yo).max())))
