"""DeformableConvolution v1: oracle self-consistency (CPU) and HIP parity (GPU).

PARITY UNPINNED: the reference delegates this op to upstream MXNet 1.6.0, which is
not vendored; the oracle restates the published algorithm.  The CPU tests therefore pin the oracle
with properties that do not depend on the restatement: zero offsets == ordinary convolution,
integer offsets == a shifted tap, col2im is the adjoint of im2col, the offset gradient matches
finite differences.  Float bar: 1e-4 (north_star) -- the GEMM summation order differs.
"""
import numpy as np
import pytest


def _case(seed, N=2, C=8, H=9, W=11, F=6, k=3, pad=1, stride=1, dil=1, dg=4, off_scale=1.5):
    rs = np.random.RandomState(seed)
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    x = rs.standard_normal((N, C, H, W)).astype(np.float32)
    off = (rs.standard_normal((N, dg * 2 * k * k, Ho, Wo)) * off_scale).astype(np.float32)
    w = (rs.standard_normal((F, C, k, k)) * 0.2).astype(np.float32)
    return x, off, w, dict(pad=pad, stride=stride, dil=dil, dgroup=dg, kernel=(k, k))


def _nok(kw):
    """kwargs without the kernel size (taken from the weight shape by the convolution entry points)"""
    return {k: v for k, v in kw.items() if k != "kernel"}


def _conv_ref(x, w, pad, stride, dil):
    N, C, H, W = x.shape
    F, _, kh, kw = w.shape
    Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    xp = np.zeros((N, C, H + 2 * pad, W + 2 * pad), np.float64)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    y = np.zeros((N, F, Ho, Wo), np.float64)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, :, i * dil:i * dil + stride * Ho:stride, j * dil:j * dil + stride * Wo:stride]
            y += np.einsum("nchw,fc->nfhw", patch, w[:, :, i, j].astype(np.float64))
    return y


# ------------------------------------------------------------------------------------------ CPU --
@pytest.mark.parametrize("cfg", [dict(), dict(stride=2), dict(pad=2, dil=2), dict(pad=0)])
def test_oracle_zero_offset_is_ordinary_convolution(oracle, cfg):
    x, off, w, kw = _case(0, off_scale=0.0, **cfg)
    y = oracle.deform_conv_fwd(x, off, w, **_nok(kw))
    np.testing.assert_allclose(y, _conv_ref(x, w, kw["pad"], kw["stride"], kw["dil"]), rtol=1e-4,
                               atol=1e-4)


def test_oracle_integer_offsets_shift_the_tap(oracle):
    x, off, w, kw = _case(1, N=1, C=4, dg=1, off_scale=0.0)
    off[:, 0::2] = 1.0   # every tap samples one row further down
    off[:, 1::2] = -2.0  # and two columns to the left
    col = oracle.deform_im2col(x[0], off[0], dgroup=1)
    C, H, W = x[0].shape
    want = np.zeros_like(col).reshape(C, 3, 3, H, W)
    for i in range(3):
        for j in range(3):
            for h in range(H):
                for w_ in range(W):
                    hh, ww = h - 1 + i + 1, w_ - 1 + j - 2  # pad 1, offsets (+1, -2)
                    if 0 <= hh < H and 0 <= ww < W:
                        want[:, i, j, h, w_] = x[0][:, hh, ww]
    np.testing.assert_array_equal(col, want.reshape(col.shape))


def test_oracle_col2im_is_adjoint_of_im2col(oracle):
    x, off, w, kw = _case(2, N=1)
    rs = np.random.RandomState(3)
    col = oracle.deform_im2col(x[0], off[0], dgroup=kw["dgroup"])
    g = rs.standard_normal(col.shape).astype(np.float32)
    dx = oracle.deform_col2im(g, off[0], x[0].shape, dgroup=kw["dgroup"])
    # <im2col(x), g> == <x, col2im(g)>  (im2col is linear in x)
    np.testing.assert_allclose((col.astype(np.float64) * g).sum(),
                               (x[0].astype(np.float64) * dx).sum(), rtol=1e-4)


def test_oracle_offset_gradient_matches_finite_differences(oracle):
    x, off, w, kw = _case(4, N=1, C=4, H=7, W=8, dg=2)
    rs = np.random.RandomState(5)
    # keep sample points away from integer coordinates (the bilinear kinks)
    frac = off - np.floor(off)
    off = (np.floor(off) + np.clip(frac, 0.2, 0.8)).astype(np.float32)
    col = oracle.deform_im2col(x[0], off[0], dgroup=2)
    g = rs.standard_normal(col.shape).astype(np.float32)
    doff = oracle.deform_col2im_coord(g, x[0], off[0], dgroup=2)
    eps = 1e-2
    for idx in [(0, 2, 3), (5, 0, 0), (17, 6, 7), (35, 3, 4), (20, 1, 6)]:
        o1, o2 = off[0].copy(), off[0].copy()
        o1[idx] += eps
        o2[idx] -= eps
        f1 = (oracle.deform_im2col(x[0], o1, dgroup=2).astype(np.float64) * g).sum()
        f2 = (oracle.deform_im2col(x[0], o2, dgroup=2).astype(np.float64) * g).sum()
        assert abs((f1 - f2) / (2 * eps) - doff[idx]) < 2e-2 * max(1.0, abs(doff[idx])), idx


# ------------------------------------------------------------------------------------------ GPU --
def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(), dict(stride=2), dict(pad=2, dil=2), dict(dg=1, C=6),
                                 dict(H=25, W=42, C=16, F=20, off_scale=3.0),
                                 dict(H=12, W=16),                      # 16-byte staging path
                                 dict(H=20, W=24, off_scale=40.0),     # wild offsets: whole-plane window, many outside
                                 dict(H=20, W=24, off_scale=0.0),      # zero offsets: minimal window
                                 dict(k=1, pad=0, H=12, W=16),         # run-time tap count
                                 dict(k=5, pad=2, H=12, W=16, dg=2),   # 25 taps: per-lane fallback kernels
                                 # groups of 4n channels with Ho*Wo % 4 == 0: the four-channel col2im
                                 dict(H=12, W=16, C=16), dict(H=20, W=24, C=16, off_scale=40.0),
                                 dict(H=16, W=20, C=32, dg=2, stride=2), dict(H=12, W=16, C=16, pad=2, dil=2),
                                 dict(N=1, C=16, H=100, W=168, F=4, off_scale=3.0)])  # ... in 4 row bands
def test_im2col_col2im_coord_match_oracle(ops, oracle, cfg):
    x, off, w, kw = _case(7, **cfg)
    a = dict(kernel=kw["kernel"], pad=kw["pad"], stride=kw["stride"], dilate=kw["dil"],
             num_deformable_group=kw["dgroup"])
    col = ops.deform_im2col(_t(x), _t(off), **a).cpu().numpy()
    want = np.stack([oracle.deform_im2col(x[n], off[n], **{k: v for k, v in kw.items()})
                     for n in range(x.shape[0])])
    np.testing.assert_array_equal(col, want)  # same float ops in the same order: bit exact
    g = np.random.RandomState(8).standard_normal(col.shape).astype(np.float32)
    wdx = np.stack([oracle.deform_col2im(g[n], off[n], x[n].shape, **kw) for n in range(x.shape[0])])
    # sd_deform_col2im_ws (round 6: fixed-point sums where the four-channel kernel applies) and the
    # workspace-free sd_deform_col2im (fp32 compare-and-swap adds)
    for ws in (True, False):
        dx = ops.deform_col2im(_t(g), _t(off), x.shape, workspace=ws, **a).cpu().numpy()
        np.testing.assert_allclose(dx, wdx, rtol=1e-4, atol=1e-4)
    again = ops.deform_col2im(_t(g), _t(off), x.shape, workspace=True, **a).cpu().numpy()
    if (x.shape[1] // kw["dgroup"]) % 4 == 0 and (col.shape[2] % 4) == 0 and kw["kernel"] == (3, 3):
        np.testing.assert_array_equal(again, ops.deform_col2im(_t(g), _t(off), x.shape, workspace=True, **a).cpu().numpy())
    do = ops.deform_col2im_coord(_t(g), _t(x), _t(off), **a).cpu().numpy()
    wdo = np.stack([oracle.deform_col2im_coord(g[n], x[n], off[n], **kw) for n in range(x.shape[0])])
    np.testing.assert_allclose(do, wdo, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(N=2, C=32, H=28, W=36, F=8, dg=4, off_scale=2.5),      # several pixel tiles, the layer's 3x3 / 4-group setup
    dict(N=1, C=256, H=50, W=84, F=8, dg=4, off_scale=2.0),     # the BASELINE plane: 64 channels per group
    dict(N=2, C=64, H=50, W=84, F=8, dg=4, off_scale=40.0),     # wild offsets: every window is the whole plane
    dict(N=2, C=32, H=28, W=36, F=8, dg=4, off_scale=2.5, stride=2),
    dict(N=1, C=12, H=20, W=24, F=8, dg=2, off_scale=1.0),      # 6 channels per group
    dict(N=1, C=32, H=25, W=42, F=8, dg=4, off_scale=1.0),      # H*W % 4 != 0: scalar window copies
    dict(N=1, C=16, H=28, W=42, F=8, dg=2, off_scale=1.5),      # H*W % 4 == 0, W % 4 != 0: the pipe kernel's staging rows must still be 16-byte aligned
])
def test_sampling_kernel_families_give_the_same_bits(ops, cfg):
    """deformable im2col and the offset gradient in their two forms -- LDS windows (`dcn_im2col` / `dcn_coord` = 1,
    the default) and per-lane global gathers (= 0, the reference's structure): same expressions in the same
    channel order, so the same bits; also with whole planes staged instead of windows (`dcn_window` = 0) and with
    im2col's stores in place (`dcn_im2col_pipe` = 0: the default stores a channel's values one trip later)."""
    import torch
    from simpledet_amd._lib import lib
    x, off, w, kw = _case(11, **cfg)
    a = dict(pad=kw["pad"], stride=kw["stride"], dilate=kw["dil"], num_deformable_group=kw["dgroup"])
    tx, to = _t(x), _t(off)
    res = {}
    g = None
    try:
        for mode in (1, 0):
            for k in ("dcn_im2col", "dcn_coord"):
                lib().set_tuning(k, mode)
            col = ops.deform_im2col(tx, to, **a)
            if g is None:
                g = torch.randn_like(col)
            res[mode] = (col, ops.deform_col2im_coord(g, tx, to, **a))
        for k in ("dcn_im2col", "dcn_coord"):
            lib().set_tuning(k, 1)
        lib().set_tuning("dcn_window", 0)
        res["whole"] = (ops.deform_im2col(tx, to, **a), ops.deform_col2im_coord(g, tx, to, **a))
        lib().set_tuning("dcn_window", 1)
        lib().set_tuning("dcn_im2col_pipe", 0)   # col stores in place instead of one channel behind the window loads
        res["inplace"] = (ops.deform_im2col(tx, to, **a), res[1][1])
    finally:
        lib().set_tuning("dcn_window", 1)
        lib().set_tuning("dcn_im2col_pipe", 1)
        for k in ("dcn_im2col", "dcn_coord"):
            lib().set_tuning(k, 1)
    for key in (0, "whole", "inplace"):
        assert torch.equal(res[1][0], res[key][0]), ("im2col", key)
        assert torch.equal(res[1][1], res[key][1]), ("col2im_coord", key)   # same products, same channel order


@pytest.mark.gpu
@pytest.mark.parametrize("split", [2, 1, 0])
@pytest.mark.parametrize("shape", [(1, 128, 128, 16), (2, 256, 300, 72), (3, 70, 4200, 33),
                                   (1, 1, 1, 1), (2, 129, 257, 17), (2, 130, 258, 200), (1, 64, 132, 67)])
def test_mfma_gemm_all_layouts(ops, shape, split):
    """sd_gemm_f32_ws in its four operand layouts and three accumulate modes, on the three matrix-core
    paths: `deform_gemm_split` = 2 (default: scaled fp16 hi/lo split after a max|.| pre-pass, 22
    mantissa bits), 1 (bf16 hi/lo split, 16 bits) and 0 (fp32 MFMA).  Tolerances relative to max|C|:
    2e-6 / 2e-5 / 2e-6 (measured 3e-7 / 6e-6 / 3e-7); shapes cover aligned and unaligned leading
    dimensions (the vector-load path and the per-element path) and a k tail."""
    from simpledet_amd._lib import lib
    Bt, M, N, K = shape
    rs = np.random.RandomState(0)
    A = rs.standard_normal((Bt, M, K)).astype(np.float32)
    B = rs.standard_normal((Bt, K, N)).astype(np.float32)
    want = np.einsum("bmk,bkn->bmn", A.astype(np.float64), B.astype(np.float64))
    tol = (2e-5 if split == 1 else 2e-6) * max(1.0, float(np.abs(want).max()))
    lib().set_tuning("deform_gemm_split", split)
    try:
        for ta in (False, True):
            for tb in (False, True):
                a = _t(A.transpose(0, 2, 1).copy() if ta else A)
                b = _t(B.transpose(0, 2, 1).copy() if tb else B)
                got = ops.gemm_f32(a, b, ta, tb).cpu().numpy()
                assert np.abs(got - want).max() <= tol, (ta, tb, np.abs(got - want).max())
        # accumulate modes
        c0 = rs.standard_normal((Bt, M, N)).astype(np.float32)
        for mode in (1, 2):
            got = ops.gemm_f32(_t(A), _t(B), out=_t(c0), accumulate=mode).cpu().numpy()
            assert np.abs(got - (want + c0)).max() <= tol, mode
    finally:
        lib().set_tuning("deform_gemm_split", 2)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 256, 300, 72), (3, 70, 4200, 33), (1, 2304, 4200, 256), (2, 130, 132, 200)])
def test_gemm_tile_rows_through_lds_are_the_same_bits(ops, shape):
    """Round 6: a whole-tile plain store of the split GEMM leaves through LDS as 512-byte tile rows
    (`deform_gemm_vecstore` = 1, default, when C is 16-byte aligned and N % 4 == 0) instead of as element stores in
    the accumulator layout (= 0): the same values, bit for bit, partial tiles in M and N included."""
    import torch
    from simpledet_amd._lib import lib
    Bt, M, N, K = shape
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randn((Bt, K, M), device="cuda", generator=g)   # trans_a: the dcol product's layout
    b = torch.randn((Bt, K, N), device="cuda", generator=g)
    got = ops.gemm_f32(a, b, trans_a=True)
    lib().set_tuning("deform_gemm_vecstore", 0)
    try:
        want = ops.gemm_f32(a, b, trans_a=True)
    finally:
        lib().set_tuning("deform_gemm_vecstore", 1)
    assert torch.equal(got, want)
    lib().set_tuning("deform_gemm_nt", 0)   # (round 6: those row stores are non-temporal by default; plain ones: same bits)
    try:
        plain = ops.gemm_f32(a, b, trans_a=True)
    finally:
        lib().set_tuning("deform_gemm_nt", 1)
    assert torch.equal(got, plain)
    ref = torch.bmm(a.double().transpose(1, 2), b.double())
    assert float((got.double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("scales", [(1.0, 1.0, 1.0), (50.0, 1e-5, 1e-3), (1e-6, 3e4, 200.0)])
def test_dcn_products_default_split_is_as_good_as_exact_fp32(ops, scales):
    """VERDICT r3 "Next 1b" / ADVICE r3 (low): the DCN layer's three products -- y = W col,
    dcol = W^T dY, dW = sum_n dY col^T -- on the layer's own tensors ((.,256,50,84), 3x3, 4 groups, 256
    filters; col sampled by the product's im2col), each against an fp64 product of the SAME operands:
    the default arithmetic (scaled fp16 split) may be at most 2x as far from it as the exact fp32 MFMA
    path (`deform_gemm_split = 0`), per tensor, in max-abs error.  If this fails the default must go
    back to fp32 MFMA.  `scales` moves the magnitudes of (x, dY, W) around (activations of 50, gradients
    of 1e-5, ...): the split's power-of-two operand scaling must make it magnitude-independent.  The bf16
    split (`= 1`, opt-in) is measured beside them and is the 9x the review asked about."""
    import torch
    from simpledet_amd._lib import lib
    torch.manual_seed(7)
    N, C, H, W, F = 4, 256, 50, 84, 256
    K, P = C * 9, H * W
    sx, sdy, sw = scales
    x = torch.randn(N, C, H, W, device="cuda") * sx
    off = torch.randn(N, 72, H, W, device="cuda") * 2
    wt = (torch.randn(F, K, device="cuda") * 0.05 * sw)
    dy = torch.randn(N, F, P, device="cuda") * sdy
    col = ops.deform_im2col(x, off, (3, 3), 1, 1, 1, 4).reshape(N, K, P)
    wb = wt[None].expand(N, F, K).contiguous()
    prods = {
        "y": (lambda: ops.gemm_f32(wb, col), torch.bmm(wb.double(), col.double())),
        "dcol": (lambda: ops.gemm_f32(wb, dy, trans_a=True), torch.bmm(wb.double().transpose(1, 2), dy.double())),
        "dW": (lambda: ops.gemm_f32(dy, col, trans_b=True).double().sum(0),
               torch.bmm(dy.double(), col.double().transpose(1, 2)).sum(0)),
    }
    err = {}
    try:
        for split in (2, 0, 1):
            lib().set_tuning("deform_gemm_split", split)
            for name, (fn, want) in prods.items():
                err[name, split] = float((fn().double() - want).abs().max() / want.abs().max())
    finally:
        lib().set_tuning("deform_gemm_split", 2)
    for name in prods:
        assert err[name, 2] <= 2.0 * err[name, 0] + 1e-9, (name, err)
        assert err[name, 2] <= 2e-6, (name, err)           # fp32-class accuracy in absolute terms too
        assert err[name, 1] <= 2e-5, (name, err)           # the opt-in bf16 split's documented bound


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["x", "W", "dY"])
def test_dcn_products_with_one_outlier_of_2_to_the_20(ops, where):
    """VERDICT r4 "Next 7": the scaled fp16 split carries 22 mantissa bits only for elements within 2^-13 of
    their OPERAND's maximum (fp16 subnormals below): ONE element of 2^20 x the median inside x, W or dY pushes
    every other element of that operand down to ~17 bits (hi: 11 bits, lo: a subnormal at 2^-24 of the scaled
    range).  Measured here on the layer's three products against fp64 products of the same operands, default
    (`deform_gemm_split = 2`) beside exact fp32 MFMA (`= 0`):
      * outputs the outlier does not reach (other rows / columns / images): the default stays within
        north_star's ABSOLUTE 1e-4 (scaled above |value| 32, `_bar`) and within 4 x the exact path's error
        (measured 0.6 .. 1.8 x: the five lost bits are rounding noise that averages out over K = 2304 or
        P = 4200 terms, below the fp32 accumulation error both paths share);
      * outputs it does reach: <= 2e-6 x max|C|, the split's ordinary bound."""
    import torch
    from simpledet_amd._lib import lib
    torch.manual_seed(11)
    N, C, H, W, F = 2, 256, 50, 84, 256
    K, P = C * 9, H * W
    x = torch.randn(N, C, H, W, device="cuda")
    off = torch.randn(N, 72, H, W, device="cuda") * 2
    wt = torch.randn(F, K, device="cuda") * 0.05
    dy = torch.randn(N, F, P, device="cuda")
    big = float(2 ** 20)
    fo, ko, po = 17, 1234, 2001          # the outlier's filter / col row / pixel, image 0
    col_clean = ops.deform_im2col(x, off, (3, 3), 1, 1, 1, 4).reshape(N, K, P)
    if where == "x":
        x[0, ko // 9, 25, 40] = big * 0.67    # reaches the col rows of channel ko // 9 around that pixel
    elif where == "W":
        wt[fo, ko] = big * 0.05 * 0.67
    else:
        dy[0, fo, po] = big * 0.67
    col = ops.deform_im2col(x, off, (3, 3), 1, 1, 1, 4).reshape(N, K, P)
    wb = wt[None].expand(N, F, K).contiguous()
    prods = {
        "y": (lambda: ops.gemm_f32(wb, col), torch.bmm(wb.double(), col.double())),
        "dcol": (lambda: ops.gemm_f32(wb, dy, trans_a=True), torch.bmm(wb.double().transpose(1, 2), dy.double())),
        "dW": (lambda: ops.gemm_f32(dy, col, trans_b=True), torch.bmm(dy.double(), col.double().transpose(1, 2))),
    }
    # which outputs the outlier takes part in (structurally: the row / column of the operand element)
    reached = {k: torch.zeros_like(v[1], dtype=torch.bool) for k, v in prods.items()}
    if where == "W":
        reached["y"][:, fo, :] = True
        reached["dcol"][:, ko, :] = True
    elif where == "dY":
        reached["dcol"][0, :, po] = True
        reached["dW"][0, fo, :] = True
    else:
        hit = col[0] != col_clean[0]                       # (K, P) col entries the outlier is sampled into
        assert 0 < int(hit.sum()) < 200
        reached["y"][0][:, hit.any(0)] = True
        reached["dW"][0][:, hit.any(1)] = True
    res = {}
    try:
        for split in (2, 0):
            lib().set_tuning("deform_gemm_split", split)
            for name, (fn, want) in prods.items():
                res[name, split] = (fn().double() - want).abs()
    finally:
        lib().set_tuning("deform_gemm_split", 2)
    seen = {}
    for name, (_, want) in prods.items():
        r, calm = reached[name], ~reached[name]
        e2, e0 = res[name, 2], res[name, 0]
        bar = _bar(want[calm].cpu().numpy())   # the file's convention: 1e-4 absolute, scaled above |value| 32
        seen[name] = (float(e2[calm].max()) / bar, float(e2[calm].max()), float(e0[calm].max()),
                      float(e2[r].max()) if bool(r.any()) else 0.0, float(want.abs().max()))
    print("outlier in", where, {k: ["%.3g" % t for t in v] for k, v in seen.items()})
    for name, (of_bar, e2c, e0c, e2r, cmax) in seen.items():
        assert of_bar <= 1.0, (name, "an output the outlier does not reach left the absolute bar", seen)
        assert e2c <= 4.0 * e0c + 1e-9, (name, seen)       # (measured: 0.6 .. 1.8 x the exact fp32 path's error)
        assert e2r <= 2e-6 * cmax, (name, "reached", seen)


@pytest.mark.gpu
def test_gemm_k_slices_of_the_last_round(ops):
    """528 tiles on 512 resident workgroups: the 16 tiles of the last round are cut into k slices
    that add into zeroed C (store mode) or into C (accumulate modes); `deform_gemm_ksplit = 0`
    keeps whole tiles.  Both agree with the fp64 product; garbage in C does not leak into the
    store-mode result."""
    import torch
    from simpledet_amd._lib import lib
    torch.manual_seed(3)
    Bt, M, N, K = 33, 512, 512, 520  # 4 x 4 tiles per image; k tail of 8
    A = torch.randn(Bt, M, K, device="cuda")
    B = torch.randn(Bt, K, N, device="cuda")
    want = torch.bmm(A.double(), B.double())
    tol = 2e-5 * float(want.abs().max())
    outs = []
    for ks in (1, 0):
        lib().set_tuning("deform_gemm_ksplit", ks)
        try:
            c = torch.full((Bt, M, N), float("nan"), device="cuda")
            got = ops.gemm_f32(A, B, out=c)
            assert float((got.double() - want).abs().max()) <= tol, ks
            outs.append(got)
            c0 = torch.randn(Bt, M, N, device="cuda")
            for mode in (1, 2):
                got = ops.gemm_f32(A, B, out=c0.clone(), accumulate=mode)
                assert float((got.double() - want - c0.double()).abs().max()) <= tol, (ks, mode)
        finally:
            lib().set_tuning("deform_gemm_ksplit", 1)
    # whole tiles are the same blocks in both modes
    assert torch.equal(outs[0][:32], outs[1][:32])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(), dict(stride=2, H=14, W=15), dict(pad=2, dil=2),
                                 dict(N=2, C=32, H=25, W=42, F=24, dg=4)])
def test_deform_conv_forward_backward(ops, oracle, cfg):
    x, off, w, kw = _case(11, **cfg)
    a = dict(pad=kw["pad"], stride=kw["stride"], dilate=kw["dil"], num_deformable_group=kw["dgroup"])
    y = ops.deform_conv_forward(_t(x), _t(off), _t(w), **a).cpu().numpy()
    want = oracle.deform_conv_fwd(x, off, w, **_nok(kw))
    assert np.abs(y - want).max() <= _bar(want)
    # backward vs oracle pieces: dcol = W^T dy ; dx = col2im(dcol) ; doff = col2im_coord(dcol) ;
    # dW = sum_n dy_n col_n^T
    rs = np.random.RandomState(12)
    dy = rs.standard_normal(want.shape).astype(np.float32)
    dx, doff, dw = [t.cpu().numpy() for t in
                    ops.deform_conv_backward(_t(dy), _t(x), _t(off), _t(w), **a)]
    N, F = dy.shape[:2]
    K = w[0].size
    wdx, wdo = np.zeros_like(x), np.zeros_like(off)
    wdw = np.zeros((F, K), np.float64)
    for n in range(N):
        dcol = (w.reshape(F, K).T.astype(np.float64) @ dy[n].reshape(F, -1)).astype(np.float32)
        wdx[n] = oracle.deform_col2im(dcol, off[n], x[n].shape, **kw)
        wdo[n] = oracle.deform_col2im_coord(dcol, x[n], off[n], **kw)
        wdw += dy[n].reshape(F, -1).astype(np.float64) @ oracle.deform_im2col(x[n], off[n], **kw).T
    for got, wnt in ((dx, wdx), (doff, wdo), (dw.reshape(F, K), wdw)):
        assert np.abs(got - wnt).max() <= _bar(wnt)
    # req = add accumulates, req = null leaves the buffer untouched
    import torch
    g0 = (torch.ones_like(_t(x)), torch.ones_like(_t(off)), torch.full_like(_t(w), 7.0))
    ops.deform_conv_backward(_t(dy), _t(x), _t(off), _t(w), req=("add", "add", "null"), grads=g0, **a)
    assert np.abs(g0[0].cpu().numpy() - (wdx + 1)).max() <= _bar(wdx)
    assert np.abs(g0[1].cpu().numpy() - (wdo + 1)).max() <= _bar(wdo)
    assert float((g0[2] - 7.0).abs().max()) == 0


@pytest.mark.gpu
def test_deform_conv_at_the_baseline_shape(ops, oracle):
    """BASELINE configs[4] / SURVEY 8(d): x (.,256,50,84), 3x3, pad 1, 4 deformable groups, 256 filters
    (models/dcn/builder.py:14-17); one image through the C ABI against the oracle: im2col bit for
    bit, convolution forward and the three gradients within 1e-4 x max (north_star's float bar; the
    GEMM sums K = 2304 products in another order than the oracle)."""
    x, off, w, kw = _case(21, N=1, C=256, H=50, W=84, F=256, dg=4, off_scale=2.0)
    w *= 0.25
    a = dict(pad=1, stride=1, dilate=1, num_deformable_group=4)
    col = ops.deform_im2col(_t(x), _t(off), (3, 3), 1, 1, 1, 4).cpu().numpy()[0]
    wcol = oracle.deform_im2col(x[0], off[0], **kw)
    np.testing.assert_array_equal(col, wcol)
    y = ops.deform_conv_forward(_t(x), _t(off), _t(w), **a).cpu().numpy()
    F, K = 256, w[0].size
    want = (w.reshape(F, K).astype(np.float64) @ wcol.astype(np.float64)).reshape(y.shape).astype(np.float32)
    assert np.abs(y - want).max() <= _bar(want)
    rs = np.random.RandomState(22)
    dy = rs.standard_normal(want.shape).astype(np.float32)
    dx, doff, dw = [t.cpu().numpy() for t in ops.deform_conv_backward(_t(dy), _t(x), _t(off), _t(w), **a)]
    dcol = (w.reshape(F, K).T.astype(np.float64) @ dy[0].reshape(F, -1)).astype(np.float32)
    wdx = oracle.deform_col2im(dcol, off[0], x[0].shape, **kw)[None]
    wdo = oracle.deform_col2im_coord(dcol, x[0], off[0], **kw)[None]
    wdw = dy[0].reshape(F, -1).astype(np.float64) @ wcol.astype(np.float64).T
    for name, got, wnt in (("d_data", dx, wdx), ("d_offset", doff, wdo), ("d_weight", dw.reshape(F, K), wdw)):
        err = float(np.abs(got - wnt).max())
        assert err <= _bar(wnt), (name, err, float(np.abs(wnt).max()))


def _bar(want):
    """north_star's 1e-4 is an ABSOLUTE bar; it is held wherever the tensor's values stay within the
    range the reference's own fp32 sums resolve to that (|value| <= 32, the convention of
    tests/test_roi_align.py's geometry fuzz), and scales with the magnitude above that (dW sums 16,800
    products per image: |dW| reaches hundreds, where fp32 itself carries ~1e-4 of rounding)."""
    return 1e-4 * max(1.0, float(np.abs(want).max()) / 32.0)


@pytest.mark.gpu
def test_deform_conv_layer_holds_the_absolute_bar_on_every_image(ops, oracle):
    """VERDICT r3 "Next 1b": the layer of models/dcn/builder.py:14-17 at the BASELINE plane, several
    images, DEFAULT arithmetic, every image checked: |err| <= 1e-4 absolute (see _bar) for y, d_data,
    d_offset and d_weight.  References: fp64 products of the oracle's col (the product's im2col is
    bit-equal to it) and the oracle's col2im / col2im_coord fed with the fp64 dcol."""
    N = 4
    x, off, w, kw = _case(33, N=N, C=256, H=50, W=84, F=256, dg=4, off_scale=2.0)
    w *= 0.25
    a = dict(pad=1, stride=1, dilate=1, num_deformable_group=4)
    F, K = 256, w[0].size
    wm = w.reshape(F, K).astype(np.float64)
    y = ops.deform_conv_forward(_t(x), _t(off), _t(w), **a).cpu().numpy()
    dy = np.random.RandomState(34).standard_normal(y.shape).astype(np.float32)
    dx, doff, dw = [t.cpu().numpy() for t in ops.deform_conv_backward(_t(dy), _t(x), _t(off), _t(w), **a)]
    wdw = np.zeros((F, K), np.float64)
    for n in range(N):
        col = oracle.deform_im2col(x[n], off[n], **kw).astype(np.float64)
        wy = (wm @ col).reshape(y[n].shape)
        assert float(np.abs(y[n] - wy).max()) <= _bar(wy), ("y", n)
        dcol = (wm.T @ dy[n].reshape(F, -1).astype(np.float64)).astype(np.float32)
        wdx = oracle.deform_col2im(dcol, off[n], x[n].shape, **kw)
        wdo = oracle.deform_col2im_coord(dcol, x[n], off[n], **kw)
        assert float(np.abs(dx[n] - wdx).max()) <= _bar(wdx), ("d_data", n, float(np.abs(dx[n] - wdx).max()))
        assert float(np.abs(doff[n] - wdo).max()) <= _bar(wdo), ("d_offset", n, float(np.abs(doff[n] - wdo).max()))
        wdw += dy[n].reshape(F, -1).astype(np.float64) @ col.T
    assert float(np.abs(dw.reshape(F, K) - wdw).max()) <= _bar(wdw), ("d_weight", float(np.abs(wdw).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(N=2, C=64, H=12, W=16, F=24, dg=4),                      # 16 channels per group: one slab each
    dict(N=3, C=64, H=20, W=24, F=70, dg=2, off_scale=3.0),       # 2 slabs per group, F not a multiple of 32
    dict(N=9, C=32, H=12, W=16, F=8, dg=2),                       # more images than XCDs, N % 8 != 0
    dict(N=1, C=64, H=25, W=44, F=300, dg=4, off_scale=2.0),      # two filter tiles (F > 256)
    dict(N=2, C=64, H=20, W=24, F=32, dg=4, off_scale=40.0),      # wild offsets: most samples outside
    dict(N=2, C=64, H=20, W=24, F=32, dg=4, off_scale=0.0),       # zero offsets: an ordinary convolution
    dict(N=2, C=32, H=16, W=20, F=16, dg=2, stride=2),
    dict(N=2, C=32, H=12, W=16, F=16, dg=2, pad=2, dil=2),
    dict(N=1, C=64, H=100, W=168, F=16, dg=4, off_scale=30.0),    # windows too large for LDS: corners from global
    dict(N=2, C=256, H=50, W=84, F=256, dg=4, off_scale=2.0),     # the layer of models/dcn/builder.py:14-17
])
def test_fused_forward_without_col_matrix(ops, oracle, cfg):
    """sd_deform_conv_fwd_nocol (sampling fused into the GEMM, no col matrix) against the unfused
    forward and the oracle: every image, absolute bar.  Also with tiles of a forced width (partial
    last tiles, one-pixel tiles) and with the fusion switched off behind the same entry point."""
    import torch
    from simpledet_amd._lib import lib
    x, off, w, kw = _case(51, **cfg)
    if cfg.get("C", 8) >= 256:
        w *= 0.25
    a = dict(pad=kw["pad"], stride=kw["stride"], dilate=kw["dil"], num_deformable_group=kw["dgroup"])
    tx, to, tw = _t(x), _t(off), _t(w)
    y = ops.deform_conv_forward(tx, to, tw, **a)                       # fused
    y_unf, _ = ops.deform_conv_forward(tx, to, tw, keep_col=True, **a)   # im2col + GEMM
    F, K = w.shape[0], w[0].size
    wm = w.reshape(F, K).astype(np.float64)
    yn = y.cpu().numpy()
    for n in range(x.shape[0]):
        col = oracle.deform_im2col(x[n], off[n], **kw).astype(np.float64)
        want = (wm @ col).reshape(yn[n].shape)
        assert float(np.abs(yn[n] - want).max()) <= _bar(want), (n, float(np.abs(yn[n] - want).max()))
    assert float((y - y_unf).abs().max()) <= _bar(y_unf.cpu().numpy())
    for tile in (96, 37, 1) if x.shape[2] * x.shape[3] <= 600 else (64,):
        lib().set_tuning("dcn_fused_tile", tile)
        try:
            yt = ops.deform_conv_forward(tx, to, tw, **a)
        finally:
            lib().set_tuning("dcn_fused_tile", 0)
        assert float((yt - y).abs().max()) <= 1e-5 * max(1.0, float(y.abs().max())), tile
    lib().set_tuning("dcn_fused", 0)
    try:
        y0 = ops.deform_conv_forward(tx, to, tw, **a)
    finally:
        lib().set_tuning("dcn_fused", 1)
    assert torch.equal(y0, y_unf)


@pytest.mark.gpu
@pytest.mark.parametrize("where", ["plane_start", "interior", "nan"])
def test_fused_forward_with_non_finite_input(ops, where):
    """An inf / nan in x may only reach the outputs whose samples touch it: the col-free forward then runs
    its global-gather instance (which reads exactly the corners the reference reads) instead of the LDS
    windows, whose weight-0 reads (slot 0 for outside samples, the neighbour behind a clamp) would carry
    0 x inf to other pixels.  Same finite / non-finite pattern as im2col + GEMM, same finite values."""
    import torch
    x, off, w, kw = _case(71, N=2, C=32, H=12, W=16, F=16, dg=2, off_scale=3.0)
    a = dict(pad=kw["pad"], stride=kw["stride"], dilate=kw["dil"], num_deformable_group=kw["dgroup"])
    bad = np.nan if where == "nan" else np.inf
    if where == "plane_start":
        x[0, 5, 0, 0] = bad          # index 0 of a channel plane: what an outside sample's corner 0 points at
    else:
        x[1, 20, 6, 7] = bad
    tx, to, tw = _t(x), _t(off), _t(w)
    y = ops.deform_conv_forward(tx, to, tw, **a)
    y_unf, _ = ops.deform_conv_forward(tx, to, tw, keep_col=True, **a)
    fin, fin_u = torch.isfinite(y), torch.isfinite(y_unf)
    assert torch.equal(fin, fin_u)
    assert not bool(fin.all()) and bool(fin.any())
    assert float((y[fin] - y_unf[fin]).abs().max()) <= _bar(y_unf[fin].cpu().numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("kshape", [(1, 9), (9, 1), (3, 3), (1, 1)])
def test_nocol_entry_takes_any_kernel_shape(ops, oracle, kshape):
    """The fused kernel is for 3x3 taps; nine taps in a row (or any other shape) must take the im2col + GEMM
    path behind sd_deform_conv_fwd_nocol, not the fused one with a wrong (i, j) per tap."""
    kh, kw = kshape
    rs = np.random.RandomState(61)
    N, C, H, W, F, dg = 2, 32, 12, 20, 8, 2
    Ho, Wo = H - (kh - 1), W - (kw - 1)
    x = rs.standard_normal((N, C, H, W)).astype(np.float32)
    off = (rs.standard_normal((N, dg * 2 * kh * kw, Ho, Wo)) * 1.5).astype(np.float32)
    w = (rs.standard_normal((F, C, kh, kw)) * 0.2).astype(np.float32)
    y = ops.deform_conv_forward(_t(x), _t(off), _t(w), pad=0, stride=1, dilate=1,
                                num_deformable_group=dg).cpu().numpy()
    want = oracle.deform_conv_fwd(x, off, w, pad=0, stride=1, dil=1, dgroup=dg)
    assert y.shape == want.shape
    assert np.abs(y - want).max() <= _bar(want)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(8, 6, 8, 2), (4, 6, 7, 1)])
def test_boundary_rules_on_the_gpu(ops, oracle, shape):
    """GPU twin of tests/test_deform_conv_autograd.py::test_boundary_rules_known_answers (which ties the oracle to
    hand-worked values of upstream's rules R1-R5): every (tap, pixel) of this tensor samples a coordinate from the
    boundary set {-0.5, -0.001, 0, H-1, H-0.5, H, H+0.5, interior} x the same for w, every combination present.
    im2col bit-equal, col2im / col2im_coord within 1e-6 of the oracle (sums of <= 72 terms), on the LDS-plane
    kernels ((8,6,8): H*W % 4 == 0, four channels per group) and on the per-lane kernels ((4,6,7))."""
    C, H, W, dg = shape
    rs = np.random.RandomState(90)
    x = rs.standard_normal((1, C, H, W)).astype(np.float32)
    hs = [-0.5, -1e-3, 0.0, H - 1.0, H - 0.5, float(H), H + 0.5, 2.25]
    ws = [-0.5, -1e-3, 0.0, W - 1.0, W - 0.5, float(W), W + 0.5, 3.5]
    off = np.zeros((1, dg * 18, H, W), np.float32)
    k = 0
    for g_ in range(dg):
        for t in range(9):
            for ho in range(H):
                for wo in range(W):
                    th, tw = hs[k % 8], ws[(k // 8) % 8]
                    k += 1 + (g_ == 1)   # another walk through the combinations in the second group
                    off[0, g_ * 18 + 2 * t, ho, wo] = np.float32(th - (ho - 1 + t // 3))
                    off[0, g_ * 18 + 2 * t + 1, ho, wo] = np.float32(tw - (wo - 1 + t % 3))
    assert k >= 64
    kw = dict(kernel=(3, 3), pad=1, stride=1, dil=1, dgroup=dg)
    a = dict(kernel=(3, 3), pad=1, stride=1, dilate=1, num_deformable_group=dg)
    col = ops.deform_im2col(_t(x), _t(off), **a).cpu().numpy()
    wcol = oracle.deform_im2col(x[0], off[0], **kw)
    np.testing.assert_array_equal(col[0], wcol)
    assert (wcol == 0).mean() > 0.3 and (wcol != 0).mean() > 0.1     # both sides of every border are populated
    g = rs.standard_normal(col.shape).astype(np.float32)
    for ws_ in (True, False):
        dx = ops.deform_col2im(_t(g), _t(off), x.shape, workspace=ws_, **a).cpu().numpy()
        np.testing.assert_allclose(dx[0], oracle.deform_col2im(g[0], off[0], x[0].shape, **kw), atol=2e-5)
    do = ops.deform_col2im_coord(_t(g), _t(x), _t(off), **a).cpu().numpy()
    np.testing.assert_allclose(do[0], oracle.deform_col2im_coord(g[0], x[0], off[0], **kw), atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["random", "pile_up", "tiny_gradients", "inf"])
def test_backward_fixed_point_col2im(ops, oracle, case):
    """Round 4: the layer's backward sums dX with integer LDS adds in fixed point (dcn_col2im_fx = 1,
    default), the unit derived from max|dcol| (GEMM epilogue) x a bound on the bilinear weights one
    pixel can collect (deform_col2im_wsum_kernel).  Against the fp32 compare-and-swap path (= 0) and
    the oracle; bit-reproducible; every sample of an image piled onto ONE pixel (the bound's worst
    case); gradients of 1e-20 (the scale is a power of two from the data, not a constant); inf in dY
    takes the float path and reaches the pixels the reference sends it to."""
    import torch
    from simpledet_amd._lib import lib
    x, off, w, kw = _case(31, N=2, C=32, H=12, W=10, F=16, dg=2)
    a = dict(pad=kw["pad"], stride=kw["stride"], dilate=kw["dil"], num_deformable_group=kw["dgroup"])
    rs = np.random.RandomState(32)
    N, C, H, W = x.shape
    F, K = w.shape[0], w[0].size
    dy = rs.standard_normal((N, F, H, W)).astype(np.float32)
    if case == "pile_up":
        # every tap of every pixel samples the point (3.0, 4.0): offset = target - (h_in + i, w_in + j)
        hh, ww = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        for t in range(9):
            off[:, 2 * t::18] = 3.0 - (hh - 1 + t // 3)
            off[:, 2 * t + 1::18] = 4.0 - (ww - 1 + t % 3)
    if case == "tiny_gradients":
        dy *= 1e-20
    if case == "inf":
        dy[1, 3, 5, 6] = np.inf
    outs = {}
    try:
        for fx in (1, 1, 0):
            lib().set_tuning("dcn_col2im_fx", fx)
            outs.setdefault(fx, []).append(
                ops.deform_conv_backward(_t(dy), _t(x), _t(off), _t(w), **a)[0].cpu().numpy())
    finally:
        lib().set_tuning("dcn_col2im_fx", 1)
    np.testing.assert_array_equal(outs[1][0], outs[1][1])          # order independent
    wdx = np.zeros_like(x)
    with np.errstate(invalid="ignore", over="ignore"):
        for n in range(N):
            dcol = (w.reshape(F, K).T.astype(np.float64) @ dy[n].reshape(F, -1).astype(np.float64)).astype(np.float32)
            wdx[n] = oracle.deform_col2im(dcol, off[n], x[n].shape, **kw)
    if case == "inf":
        # image 0 is finite and unaffected; in image 1 the non-finite pattern is the float path's
        assert np.isfinite(outs[1][0][0]).all() and np.abs(outs[1][0][0] - wdx[0]).max() <= _bar(wdx[0])
        np.testing.assert_array_equal(np.isfinite(outs[1][0][1]), np.isfinite(outs[0][0][1]))
        assert not np.isfinite(outs[1][0][1]).all()
        return
    scale = float(np.abs(wdx).max())
    assert scale > 0
    for got in (outs[1][0], outs[0][0]):
        assert np.abs(got - wdx).max() <= 2e-5 * scale, (case, np.abs(got - wdx).max(), scale)
    if case == "pile_up":
        # one pixel per (image, channel) holds the whole gradient: 9 * H * W samples with weight 1
        nz = np.abs(wdx) > 0
        assert nz.reshape(N, C, -1).sum(-1).max() == 1 and nz[:, :, 3, 4].all()


@pytest.mark.gpu
def test_backward_with_the_forward_col_matrix(ops):
    """sd_deform_conv_bwd_cached: the col matrix left in the forward's workspace replaces the
    backward's own im2col -- same bits for d_offset (a per-lane reduction), d_data and d_weight up
    to the order of their atomic accumulations (LDS adds / adds over images: 1e-5 x max, as between
    two runs of the plain backward); the autograd mirror uses it."""
    import torch
    from simpledet_amd import contrib
    torch.manual_seed(4)
    N, C, H, W, F = 3, 32, 20, 28, 24
    x = torch.randn(N, C, H, W, device="cuda")
    off = torch.randn(N, 4 * 18, H, W, device="cuda") * 1.5
    w = torch.randn(F, C, 3, 3, device="cuda") * 0.1
    y, fws = ops.deform_conv_forward(x, off, w, 1, 1, 1, 4, keep_col=True)
    dy = torch.randn_like(y)
    a = ops.deform_conv_backward(dy, x, off, w, 1, 1, 1, 4)
    b = ops.deform_conv_backward(dy, x, off, w, 1, 1, 1, 4, fwd_ws=fws)
    close = lambda u, v: float((u - v).abs().max()) <= 1e-5 * float(v.abs().max())
    assert torch.equal(a[1], b[1]) and close(b[0], a[0]) and close(b[2], a[2])
    # autograd mirror == raw calls
    xr, offr, wr = (t.clone().requires_grad_(True) for t in (x, off, w))
    out = contrib.DeformableConvolution(xr, offr, wr, kernel=(3, 3), pad=(1, 1), num_filter=F,
                                        num_deformable_group=4, no_bias=True)
    out.backward(dy)
    assert torch.equal(out.detach(), y) and torch.equal(offr.grad, a[1])
    assert close(xr.grad, a[0]) and close(wr.grad, a[2])


# ---- the operator with a bias and num_group (models/RepPoints/builder.py:215-245, models/sepc/sepc_dconv.py:5-16)
@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(N=2, C=64, H=12, W=16, F=24, dg=4, bias=True),                       # fused forward + bias epilogue
    dict(N=2, C=256, H=50, W=84, F=256, dg=1, bias=True, off_scale=2.0),      # the RepPoints head: 256 -> 256, one dgroup
    dict(N=1, C=64, H=25, W=44, F=300, dg=4, bias=True, off_scale=2.0),       # two filter tiles
    dict(N=2, C=64, H=25, W=42, F=24, dg=4, bias=True),                       # H*W % 4 != 0: im2col + GEMM + bias pass
    dict(N=3, C=8, H=9, W=11, F=6, dg=1, bias=True),                          # tiny, unaligned rows
    dict(N=2, C=64, H=12, W=16, F=24, dg=4, bias=True, G=2),                  # filter groups
    dict(N=2, C=32, H=9, W=11, F=12, dg=2, bias=False, G=4),
    dict(N=2, C=24, H=10, W=12, F=8, dg=3, bias=True, G=2, stride=2),         # filter groups != deformable groups
    dict(N=2, C=32, H=12, W=16, F=16, dg=2, bias=True, G=1, pad=2, dil=2),
])
def test_full_operator_bias_and_groups(ops, oracle, cfg):
    """sd_deform_convolution_fwd / _bwd against the oracle's DeformableConvolutionOp restatement: y and the
    four gradients, col-free and col-keeping forward (same bits whenever both are im2col + GEMM), the
    backward with and without the kept col, req = add on every gradient."""
    import torch
    cfg = dict(cfg)
    has_bias, G = cfg.pop("bias"), cfg.pop("G", 1)
    x, off, w, kw = _case(61, **cfg)
    if x.shape[1] >= 256:
        w *= 0.25
    F = w.shape[0]
    w = np.ascontiguousarray(w[:, : x.shape[1] // G])
    b = (np.random.RandomState(62).standard_normal(F) * 3).astype(np.float32) if has_bias else None
    a = dict(pad=kw["pad"], stride=kw["stride"], dilate=kw["dil"], num_deformable_group=kw["dgroup"], num_group=G)
    tx, to, tw, tb = _t(x), _t(off), _t(w), (_t(b) if has_bias else None)
    want = oracle.deform_convolution_fwd(x, off, w, b, num_group=G, **_nok(kw))
    y = ops.deform_conv_forward(tx, to, tw, bias=tb, **a)
    yk, fws = ops.deform_conv_forward(tx, to, tw, bias=tb, keep_col=True, **a)
    for got in (y, yk):
        err = float(np.abs(got.cpu().numpy() - want).max())
        assert err <= _bar(want), err
    if has_bias:   # the bias really is added after the products: y - y(no bias) == bias up to one rounding
        y0 = ops.deform_conv_forward(tx, to, tw, **a)
        d = (y - y0).cpu().numpy() - b.reshape(1, -1, 1, 1)
        assert float(np.abs(d).max()) <= 2e-6 * max(1.0, float(np.abs(want).max()))
    dy = np.random.RandomState(63).standard_normal(want.shape).astype(np.float32)
    wg = oracle.deform_convolution_bwd(dy, x, off, w, bias=has_bias, num_group=G, **_nok(kw))
    tdy = _t(dy)
    names = ("d_data", "d_offset", "d_weight", "d_bias")
    for fw in (None, fws):
        g = ops.deform_conv_backward(tdy, tx, to, tw, fwd_ws=fw, bias=has_bias, **a)
        assert len(g) == len(wg)
        for nm, got, wnt in zip(names, g, wg):
            err = float(np.abs(got.cpu().numpy() - wnt).max())
            assert err <= _bar(wnt), (nm, fw is not None, err, float(np.abs(wnt).max()))
    # req = add on everything: gradients land on top of what the buffers held
    g0 = [torch.full_like(t, 2.0) for t in (tx, to, tw)] + ([torch.full((F,), 2.0, device="cuda")] if has_bias else [])
    ops.deform_conv_backward(tdy, tx, to, tw, req=("add",) * len(g0), grads=tuple(g0), bias=has_bias, **a)
    for nm, got, wnt in zip(names, g0, wg):
        assert float(np.abs(got.cpu().numpy() - (wnt + 2)).max()) <= _bar(wnt), nm
    # req = null on the bias leaves it alone
    if has_bias:
        gb = torch.full((F,), 5.0, device="cuda")
        ops.deform_conv_backward(tdy, tx, to, tw, req=("write", "write", "write", "null"),
                                 grads=(torch.empty_like(tx), torch.empty_like(to), torch.empty_like(tw), gb),
                                 bias=True, **a)
        assert float((gb - 5.0).abs().max()) == 0
    # d_bias is a fixed-order sum: the same bits in every run
    if has_bias:
        g1 = ops.deform_conv_backward(tdy, tx, to, tw, bias=True, **a)
        g2 = ops.deform_conv_backward(tdy, tx, to, tw, bias=True, **a)
        assert torch.equal(g1[3], g2[3])


@pytest.mark.gpu
def test_autograd_mirror_with_bias_and_groups(ops):
    """contrib.DeformableConvolution(data, offset, weight, bias, num_group=...) == the raw calls; under
    no_grad (or when nothing needs a gradient) the forward keeps no col matrix."""
    import torch
    from simpledet_amd import contrib
    torch.manual_seed(7)
    N, C, H, W, F, G = 2, 32, 12, 16, 16, 2
    x = torch.randn(N, C, H, W, device="cuda")
    off = torch.randn(N, 2 * 18, H, W, device="cuda")
    w = torch.randn(F, C // G, 3, 3, device="cuda") * 0.1
    b = torch.randn(F, device="cuda")
    a = dict(pad=1, stride=1, dilate=1, num_deformable_group=2, num_group=G)
    y = ops.deform_conv_forward(x, off, w, bias=b, **a)
    dy = torch.randn_like(y)
    g = ops.deform_conv_backward(dy, x, off, w, bias=True, **a)
    xr, offr, wr, br = (t.clone().requires_grad_(True) for t in (x, off, w, b))
    out = contrib.DeformableConvolution(xr, offr, wr, bias=br, kernel=(3, 3), pad=(1, 1), num_filter=F, num_group=G,
                                        num_deformable_group=2)
    out.backward(dy)
    close = lambda u, v: float((u - v).abs().max()) <= 1e-5 * max(1.0, float(v.abs().max()))
    assert close(out.detach(), y) and torch.equal(offr.grad, g[1]) and torch.equal(br.grad, g[3])
    assert close(xr.grad, g[0]) and close(wr.grad, g[2])
    with pytest.raises(ValueError):
        contrib.DeformableConvolution(x, off, w, bias=None, kernel=(3, 3), pad=(1, 1), num_group=G)   # no_bias=False, no bias
    # no gradient wanted: the col-free path (its workspace is a few MB, not N*C*9*Ho*Wo*4 bytes)
    calls = []
    real = ops.deform_conv_forward
    try:
        ops.deform_conv_forward = lambda *p, **k: (calls.append(k.get("keep_col", False)), real(*p, **k))[1]
        with torch.no_grad():
            contrib.DeformableConvolution(xr, offr, wr, bias=br, kernel=(3, 3), pad=(1, 1), num_group=G,
                                          num_deformable_group=2)
        contrib.DeformableConvolution(x, off, w, bias=b, kernel=(3, 3), pad=(1, 1), num_group=G, num_deformable_group=2)
        contrib.DeformableConvolution(xr, offr, wr, bias=br, kernel=(3, 3), pad=(1, 1), num_group=G,
                                      num_deformable_group=2)
    finally:
        ops.deform_conv_forward = real
    assert calls == [False, False, True]
