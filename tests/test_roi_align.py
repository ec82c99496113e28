"""ROIAlign_v2 / fused FPN RoIAlign: oracle self-checks (CPU) and HIP-vs-oracle parity (GPU).

Reference: operator_cxx/contrib/roi_align_v2{-inl.h,.cc,.cu}, models/FPN/builder.py:567-610.
Bars: forward is compared BIT-EXACTLY (values and float argmax; both sides are built with
-ffp-contract=off and IEEE divide), backward within 1e-4 (north_star tolerance; summation order of
the scatter differs).
"""
import numpy as np
import pytest

from simpledet_amd import synth

from . import pyref

STRIDES = list(synth.FPN_STRIDES)


def small_case(seed, B=2, C=3, H=13, W=17, R=9, stride=16):
    rs = np.random.RandomState(seed)
    data = rs.standard_normal((B, C, H, W)).astype(np.float32)
    rois = synth.random_rois(seed, B, R, H * stride, W * stride, degenerate=False,
                             min_size=8, max_size=200)
    d = synth.degenerate_rois(H * stride, W * stride)
    rois[0, :4] = d[[0, 1, 2, 7]]
    rois[1, :3] = d[[3, 8, 9]]
    return data, rois


# ------------------------------------------------------------------------------------------ CPU --
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_matches_python_restatement(oracle, seed):
    data, rois = small_case(seed)
    o, ax, ay = oracle.roi_align_v2_fwd(data, rois, (7, 7), 1 / 16.0)
    po, pax, pay = pyref.roi_align_v2_fwd(data, rois, (7, 7), 1 / 16.0)
    np.testing.assert_array_equal(o, po)
    np.testing.assert_array_equal(ax, pax)
    np.testing.assert_array_equal(ay, pay)


def test_oracle_empty_and_degenerate_bins(oracle):
    data, rois = small_case(3)
    o, ax, ay = oracle.roi_align_v2_fwd(data, rois, (7, 7), 1 / 16.0)
    # zero box / outside / inverted boxes pool nothing: value 0, argmax -1
    for b, r in [(0, 0), (0, 1), (1, 0), (1, 1)]:
        assert np.all(o[b, r] == 0) and np.all(ax[b, r] == -1) and np.all(ay[b, r] == -1)
    # a pooled bin stores the float coordinates of its arg-max sample inside the feature map
    v = ax != -1
    assert v.any()
    assert ax[v].min() >= 0 and ax[v].max() <= data.shape[3] - 1
    assert ay[v].min() >= 0 and ay[v].max() <= data.shape[2] - 1
    assert np.all((ax == -1) == (ay == -1))


def test_oracle_backward_is_adjoint_of_forward_selection(oracle):
    """dX.sum() == sum of dY over pooled bins (bilinear weights sum to 1)."""
    data, rois = small_case(4)
    o, ax, ay = oracle.roi_align_v2_fwd(data, rois, (7, 7), 1 / 16.0)
    dy = np.random.RandomState(0).standard_normal(o.shape).astype(np.float32)
    dx = oracle.roi_align_v2_bwd(dy, ax, ay, data.shape)
    np.testing.assert_allclose(dx.sum(dtype=np.float64), dy[ax != -1].sum(dtype=np.float64),
                               rtol=1e-4, atol=1e-3)
    # kAddTo accumulates on top of the existing gradient
    dx2 = oracle.roi_align_v2_bwd(dy, ax, ay, data.shape, req=3, dx=dx.copy())
    np.testing.assert_allclose(dx2, 2 * dx, rtol=1e-6, atol=1e-6)


def test_oracle_cpu_gather_backward_diverges_on_degenerate_bins(oracle):
    """the reference's CPU backward is NOT the spec (the GPU scatter is).  On a
    RoI whose samples sit exactly on integer rows (hlow == hhigh) the CPU if/else-if chain drops
    half of the gradient; on ordinary RoIs the two agree."""
    rs = np.random.RandomState(0)
    data = rs.standard_normal((1, 1, 12, 12)).astype(np.float32)
    ok = np.array([[[17.3, 20.9, 130.2, 150.4]]], np.float32)
    o, ax, ay = oracle.roi_align_v2_fwd(data, ok, (7, 7), 1 / 16.0)
    dy = np.ones_like(o)
    g = oracle.roi_align_v2_bwd(dy, ax, ay, data.shape)
    c = oracle.roi_align_v2_bwd_cpu_gather(dy, ok, ax, ay, data.shape, 1 / 16.0)
    np.testing.assert_allclose(g, c, rtol=1e-5, atol=1e-6)
    # integer-aligned samples: 21 px tall bins of 3 px -> samples at integer + {1, 2}
    al = np.array([[[16.0, 16.0, 16.0 + 21 * 16, 16.0 + 21 * 16]]], np.float32)
    data2 = rs.standard_normal((1, 1, 40, 40)).astype(np.float32)
    o, ax, ay = oracle.roi_align_v2_fwd(data2, al, (7, 7), 1 / 16.0)
    assert np.all(ay == np.floor(ay)) and np.all(ax == np.floor(ax))
    dy = np.ones_like(o)
    g = oracle.roi_align_v2_bwd(dy, ax, ay, data2.shape)
    c = oracle.roi_align_v2_bwd_cpu_gather(dy, al, ax, ay, data2.shape, 1 / 16.0)
    assert abs(g.sum() - 49.0) < 1e-3          # spec: all of dY arrives
    assert c.sum() < 0.6 * g.sum()             # CPU gather drops the duplicate-corner terms


def test_oracle_fpn_assign_levels(oracle):
    # sqrt(area) = 224 -> level 4 (stride 16); 112 -> 3 (stride 8); tiny -> stride 4; huge -> 32
    def box(s):
        return [10, 10, 10 + s - 1, 10 + s - 1]
    rois = np.array([[box(224), box(112), box(20), box(1000), box(447), box(449), [5, 5, 1, 1],
                      [0, 0, 0, 0]]], np.float32)
    per, lvl = oracle.fpn_roi_assign(rois, STRIDES)
    assert lvl.tolist() == [[2, 1, 0, 3, 2, 3, 0, 0]]
    for l in range(4):
        m = lvl[0] == l
        np.testing.assert_array_equal(per[l][0][m], rois[0][m])
        assert np.all(per[l][0][~m] == 0)
    # negative area -> sqrt = nan -> matches no stride (all outputs zero)
    _, lvl = oracle.fpn_roi_assign(np.array([[[50, 50, 10, 200]]], np.float32), STRIDES)
    assert lvl.tolist() == [[-1]]


def test_oracle_fpn_equals_assigned_level(oracle):
    feats = synth.feature_maps(0, batch=1, channels=2)
    rois = synth.random_rois(5, 1, 24)
    out, ax, ay = oracle.fpn_roi_align_fwd(feats, rois, STRIDES, (7, 7))
    _, lvl = oracle.fpn_roi_assign(rois, STRIDES)
    for l, s in enumerate(STRIDES):
        o, x, y = oracle.roi_align_v2_fwd(feats[l], rois, (7, 7), 1.0 / s)
        m = lvl[0] == l
        np.testing.assert_array_equal(out[0][m], o[0][m] + np.float32(0))
        np.testing.assert_array_equal(ax[0][m], x[0][m])
        np.testing.assert_array_equal(ay[0][m], y[0][m])


# ------------------------------------------------------------------------------------------ GPU --
def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _assert_bwd_close(got, want, tol=1e-4):
    """north_star bar: |err| <= 1e-4 ELEMENTWISE (absolute; dY ~ N(0,1), gradient sums reach
    ~30), not relative to the largest gradient."""
    err = float(np.abs(got - want).max())
    assert err <= tol, "max abs err %g" % err


def _fwd_path(path):
    """the forward's three kernel families: "band" (default: band-resident kernel where the call has a
    workspace and the shape fits), "tiled" (what workspace-free calls and other shapes get), "naive"
    (per-element kernel: any pooled size)"""
    from simpledet_amd._lib import lib
    lib().set_tuning("roi_align_fwd", 0 if path == "naive" else 1)
    lib().set_tuning("roi_align_fwd_band", 0 if path == "tiled" else 1)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["band", "naive", "tiled"])
@pytest.mark.parametrize("case", ["small", "c4", "p5", "p2", "mask14", "odd_pool"])
def test_single_level_forward_bit_exact(ops, oracle, case, variant):
    from simpledet_amd._lib import lib
    _fwd_path(variant)
    try:
        if case == "small":
            data, rois = small_case(0, C=8)
            pooled, scale = (7, 7), 1 / 16.0
        elif case == "c4":   # single-level C4 family (config/faster_r50v1c4_c5_512roi_1x.py)
            data = synth.feature_maps(1, 2, 64, ((50, 84),))[0]
            rois = synth.random_rois(1, 2, 128)
            pooled, scale = (7, 7), 1 / 16.0
        elif case == "p5":   # a plane that is not whole 16-byte words (25 x 42): the whole-plane kernel's element fill
            data = synth.feature_maps(5, 2, 8, ((25, 42),))[0]
            rois = synth.random_rois(5, 2, 96)
            pooled, scale = (7, 7), 1 / 32.0
        elif case == "p2":   # large RoIs on the finest level: sparse sample grid
            data = synth.feature_maps(2, 1, 16, ((200, 334),))[0]
            rois = synth.random_rois(2, 1, 96)
            pooled, scale = (7, 7), 1 / 4.0
        elif case == "mask14":
            data = synth.feature_maps(3, 1, 8, ((50, 84),))[0]
            rois = synth.random_rois(3, 1, 64)
            pooled, scale = (14, 14), 1 / 16.0
        else:                # generic pooled size -> naive kernel
            data = synth.feature_maps(4, 1, 5, ((25, 42),))[0]
            rois = synth.random_rois(4, 1, 40)
            pooled, scale = (3, 5), 1 / 32.0
        want = oracle.roi_align_v2_fwd(data, rois, pooled, scale, nthreads=8)
        got = ops.roi_align_v2_forward(_t(data), _t(rois), pooled, scale)
        for g, w, name in zip(got, want, ("output", "maxidx_x", "maxidx_y")):
            np.testing.assert_array_equal(g.cpu().numpy(), w, err_msg=name)
    finally:
        _fwd_path("band")


@pytest.mark.gpu
def test_c4_whole_plane_forward_equals_band_kernel_at_full_size(ops):
    """roi_align_fwd_quad (round 5: the drop-in forward of the C4 family -- four channel-last planes per workgroup,
    a wave per RoI, 784-byte stores through wave-private LDS) against the band kernel (`roi_align_fwd_quad` = 0) at
    BASELINE's C4 size, degenerate RoIs included (three-sample bins and empty RoIs take the exact path inside
    either kernel): the same bits in all three outputs, run after run (the staging has no barrier: it relies on
    the in-order LDS of a wave)."""
    import torch
    from simpledet_amd._lib import lib
    data = torch.randn((2, 1024, 50, 84), device="cuda")
    rois = _t(synth.random_rois(21, 2, 512))
    lib().set_tuning("roi_align_fwd_quad", 0)
    try:
        want = [t.clone() for t in ops.roi_align_v2_forward(data, rois, (7, 7), 1 / 16.0)]
        assert "fwd_band" in (lib().cdll.sd_last_dispatch() or b"").decode()
    finally:
        lib().set_tuning("roi_align_fwd_quad", 1)
    for _ in range(3):
        got = ops.roi_align_v2_forward(data, rois, (7, 7), 1 / 16.0)
        assert "fwd_quad" in (lib().cdll.sd_last_dispatch() or b"").decode()
        for g, w, name in zip(got, want, ("output", "maxidx_x", "maxidx_y")):
            assert torch.equal(g, w), name
        del got


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [2, 1, 0])
@pytest.mark.parametrize("case", ["small", "p2_bands", "c4", "mask14", "odd_pool"])
def test_single_level_backward(ops, oracle, case, variant):
    from simpledet_amd._lib import lib
    lib().set_tuning("roi_align_bwd", variant)
    try:
        if case == "small":
            data, rois = small_case(0, C=8)
            pooled, scale = (7, 7), 1 / 16.0
        elif case == "p2_bands":   # plane larger than LDS -> row bands
            data = synth.feature_maps(2, 2, 8, ((200, 334),))[0]
            rois = synth.random_rois(2, 2, 96)
            pooled, scale = (7, 7), 1 / 4.0
        elif case == "c4":
            data = synth.feature_maps(1, 2, 64, ((50, 84),))[0]
            rois = synth.random_rois(1, 2, 128)
            pooled, scale = (7, 7), 1 / 16.0
        elif case == "mask14":
            data = synth.feature_maps(3, 1, 8, ((50, 84),))[0]
            rois = synth.random_rois(3, 1, 64)
            pooled, scale = (14, 14), 1 / 16.0
        else:
            data = synth.feature_maps(4, 1, 5, ((25, 42),))[0]
            rois = synth.random_rois(4, 1, 40)
            pooled, scale = (3, 5), 1 / 32.0
        o, ax, ay = oracle.roi_align_v2_fwd(data, rois, pooled, scale, nthreads=8)
        dy = np.random.RandomState(7).standard_normal(o.shape).astype(np.float32)
        want = oracle.roi_align_v2_bwd(dy, ax, ay, data.shape)
        got, d_rois = ops.roi_align_v2_backward(_t(dy), _t(rois), _t(ax), _t(ay), data.shape, scale)
        _assert_bwd_close(got.cpu().numpy(), want)
        assert d_rois.shape == rois.shape and float(d_rois.abs().max()) == 0.0
        # kAddTo on top of a non-zero gradient (roi_align_v2.cu:130-133)
        import torch
        base = np.random.RandomState(8).standard_normal(data.shape).astype(np.float32)
        acc = _t(base)
        ops.roi_align_v2_backward(_t(dy), _t(rois), _t(ax), _t(ay), data.shape, scale,
                                  req_data="add", req_rois="null", d_data=acc)
        torch.cuda.synchronize()
        _assert_bwd_close(acc.cpu().numpy(), want + base)
    finally:
        lib().set_tuning("roi_align_bwd", 2)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["c4", "mask14", "late_outlier", "inf", "piled", "nan_roi"])
def test_c4_backward_fixed_point_planes(ops, oracle, case):
    """roi_align_bwd_flt4_kernel (the drop-in backward of the C4 family: four channel planes of one image per
    workgroup) sums in 32-bit fixed point since round 5: against the oracle, against its own fp32
    compare-and-swap adds (`roi_align_bwd_fx` = 0), and bit-reproducible from run to run.
      late_outlier: one gradient 1000 x the rest that no thread sees in its first item -> the optimistic
                    scale fails its check and the planes are summed again with the exact maximum
      inf:          a non-finite gradient must reach its four pixels as the reference's float adds deliver it
      piled:        512 copies of one tiny RoI: per-pixel weight bound 25,088 > 2048 -> float adds
      nan_roi:      a RoI of NaNs (its arg-max planes are -1: nothing pooled) counts on the whole plane"""
    import torch
    from simpledet_amd._lib import lib
    pooled = (14, 14) if case == "mask14" else (7, 7)
    R = 96 if case == "mask14" else 512
    data = synth.feature_maps(11, 2, 8, ((50, 84),))[0]
    rois = synth.random_rois(12, 2, R)
    if case == "piled":
        rois[:] = np.array([400.0, 300.0, 415.0, 316.0], np.float32)
    if case == "nan_roi":
        rois[0, 5] = np.nan
    o, ax, ay = oracle.roi_align_v2_fwd(data, rois, pooled, 1 / 16.0, nthreads=8)
    dy = np.random.RandomState(13).standard_normal(o.shape).astype(np.float32)
    if case == "late_outlier":
        dy[1, R - 3, 5, 3, 3] = 4000.0
    if case == "inf":
        dy[0, 17, 2, 4, 4] = np.inf
    want = oracle.roi_align_v2_bwd(dy, ax, ay, data.shape)
    args = (_t(dy), _t(rois), _t(ax), _t(ay), data.shape, 1 / 16.0)
    got = ops.roi_align_v2_backward(*args)[0]
    again = ops.roi_align_v2_backward(*args)[0]
    lib().set_tuning("roi_align_bwd_fx", 0)
    try:
        flt = ops.roi_align_v2_backward(*args)[0]
    finally:
        lib().set_tuning("roi_align_bwd_fx", 1)
    g, f = got.cpu().numpy(), flt.cpu().numpy()
    # integer sums do not depend on the order of the adds (float adds do; round 6: a workgroup whose max|dY| is
    # beyond kFxUnitsPerMean x its mean |dY| -- the planes the 1000 x outlier lands in -- takes the float adds too)
    if case not in ("piled", "inf", "late_outlier"):
        assert torch.equal(got, again)
    fin = np.isfinite(want)
    np.testing.assert_array_equal(np.isfinite(g), fin)
    np.testing.assert_array_equal(g[~fin], want[~fin])
    # one unit is <= 2 max|dY| * 2048 * 2^-30: the elementwise bar scales with the largest gradient
    gmax = float(np.abs(dy[np.isfinite(dy)]).max())
    # (and with the sums themselves -- gradient sums reach ~30 in the other cases, hundreds when 512 RoIs coincide:
    # there the float adds in another order than the oracle's differ by more)
    tol = 1e-4 * max(1.0, gmax / 5.0, float(np.abs(want[fin]).max()) / 30.0)
    if case == "piled":   # 25,088 float adds per pixel in hardware order against the oracle's order: fp32-relative only
        tol = 3e-5 * float(np.abs(want[fin]).max())
    assert float(np.abs(g[fin] - want[fin]).max()) <= tol
    assert float(np.abs(f[fin] - want[fin]).max()) <= tol


@pytest.mark.gpu
def test_c4_backward_full_size_properties(ops):
    """The C4 backward at BASELINE's full size (2,1024,50,84) x 512 RoIs/img, through properties that need no oracle:
    run-to-run bit equality (integer sums), exact homogeneity for powers of two (the unit scales with max|dY|, so
    the same integers are added), additivity within the fixed-point unit, and conservation -- the four bilinear
    weights of a bin sum to one, so sum(dX) = sum of dY over the bins that pooled something."""
    import torch
    data = torch.randn((2, 1024, 50, 84), device="cuda")
    rois = _t(synth.random_rois(31, 2, 512))
    o, ax, ay = ops.roi_align_v2_forward(data, rois, (7, 7), 1 / 16.0)
    g = torch.Generator(device="cuda").manual_seed(5)
    dy1 = torch.randn(o.shape, device="cuda", generator=g)
    dy2 = torch.randn(o.shape, device="cuda", generator=g)
    bwd = lambda dy: ops.roi_align_v2_backward(dy, rois, ax, ay, tuple(data.shape), 1 / 16.0)[0]
    d1 = bwd(dy1)
    assert torch.equal(d1, bwd(dy1))
    assert torch.equal(bwd(dy1 * 4.0), d1 * 4.0) and torch.equal(bwd(dy1 * 0.125), d1 * 0.125)
    d2, d12 = bwd(dy2), bwd(dy1 + dy2)
    # one unit is <= 2 max|dY| x bound x 2^-30 (bound <= 2048): additivity holds to a few hundred units
    assert float((d12 - (d1 + d2)).abs().max()) <= 1e-3
    pooled = (ax != -1) & (ay != -1)
    want = float((dy1.double() * pooled).sum())
    got = float(d1.double().sum())
    assert abs(got - want) <= 1e-6 * float(dy1.double().abs().sum())


@pytest.mark.gpu
def test_backward_rejects_write_inplace(ops):
    import torch
    from simpledet_amd._lib import SimpleDetOpsError
    z = torch.zeros((1, 1, 1, 7, 7), device="cuda")
    r = torch.zeros((1, 1, 4), device="cuda")
    with pytest.raises(SimpleDetOpsError, match="kWriteInplace"):
        ops.roi_align_v2_backward(z, r, z, z, (1, 1, 8, 8), 0.25, req_data=2)


@pytest.mark.gpu
@pytest.mark.parametrize("rois_kind", ["random", "balanced"])
def test_fpn_fused_forward_backward_small(ops, oracle, rois_kind):
    feats = synth.feature_maps(0, batch=2, channels=16)
    rois = (synth.random_rois(1, 2, 64) if rois_kind == "random"
            else synth.level_balanced_rois(1, 2, 16))
    want = oracle.fpn_roi_align_fwd(feats, rois, STRIDES, (7, 7), nthreads=8)
    tf = [_t(f) for f in feats]
    got = ops.fpn_roi_align_forward(tf, _t(rois), STRIDES, (7, 7))
    for g, w, name in zip(got, want, ("output", "maxidx_x", "maxidx_y")):
        np.testing.assert_array_equal(g.cpu().numpy(), w, err_msg=name)
    dy = np.random.RandomState(3).standard_normal(want[0].shape).astype(np.float32)
    wd = oracle.fpn_roi_align_bwd(dy, rois, want[1], want[2], [f.shape for f in feats], STRIDES)
    gd = ops.fpn_roi_align_backward(_t(dy), _t(rois), got[1], got[2], [f.shape for f in feats],
                                    STRIDES)
    for g, w in zip(gd, wd):
        _assert_bwd_close(g.cpu().numpy(), w)


@pytest.mark.gpu
def test_fpn_fused_equals_reference_graph_of_four_ops(ops, oracle):
    """The fused op == fpn_roi_assign -> 4 x ROIAlign_v2 -> add_n (models/FPN/builder.py:573-605),
    all on the GPU through the drop-in per-level ops."""
    import torch
    feats = [_t(f) for f in synth.feature_maps(5, batch=2, channels=16)]
    rois = _t(synth.random_rois(6, 2, 64))
    per, level = ops.fpn_roi_assign(rois, STRIDES)
    _, olvl = oracle.fpn_roi_assign(rois.cpu().numpy(), STRIDES)
    np.testing.assert_array_equal(level.cpu().numpy(), olvl)
    total = None
    for f, p, s in zip(feats, per, STRIDES):
        o, _, _ = ops.roi_align_v2_forward(f, p.contiguous(), (7, 7), 1.0 / s)
        total = o if total is None else total + o
    fused, _, _ = ops.fpn_roi_align_forward(feats, rois, STRIDES, (7, 7))
    assert torch.equal(fused, total)


@pytest.mark.gpu
def test_fpn_full_size_baseline_config(ops, oracle):
    """BASELINE config: P2-P5, 256 ch, 800x1333, N=2, 512 RoIs/img, 7x7 -- bit-exact forward,
    1e-4 backward, at full size (the CPU oracle needs a few seconds on 8 cores)."""
    import torch
    feats = synth.feature_maps(0)
    rois = synth.random_rois(0)
    want = oracle.fpn_roi_align_fwd(feats, rois, STRIDES, (7, 7), nthreads=8)
    tf = [_t(f) for f in feats]
    got = ops.fpn_roi_align_forward(tf, _t(rois), STRIDES, (7, 7))
    for g, w, name in zip(got, want, ("output", "maxidx_x", "maxidx_y")):
        np.testing.assert_array_equal(g.cpu().numpy(), w, err_msg=name)
    dy = np.random.RandomState(11).standard_normal(want[0].shape).astype(np.float32)
    wd = oracle.fpn_roi_align_bwd(dy, rois, want[1], want[2], [f.shape for f in feats], STRIDES)
    gd = ops.fpn_roi_align_backward(_t(dy), _t(rois), got[1], got[2], [f.shape for f in feats],
                                    STRIDES)
    for g, w in zip(gd, wd):
        _assert_bwd_close(g.cpu().numpy(), w)
    # size-independent property: every pooled bin's gradient mass arrives exactly once
    mass = sum(float(g.double().sum()) for g in gd)
    want_mass = float(dy[want[1] != -1].astype(np.float64).sum())
    assert abs(mass - want_mass) <= 1e-3 * max(1.0, abs(want_mass))
    # the naive (reference-structure) kernels agree with the tiled/LDS ones
    from simpledet_amd._lib import lib
    lib().set_tuning("roi_align_fwd", 0)
    lib().set_tuning("roi_align_bwd", 0)
    try:
        got0 = ops.fpn_roi_align_forward(tf, _t(rois), STRIDES, (7, 7))
        for a, b in zip(got, got0):
            assert torch.equal(a, b)
        gd0 = ops.fpn_roi_align_backward(_t(dy), _t(rois), got[1], got[2],
                                         [f.shape for f in feats], STRIDES)
        for g, w in zip(gd0, wd):
            _assert_bwd_close(g.cpu().numpy(), w)
        # ... and so do the per-level LDS-plane kernels (fixed-point accumulators)
        lib().set_tuning("roi_align_bwd", 1)
        gd1 = ops.fpn_roi_align_backward(_t(dy), _t(rois), got[1], got[2],
                                         [f.shape for f in feats], STRIDES)
        for g, w in zip(gd1, wd):
            _assert_bwd_close(g.cpu().numpy(), w)
    finally:
        lib().set_tuning("roi_align_fwd", 1)
        lib().set_tuning("roi_align_bwd", 2)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["b8_r2000", "one_band", "few_channels"])
def test_fpn_band_forward_many_virtual_units(ops, oracle, case):
    """the band-resident forward cuts a unit's items into rounds ("virtual
    units"); with many images / RoIs, or every RoI in one band, there are far more virtual units
    than workgroups-at-start, and every one of them must still be written (float and packed)."""
    if case == "b8_r2000":       # ~90 units + ~190-260 extra rounds
        feats = synth.feature_maps(21, batch=8, channels=8)
        rois = synth.random_rois(22, 8, 2000)
    elif case == "one_band":     # 1500 small boxes inside the first rows of P2: one unit, many rounds
        feats = synth.feature_maps(23, batch=2, channels=16)
        rs = np.random.RandomState(24)
        rois = np.empty((2, 1500, 4), np.float32)
        for b in range(2):
            cx = rs.uniform(20, 1300, 1500)
            cy = rs.uniform(10, 110, 1500)
            w = rs.uniform(8, 60, 1500)
            h = rs.uniform(8, 40, 1500)
            rois[b] = np.stack([cx - w / 2, np.maximum(cy - h / 2, 0), cx + w / 2, cy + h / 2], 1)
    else:                        # C < 3 * planes-per-fill: units nobody starts on must still be taken
        feats = synth.feature_maps(25, batch=4, channels=8)
        rois = synth.level_balanced_rois(26, 4, 300)
    want = oracle.fpn_roi_align_fwd(feats, rois, STRIDES, (7, 7), nthreads=8)
    tf = [_t(f) for f in feats]
    got = ops.fpn_roi_align_forward(tf, _t(rois), STRIDES, (7, 7))
    for g, w, name in zip(got, want, ("output", "maxidx_x", "maxidx_y")):
        np.testing.assert_array_equal(g.cpu().numpy(), w, err_msg=name)
    outp, amp = ops.fpn_roi_align_forward_packed(tf, _t(rois), STRIDES, (7, 7))
    np.testing.assert_array_equal(outp.cpu().numpy(), want[0])
    # the packed arg-max names the same winners: the naive (per-element) kernel is the in-device
    # reference for the codes, the oracle for which bins pooled nothing
    import torch
    from simpledet_amd._lib import lib
    amn = ops.argmax_codes(amp[0], (7, 7))
    np.testing.assert_array_equal(amn.cpu().numpy() == 255, want[1] == -1)
    lib().set_tuning("roi_align_fwd", 0)
    try:
        out0, am0 = ops.fpn_roi_align_forward_packed(tf, _t(rois), STRIDES, (7, 7))
    finally:
        lib().set_tuning("roi_align_fwd", 1)
    assert torch.equal(out0, outp) and torch.equal(ops.argmax_codes(am0[0], (7, 7)), amn)
    assert torch.equal(am0[1], amp[1])


# ------------------------------------------------------------------- packed (one-byte) arg-max --
def _decode_packed(am, rois, feats_shapes, strides, level):
    """numpy float32 restatement of the device decode (sample_coord): (ax, ay) from (k, l)."""
    f = np.float32
    B, R, C, PH, PW = am.shape
    ax = -np.ones(am.shape, f)
    ay = -np.ones(am.shape, f)
    for b in range(B):
        for r in range(R):
            lv = level[b, r]
            if lv < 0:
                continue
            H, W = feats_shapes[lv][2], feats_shapes[lv][3]
            scale = f(1.0) / f(strides[lv])
            x1, y1, x2, y2 = [f(v) for v in rois[b, r]]

            def coord(p, pooled, s, e, size, k):
                rs, re = f(s * scale), f(e * scale)
                bn = f(f(re - rs) / f(pooled))
                lo = f(min(max(f(f(p) * bn) + rs, f(0)), f(size - 1)))
                hi = f(min(max(f(f(p + 1) * bn) + rs, f(0)), f(size - 1)))
                st = f(f(hi - lo) / f(3.0))
                step = f(max(st, f(0.01)))
                v = f(lo + st)
                for _ in range(k):
                    v = f(v + step)
                return v
            code = am[b, r]
            for p in range(PH):
                for q in range(PW):
                    cc = code[:, p, q]
                    for cv in np.unique(cc):
                        if cv == 255:
                            continue
                        m = cc == cv
                        ay[b, r, m, p, q] = coord(p, PH, y1, y2, H, int(cv) // 3)
                        ax[b, r, m, p, q] = coord(q, PW, x1, x2, W, int(cv) % 3)
    return ax, ay


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["band", "naive", "tiled"])
def test_fpn_packed_argmax_forward_backward(ops, oracle, variant):
    import torch
    from simpledet_amd._lib import lib
    feats = synth.feature_maps(5, batch=2, channels=16)
    rois = synth.random_rois(5, 2, 96)
    want = oracle.fpn_roi_align_fwd(feats, rois, STRIDES, (7, 7))
    _, level = oracle.fpn_roi_assign(rois, STRIDES)
    tf = [_t(f) for f in feats]
    _fwd_path(variant)
    try:
        out, am = ops.fpn_roi_align_forward_packed(tf, _t(rois), STRIDES, (7, 7))
    finally:
        _fwd_path("band")
    np.testing.assert_array_equal(out.cpu().numpy(), want[0])
    amn, con = ops.argmax_codes(am[0], (7, 7)).cpu().numpy(), am[1].cpu().numpy()
    np.testing.assert_array_equal(amn == 255, want[1] == -1)
    dax, day = _decode_packed(amn, rois, [f.shape for f in feats], STRIDES, level)
    np.testing.assert_array_equal(dax, want[1])  # decoded coordinates are the forward's floats
    np.testing.assert_array_equal(day, want[2])
    # ... and so is the device's coordinate table, entry by entry where a code points into it
    B, R = rois.shape[:2]
    pp, qq = np.meshgrid(np.arange(7), np.arange(7), indexing="ij")
    code = amn.astype(np.int64)
    ok = code != 255
    ty = np.take_along_axis(con[:, :, None, None, None, :21].repeat(16, 2).repeat(7, 3).repeat(7, 4),
                            np.where(ok, pp[None, None, None] * 3 + code // 3, 0)[..., None], -1)[..., 0]
    tx = np.take_along_axis(con[:, :, None, None, None, 21:42].repeat(16, 2).repeat(7, 3).repeat(7, 4),
                            np.where(ok, qq[None, None, None] * 3 + code % 3, 0)[..., None], -1)[..., 0]
    np.testing.assert_array_equal(np.where(ok, ty, -1), want[2])
    np.testing.assert_array_equal(np.where(ok, tx, -1), want[1])
    dy = np.random.RandomState(6).standard_normal(want[0].shape).astype(np.float32)
    wd = oracle.fpn_roi_align_bwd(dy, rois, want[1], want[2], [f.shape for f in feats], STRIDES)
    gd = ops.fpn_roi_align_backward_packed(_t(dy), _t(rois), am, [f.shape for f in feats], STRIDES)
    for g, w in zip(gd, wd):
        _assert_bwd_close(g.cpu().numpy(), w)
    # kAddTo
    acc = [torch.ones_like(g) for g in gd]
    ops.fpn_roi_align_backward_packed(_t(dy), _t(rois), am, [f.shape for f in feats], STRIDES,
                                      req_data="add", d_feats=acc)
    for g, w in zip(acc, wd):
        _assert_bwd_close(g.cpu().numpy(), w + 1)


@pytest.mark.gpu
def test_fpn_packed_equals_float_argmax_path_full_size(ops):
    """Baseline shapes: the packed path and the float arg-max path give the same output and the
    same gradients (size-independent cross-check, no oracle needed)."""
    import torch
    feats = [_t(f) for f in synth.feature_maps(1, batch=2, channels=64)]
    rois = _t(synth.random_rois(1, 2, 512))
    o1, mx, my = ops.fpn_roi_align_forward(feats, rois, STRIDES, (7, 7))
    o2, am = ops.fpn_roi_align_forward_packed(feats, rois, STRIDES, (7, 7))
    assert torch.equal(o1, o2)
    assert torch.equal(ops.argmax_codes(am[0], (7, 7)) == 255, mx == -1)
    dy = torch.randn_like(o1)
    shapes = [f.shape for f in feats]
    g1 = ops.fpn_roi_align_backward(dy, rois, mx, my, shapes, STRIDES)
    g2 = ops.fpn_roi_align_backward_packed(dy, rois, am, shapes, STRIDES)
    for a, b in zip(g1, g2):
        assert float((a - b).abs().max()) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["tiled", "naive"])
def test_forward_variants_agree_at_full_size(ops, variant):
    """Baseline shapes (2 x 512 RoIs x 256 channels, degenerate RoIs included): the tiled fallback kernel
    and the naive kernel give the default kernel's bits, in both the packed and the float arg-max
    form."""
    import torch
    from simpledet_amd._lib import lib
    feats = [_t(f) for f in synth.feature_maps(2, batch=2, channels=256)]
    rois = _t(synth.random_rois(2, 2, 512))
    o1, am1 = ops.fpn_roi_align_forward_packed(feats, rois, STRIDES, (7, 7))
    f1 = ops.fpn_roi_align_forward(feats, rois, STRIDES, (7, 7))
    _fwd_path(variant)
    try:
        o2, am2 = ops.fpn_roi_align_forward_packed(feats, rois, STRIDES, (7, 7))
        f2 = ops.fpn_roi_align_forward(feats, rois, STRIDES, (7, 7))
    finally:
        _fwd_path("band")
    assert torch.equal(o1, o2)
    assert torch.equal(ops.argmax_codes(am1[0], (7, 7)), ops.argmax_codes(am2[0], (7, 7)))
    # the coordinate table has entries only for RoIs assigned to a level (the rest is never read)
    lvl = ops.fpn_roi_assign(rois, STRIDES)[1].reshape(-1) >= 0
    assert torch.equal(am1[1].reshape(lvl.numel(), -1)[lvl], am2[1].reshape(lvl.numel(), -1)[lvl])
    for x, y in zip(f1, f2):
        assert torch.equal(x, y)


@pytest.mark.gpu
def test_contrib_fpn_roi_align_autograd(ops):
    import torch
    from simpledet_amd import contrib
    feats = [_t(f).requires_grad_(True) for f in synth.feature_maps(2, batch=1, channels=8)]
    rois = _t(synth.random_rois(2, 1, 64))
    out = contrib.fpn_roi_align(feats, rois, STRIDES, (7, 7))
    out.sum().backward()
    o2, mx, my = ops.fpn_roi_align_forward([f.detach() for f in feats], rois, STRIDES, (7, 7))
    g = ops.fpn_roi_align_backward(torch.ones_like(o2), rois, mx, my, [f.shape for f in feats], STRIDES)
    for f, w in zip(feats, g):
        assert float((f.grad - w).abs().max()) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("pooled,num", [((7, 7), 512), ((14, 14), 128)])
def test_packed_backward_is_bit_reproducible_and_within_1e4_at_full_size(ops, oracle, pooled, num):
    """BASELINE configs[1] / [4] shapes at reduced channels: the fixed-point band accumulation makes
    the backward independent of the order of the LDS adds (two runs agree bit for bit) and stays
    within 1e-4 of the oracle elementwise; kAddTo and non-finite gradients (float fallback: inf
    must reach the pixels it touches) behave like the reference scatter."""
    import torch
    feats = synth.feature_maps(9, batch=2, channels=8)
    rois = synth.random_rois(9, 2, num)
    want = oracle.fpn_roi_align_fwd(feats, rois, STRIDES, pooled, nthreads=8)
    tf = [_t(f) for f in feats]
    out, am = ops.fpn_roi_align_forward_packed(tf, _t(rois), STRIDES, pooled)
    np.testing.assert_array_equal(out.cpu().numpy(), want[0])
    dy = np.random.RandomState(10).standard_normal(want[0].shape).astype(np.float32)
    shapes = [f.shape for f in feats]
    wd = oracle.fpn_roi_align_bwd(dy, rois, want[1], want[2], shapes, STRIDES)
    g1 = ops.fpn_roi_align_backward_packed(_t(dy), _t(rois), am, shapes, STRIDES)
    g2 = ops.fpn_roi_align_backward_packed(_t(dy), _t(rois), am, shapes, STRIDES)
    for a, b, w in zip(g1, g2, wd):
        assert torch.equal(a, b), "backward differs between two runs"
        _assert_bwd_close(a.cpu().numpy(), w)
    # a tiny and a huge gradient scale: the per-workgroup scale follows max|dY|
    for mul in (1e-6, 1e6):
        gs = ops.fpn_roi_align_backward_packed(_t(dy * np.float32(mul)), _t(rois), am, shapes, STRIDES)
        for a, w in zip(gs, wd):
            assert np.abs(a.cpu().numpy() / np.float32(mul) - w).max() <= 1e-4
    # heavy tails: gradients 100x larger on every other RoI defeat the optimistic scale taken from
    # the first RoIs of a band; the band is then accumulated again with the exact maximum
    dyo = dy.copy()
    dyo[:, 1::2] *= np.float32(100.0)
    wo = oracle.fpn_roi_align_bwd(dyo, rois, want[1], want[2], shapes, STRIDES)
    go1 = ops.fpn_roi_align_backward_packed(_t(dyo), _t(rois), am, shapes, STRIDES)
    go2 = ops.fpn_roi_align_backward_packed(_t(dyo), _t(rois), am, shapes, STRIDES)
    for a, b, w in zip(go1, go2, wo):
        # (two populations a factor 100 apart sit at the edge of what one fixed-point unit resolves: exponent spread
        # ~7 bits + the bits of the weight bound against kFxRangeBits = 15.  Workgroups on the far side take the
        # float adds -- hardware order, equal up to float rounding from run to run -- the others integer sums)
        assert float((a - b).abs().max()) <= 1e-3
        assert np.abs(a.cpu().numpy() - w).max() <= 1e-4 * 100
    # inf / nan propagate to exactly the pixels the oracle sends them to
    dyn = dy.copy()
    valid = np.argwhere(want[1] != -1)
    b, r, c, p, q = valid[len(valid) // 2]
    dyn[b, r, c, p, q] = np.inf
    wn = oracle.fpn_roi_align_bwd(dyn, rois, want[1], want[2], shapes, STRIDES)
    gn = ops.fpn_roi_align_backward_packed(_t(dyn), _t(rois), am, shapes, STRIDES)
    for a, w in zip(gn, wn):
        a = a.cpu().numpy()
        np.testing.assert_array_equal(np.isfinite(a), np.isfinite(w))
        m = np.isfinite(w)
        assert np.abs(a[m] - w[m]).max() <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("pooled,num,channels", [((7, 7), 512, 256), ((14, 14), 128, 64), ((7, 7), 96, 16)])
@pytest.mark.parametrize("knobs", [{}, {"roi_align_fwd_band": 0}, {"roi_align_bwd_lists": 2}])
def test_planned_pair_one_prepass_equals_the_unplanned_pair(ops, pooled, num, channels, knobs):
    """sd_fpn_roi_align_fwd_packed_plan / _bwd_packed_plan: both rois-only pre-passes in the forward's
    single pre-pass launch (3 launches per step instead of 4).  Same bits as the _ws pair: output,
    arg-max codes, coordinate table, gradients (the planned backward reads band lists / tap tables
    whose sample coordinates were recomputed from rois instead of read from the forward's table).
    knobs: the forward on its fallback kernels (the plan's list pre-pass then runs stand-alone), and a
    setting under which the tap-table form does not apply (both calls fall back by the same decision)."""
    import torch
    from simpledet_amd._lib import lib
    feats = [_t(f) for f in synth.feature_maps(41, batch=2, channels=channels)]
    rois = _t(synth.random_rois(42, 2, num))
    shapes = [f.shape for f in feats]
    o0, st0 = ops.fpn_roi_align_forward_packed(feats, rois, STRIDES, pooled)
    torch.manual_seed(43)
    dy = torch.randn_like(o0)
    g0 = ops.fpn_roi_align_backward_packed(dy, rois, st0, shapes, STRIDES)
    for k, v in knobs.items():
        lib().set_tuning(k, v)
    try:
        o1, st1 = ops.fpn_roi_align_forward_packed(feats, rois, STRIDES, pooled, plan=True)
        assert len(st1) == 3
        g1 = ops.fpn_roi_align_backward_packed(dy, rois, st1, shapes, STRIDES)
        # the plan is op state: a second backward from the same plan (e.g. gradient accumulation)
        acc = [torch.ones_like(g) for g in g1]
        ops.fpn_roi_align_backward_packed(dy, rois, st1, shapes, STRIDES, req_data="add", d_feats=acc)
    finally:
        for k in knobs:
            lib().set_tuning(k, 1)
    assert torch.equal(o1, o0)
    assert torch.equal(ops.argmax_codes(st1[0], pooled), ops.argmax_codes(st0[0], pooled))   # (rows are padded)
    lvl = ops.fpn_roi_assign(rois, STRIDES)[1].reshape(-1) >= 0
    assert torch.equal(st1[1].reshape(lvl.numel(), -1)[lvl], st0[1].reshape(lvl.numel(), -1)[lvl])
    same_tables = "roi_align_bwd_lists" not in knobs     # (lists-only mode has other bands: other fixed-point scales)
    for a, b, c in zip(g1, g0, acc):
        if same_tables:
            assert torch.equal(a, b)
        else:   # two fixed-point scales, each within ~5e-5 of the exact sum
            assert float((a - b).abs().max()) <= 2e-4
        assert float((c - 1 - a).abs().max()) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("pooled,num", [((7, 7), 512), ((14, 14), 128)])
def test_packed_backward_workspace_modes(ops, oracle, pooled, num):
    """The packed backward with the workspace pre-pass (1: per-band RoI lists + tap tables, 27 KB
    bands; 2: lists only, 36 KB bands) and without (0: every workgroup builds its list).  Lists
    only == none bit for bit (same bands, same arithmetic); the tap-table mode cuts the planes into
    different bands, hence other fixed-point scales: every mode is within 1e-4 of the exact sums (the
    oracle), elementwise, and bit-reproducible."""
    import torch
    from simpledet_amd._lib import lib
    feats = [_t(f) for f in synth.feature_maps(8, batch=2, channels=64)]
    rois = _t(synth.random_rois(8, 2, num))
    out, am = ops.fpn_roi_align_forward_packed(feats, rois, STRIDES, pooled)
    dy = _t(np.random.RandomState(81).standard_normal(tuple(out.shape)).astype(np.float32))
    shapes = [f.shape for f in feats]
    fw = oracle.fpn_roi_align_fwd([f.cpu().numpy() for f in feats], rois.cpu().numpy(), STRIDES, pooled, nthreads=8)
    want = oracle.fpn_roi_align_bwd(dy.cpu().numpy(), rois.cpu().numpy(), fw[1], fw[2], shapes, STRIDES, nthreads=8)
    res = {}
    # (mode, roi_align_bwd_pixbound): round 6's list pre-pass bounds the weight per pixel of a band, which gives the
    # workspace modes a finer fixed-point unit than the workspace-free call (whose workgroups sum the bound over
    # the band's RoIs themselves); with the per-pixel bound off, lists-only and no workspace are the same sums
    for mode, pix in ((1, 1), (2, 1), (2, 0), (0, 1)):
        lib().set_tuning("roi_align_bwd_lists", mode)
        lib().set_tuning("roi_align_bwd_pixbound", pix)
        try:
            res[mode, pix] = ops.fpn_roi_align_backward_packed(dy, rois, am, shapes, STRIDES)
            again = ops.fpn_roi_align_backward_packed(dy, rois, am, shapes, STRIDES)
        finally:
            lib().set_tuning("roi_align_bwd_lists", 1)
            lib().set_tuning("roi_align_bwd_pixbound", 1)
        for a, b in zip(res[mode, pix], again):
            assert torch.equal(a, b), "mode %d is not bit-reproducible" % mode
    for a, b in zip(res[2, 0], res[0, 1]):
        assert torch.equal(a, b)
    # every mode against the exact sums: the north_star bar, elementwise
    for key in res:
        for a, w in zip(res[key], want):
            _assert_bwd_close(a.cpu().numpy(), w)


@pytest.mark.gpu
def test_fused_fpn_roi_align_geometry_fuzz(ops, oracle):
    """40 random geometries: image sizes from 70 to 900 px (feature maps down to 3x3), 1..3 images,
    channel counts that are not multiples of the slice count (1, 3, 20, 33, 64), 1..77 RoIs incl. the
    degenerate ones, 7x7 and 14x14: packed and float arg-max forward bit-exact, both backwards
    within 1e-4, integer-coordinate boxes (coincident taps) included."""
    rs = np.random.RandomState(4242)
    for it in range(40):
        B = int(rs.randint(1, 4))
        C = int(rs.choice([1, 3, 8, 20, 33, 64]))
        ih, iw = int(rs.randint(70, 900)), int(rs.randint(70, 900))
        shapes = [(-(-ih // s), -(-iw // s)) for s in STRIDES]
        if min(min(s) for s in shapes) < 2:
            continue
        R = int(rs.choice([1, 2, 3, 7, 31, 64, 77]))
        pooled = (14, 14) if it % 5 == 4 else (7, 7)
        feats = synth.feature_maps(500 + it, batch=B, channels=C, shapes=shapes)
        rois = synth.random_rois(600 + it, B, R, ih, iw, degenerate=bool(rs.randint(0, 2)), min_size=4.0,
                                 max_size=float(max(ih, iw)))
        if rs.randint(0, 2):
            rois = np.round(rois / 4) * 4  # sample coordinates on integers: left == right taps
        msg = "problem %d B=%d C=%d image %dx%d R=%d pooled=%s" % (it, B, C, ih, iw, R, pooled)
        want = oracle.fpn_roi_align_fwd(feats, rois, STRIDES, pooled, nthreads=8)
        tf = [_t(f) for f in feats]
        got = ops.fpn_roi_align_forward(tf, _t(rois), STRIDES, pooled)
        for g, w, name in zip(got, want, ("output", "maxidx_x", "maxidx_y")):
            np.testing.assert_array_equal(g.cpu().numpy(), w, err_msg=msg + " " + name)
        out, am = ops.fpn_roi_align_forward_packed(tf, _t(rois), STRIDES, pooled)
        np.testing.assert_array_equal(out.cpu().numpy(), want[0], err_msg=msg + " packed output")
        np.testing.assert_array_equal(ops.argmax_codes(am[0], pooled).cpu().numpy() == 255, want[1] == -1,
                                      err_msg=msg + " packed codes")
        dy = rs.standard_normal(want[0].shape).astype(np.float32)
        shp = [f.shape for f in feats]
        wd = oracle.fpn_roi_align_bwd(dy, rois, want[1], want[2], shp, STRIDES)
        # 1e-4 absolute where the gradient sums stay in the baseline's range (|dX| <~ 32); where
        # hundreds of bins pile onto one pixel of a 3x3 map the fp32 sum itself is only defined to
        # ~|dX| * 1e-6 * sqrt(#adds) (the reference's atomics reorder it from run to run), so the
        # bar scales with the magnitude there
        for which, gd in (("float", ops.fpn_roi_align_backward(_t(dy), _t(rois), got[1], got[2], shp, STRIDES)),
                          ("packed", ops.fpn_roi_align_backward_packed(_t(dy), _t(rois), am, shp, STRIDES))):
            for g, w in zip(gd, wd):
                if not w.size:
                    continue
                err = float(np.abs(g.cpu().numpy() - w).max())
                tol = 1e-4 * max(1.0, float(np.abs(w).max()) / 32.0)
                assert err <= tol, msg + " %s backward max abs err %g (|dX| max %g)" % (which, err, np.abs(w).max())


@pytest.mark.gpu
@pytest.mark.parametrize("pooled,num,channels", [((7, 7), 96, 16), ((14, 14), 40, 8), ((7, 7), 512, 64)])
def test_fpn_packed_forward_fp16_io(ops, oracle, pooled, num, channels):
    """fp16 feature maps in, fp16 output out, fp32 arithmetic: bit-equal to what an fp16 graph
    computes with the reference's casts around the op (X.to_fp32 -> ROIAlign_v2 -> X.to_fp16,
    models/FPN/builder.py:581-586, 607-608); the packed arg-max decodes to the oracle's floats."""
    import torch
    feats16 = [f.astype(np.float16) for f in synth.feature_maps(5, batch=2, channels=channels)]
    rois = synth.random_rois(5, 2, num)
    want = oracle.fpn_roi_align_fwd([f.astype(np.float32) for f in feats16], rois, STRIDES, pooled, nthreads=8)
    out, (am, coords) = ops.fpn_roi_align_forward_packed_f16([_t(f) for f in feats16], _t(rois), STRIDES, pooled)
    assert out.dtype == torch.float16
    np.testing.assert_array_equal(out.cpu().numpy(), want[0].astype(np.float16))
    # the same state as the fp32 op on the converted maps
    o32, (am32, c32) = ops.fpn_roi_align_forward_packed([_t(f.astype(np.float32)) for f in feats16], _t(rois),
                                                         STRIDES, pooled)
    assert torch.equal(ops.argmax_codes(am, pooled), ops.argmax_codes(am32, pooled))
    lvl = ops.fpn_roi_assign(_t(rois), STRIDES)[1].reshape(-1) >= 0
    assert torch.equal(coords.reshape(lvl.numel(), -1)[lvl], c32.reshape(lvl.numel(), -1)[lvl])
    assert torch.equal(out, o32.half())
    # where the band-resident kernel is not eligible the fp16 op still works
    # (cast -> fp32 op -> cast instead of SD_ERR_UNSUPPORTED), with the same bits
    from simpledet_amd._lib import lib
    lib().set_tuning("roi_align_fwd_band", 0)
    try:
        out_fb, (am_fb, _) = ops.fpn_roi_align_forward_packed_f16([_t(f) for f in feats16], _t(rois),
                                                                  STRIDES, pooled)
    finally:
        lib().set_tuning("roi_align_fwd_band", 1)
    assert torch.equal(out_fb, out)
    assert torch.equal(ops.argmax_codes(am_fb, pooled), ops.argmax_codes(am, pooled))


@pytest.mark.gpu
@pytest.mark.parametrize("pooled,num,channels", [((7, 7), 128, 16), ((14, 14), 40, 8)])
def test_fpn_packed_backward_fp16_io(ops, oracle, pooled, num, channels):
    """fp16 gradient in, fp16 gradients out: the fp32 sums (within 1e-4 of the oracle's) rounded to
    fp16 -- at most one fp16 step from the rounded exact sum wherever 1e-4 crosses a rounding boundary."""
    import torch
    feats16 = [f.astype(np.float16) for f in synth.feature_maps(6, batch=2, channels=channels)]
    rois = synth.random_rois(6, 2, num)
    out, am = ops.fpn_roi_align_forward_packed_f16([_t(f) for f in feats16], _t(rois), STRIDES, pooled)
    dy16 = np.random.RandomState(7).standard_normal(tuple(out.shape)).astype(np.float16)
    shapes = [f.shape for f in feats16]
    g16 = ops.fpn_roi_align_backward_packed_f16(_t(dy16), _t(rois), am, shapes, STRIDES)
    fw = oracle.fpn_roi_align_fwd([f.astype(np.float32) for f in feats16], rois, STRIDES, pooled, nthreads=8)
    want = oracle.fpn_roi_align_bwd(dy16.astype(np.float32), rois, fw[1], fw[2], shapes, STRIDES, nthreads=8)
    for g, w in zip(g16, want):
        assert g.dtype == torch.float16
        got = g.cpu().numpy().astype(np.float32)
        step = np.maximum(np.abs(w) * 2.0 ** -10, 2.0 ** -24)   # one fp16 step at the value's magnitude
        assert np.all(np.abs(got - w) <= 1e-4 + step)
    # the kernel's fp16-I/O instance == the casts around the fp32 kernel, bit for bit (write and add)
    gc = ops.fpn_roi_align_backward_packed_f16(_t(dy16), _t(rois), am, shapes, STRIDES, native=False)
    for a_, b_ in zip(g16, gc):
        assert torch.equal(a_, b_)
    acc_n = [torch.full(tuple(s), 0.5, device="cuda", dtype=torch.float16) for s in shapes]
    acc_c = [a_.clone() for a_ in acc_n]
    ops.fpn_roi_align_backward_packed_f16(_t(dy16), _t(rois), am, shapes, STRIDES, req_data="add", d_feats=acc_n)
    ops.fpn_roi_align_backward_packed_f16(_t(dy16), _t(rois), am, shapes, STRIDES, req_data="add", d_feats=acc_c,
                                          native=False)
    for a_, b_ in zip(acc_n, acc_c):
        assert torch.equal(a_, b_)
    # req = add accumulates into fp16 gradients
    acc = [torch.ones(tuple(s), device="cuda", dtype=torch.float16) for s in shapes]
    ops.fpn_roi_align_backward_packed_f16(_t(dy16), _t(rois), am, shapes, STRIDES, req_data="add", d_feats=acc)
    for a_, w in zip(acc, want):
        step = np.maximum(np.abs(w + 1) * 2.0 ** -10, 2.0 ** -24)
        assert np.all(np.abs(a_.cpu().numpy().astype(np.float32) - (w + 1)) <= 1e-4 + 2 * step)
