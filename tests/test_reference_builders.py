"""The reference's REAL builder and config code against the plugin (VERDICT r3 "Next 5"; the stand-in
for BASELINE configs[0] "plumbing"): `symbol/builder.py`, `models/FPN/builder.py`,
`models/maskrcnn/builder.py`, `models/dcn/builder.py` and four complete config files are imported
UNMODIFIED from /root/reference under graph-recording stand-ins of `mxnet` / `mxnext`
(tests/ref_stubs.py -- MXNet and mxnext are not installable here), `install()` is called, and the
symbols the reference code then builds are inspected: every hot-path operator must arrive as an
`sd_*` CustomOp node with the reference's own keyword arguments, and nothing of the reference's
per-level RoIAlign subgraph may remain.

CPU only.  Skipped where /root/reference is absent (the GPU box)."""
import collections
import importlib
import os

import pytest

from . import ref_stubs as RS

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="/root/reference not present")


def _install(R):
    from simpledet_amd import mxnet_plugin
    mxnet_plugin._state.update(registered=False)
    props = mxnet_plugin.install(R.mx)
    return mxnet_plugin, props


def _model_param(out):
    return [o for o in out if getattr(o, "__name__", "").startswith("ModelParam")][0]


def _ops(sym):
    return collections.Counter(n.op_type for n in RS.walk(sym).values())


def _one(sym, op_type):
    nodes = RS.find(sym, op_type)
    assert len(nodes) == 1, (op_type, len(nodes))
    return nodes[0]


# ------------------------------------------------------------------ whole config files ------------
def test_faster_r50v1_fpn_1x_train_and_test_symbols_route_to_the_plugin():
    """config/faster_r50v1_fpn_1x.py (BASELINE configs[0] / [3]): get_config(True/False) builds the
    detector through FasterRcnn.get_train_symbol / get_test_symbol (symbol/builder.py:42-95)."""
    with RS.reference_modules() as R:
        cfg = importlib.import_module("config.faster_r50v1_fpn_1x")
        plug, _ = _install(R)
        assert plug._state["fpn_patched"] is True
        # (this mxnext stand-in looks mx.sym.* up at call time: the probes see sd_* nodes, nothing is rebound)
        assert plug._state["mxnext_probe"]["roi_align"].startswith("late binding")
        train = _model_param(cfg.get_config(True)).train_symbol
        ops = _ops(train)
        # the FPN extractor: ONE fused node instead of assign + 4 x ROIAlign_v2 + add_n
        assert ops["sd_fpn_roi_align"] == 1 and ops["sd__contrib_ROIAlign_v2"] == 0
        assert ops["assign_layer_fpn"] == 0     # (the add_n nodes left are the FPN neck's top-down sums)
        node = _one(train, "sd_fpn_roi_align")
        assert node.params == {"rcnn_stride": "(4, 8, 16, 32)", "pooled_size": "(7, 7)",
                               "roi_canonical_scale": "224", "roi_canonical_level": "4"}
        assert [RS.source(i).op_type for i in node.inputs[:4]] == ["Convolution"] * 4   # P2..P5 of FPNNeck
        # ProposalTarget with the config's values (models/FPN/builder.py:347-363, config :73-87)
        pt = _one(train, "sd_ProposalTarget")
        assert pt.params["num_classes"] == "81" and pt.params["image_rois"] == "512"
        assert pt.params["batch_images"] == "2" and pt.params["fg_fraction"] == "0.25"
        assert pt.params["bbox_std"] == "(0.1, 0.1, 0.2, 0.2)" and pt.params["class_agnostic"] == "False"
        assert RS.source(node.inputs[4]) is pt                                        # rois = its output 0
        # Proposal_v3 on the coarse levels (this config sends strides < 32 to mxnext's nnvm proposal,
        # models/FPN/builder.py:289-313), then get_top_proposal -> the plugin's
        pv = RS.find(train, "sd__contrib_Proposal_v3")
        assert sorted(p.params["feature_stride"] for p in pv) == ["32", "64"]
        for p in pv:
            assert p.params["rpn_pre_nms_top_n"] == "2000" and p.params["rpn_post_nms_top_n"] == "2000"
            assert p.params["threshold"] == "0.7" and p.params["output_score"] == "True"
            assert p.params["scales"] == "(8,)" and p.params["ratios"] == "(0.5, 1.0, 2.0)"
            assert p.params["iou_loss"] == "False" and p.params["rpn_min_size"] == "0"
        top = _one(train, "sd_get_top_proposal")
        assert top.params == {"top_n": "2000"}
        test = _model_param(cfg.get_config(False)).test_symbol
        tops = _ops(test)
        assert tops["sd_fpn_roi_align"] == 1 and tops["sd__contrib_DecodeBBox"] == 1
        db = _one(test, "sd__contrib_DecodeBBox")
        assert db.params["bbox_std"] == "(0.1, 0.1, 0.2, 0.2)" and db.params["class_agnostic"] == "False"
        assert _one(test, "sd_get_top_proposal").params == {"top_n": "1000"}


def test_mask_r50v1_fpn_1x_symbols():
    """config/mask_r50v1_fpn_1x.py (BASELINE configs[4]): two fused extractors (7x7 box head, 14x14
    mask head) and ProposalMaskTarget (models/maskrcnn/builder.py:115-134)."""
    with RS.reference_modules() as R:
        cfg = importlib.import_module("config.mask_r50v1_fpn_1x")
        _install(R)
        train = _model_param(cfg.get_config(True)).train_symbol
        ext = RS.find(train, "sd_fpn_roi_align")
        assert sorted(e.params["pooled_size"] for e in ext) == ["(14, 14)", "(7, 7)"]
        mt = _one(train, "sd_ProposalMaskTarget")
        assert mt.params["mask_size"] == "28" and mt.params["output_iou"] == "True"
        assert mt.params["num_classes"] == "81" and mt.params["image_rois"] == "512"
        assert len(mt.inputs) == 3                                  # proposal, gt_bbox, gt_poly
        assert _ops(train)["sd__contrib_ROIAlign_v2"] == 0
        test = _model_param(cfg.get_config(False)).test_symbol
        assert _ops(test)["sd_fpn_roi_align"] == 2


def test_faster_r50v1c4_symbols_use_the_drop_in_roi_align():
    """config/faster_r50v1c4_c5_512roi_1x.py: the single-level RoiAlign (symbol/builder.py:879-898) ->
    X.roi_align -> ROIAlign_v2; RpnHead.get_all_proposal -> X.proposal -> Proposal_v3 (:241-254)."""
    with RS.reference_modules() as R:
        cfg = importlib.import_module("config.faster_r50v1c4_c5_512roi_1x")
        _install(R)
        train = _model_param(cfg.get_config(True)).train_symbol
        ra = _one(train, "sd__contrib_ROIAlign_v2")
        assert ra.params == {"pooled_size": "(7, 7)", "spatial_scale": "0.0625"}
        assert RS.source(ra.inputs[1]).op_type == "sd_ProposalTarget"
        pv = _one(train, "sd__contrib_Proposal_v3")
        assert pv.params["feature_stride"] == "16" and pv.params["output_score"] == "True"
        assert pv.params["rpn_pre_nms_top_n"] == "12000" or int(pv.params["rpn_pre_nms_top_n"]) > 0
        test = _model_param(cfg.get_config(False)).test_symbol
        assert _ops(test)["sd__contrib_ROIAlign_v2"] == 1 and _ops(test)["sd__contrib_DecodeBBox"] == 1


def test_dcn_config_emits_deformable_convolution_nodes():
    """config/dcn/faster_dcnv2_r50v1bc4_c5_512roi_1x.py: models/dcn/builder.py:14-17 calls
    mx.sym.contrib.DeformableConvolution(relu1, conv2_offset, kernel=(3,3), pad=(1,1),
    num_deformable_group=4, no_bias=True) in every special unit."""
    with RS.reference_modules() as R:
        cfg = importlib.import_module("config.dcn.faster_dcnv2_r50v1bc4_c5_512roi_1x")
        _install(R)
        train = _model_param(cfg.get_config(True)).train_symbol
        dcn = RS.find(train, "sd__contrib_DeformableConvolution")
        assert len(dcn) == 10
        for d in dcn:
            assert d.params["kernel"] == "(3, 3)" and d.params["pad"] == "(1, 1)"
            assert d.params["num_deformable_group"] == "4" and d.params["no_bias"] == "True"
            assert d.params["stride"] in ("(1, 1)", "(2, 2)")
            assert len(d.inputs) == 2 and RS.source(d.inputs[1]).op_type == "Convolution"   # the offset branch
            assert RS.source(d.inputs[1]).params["num_filter"] == 72
        assert {d.params["num_filter"] for d in dcn} <= {"128", "256", "512"}


# ------------------------------------------------------------------ single builder classes --------
def _roi_param(**kw):
    class RoiParam:
        fp16 = False
        out_size = 7
        stride = (4, 8, 16, 32)
        roi_canonical_scale = 224
        roi_canonical_level = 4
    for k, v in kw.items():
        setattr(RoiParam, k, v)
    return RoiParam


def test_real_fpn_roi_align_class_patched_and_original_kept():
    with RS.reference_modules() as R:
        mx = R.mx
        FB = importlib.import_module("models.FPN.builder")
        plug, _ = _install(R)
        feats = {"stride%d" % s: mx.sym.var("P%d" % s) for s in (4, 8, 16, 32)}
        rois = mx.sym.var("proposal")
        ext = FB.FPNRoiAlign(_roi_param())
        out = ext.get_roi_feature(dict(feats), rois)
        assert out.op_type == "reshape" and out.params["shape"] == (-3, -2)
        node = RS.source(out.inputs[0])
        assert node.op_type == "sd_fpn_roi_align" and node.inputs == [feats["stride4"], feats["stride8"],
                                                                     feats["stride16"], feats["stride32"], rois]
        # the untouched reference method still builds the reference subgraph -- through the aliased
        # drop-in op: assign -> 4 x ROIAlign_v2 -> reshape -> add_n (models/FPN/builder.py:573-605)
        ref = ext._sd_reference_get_roi_feature(dict(feats), rois)
        ops = _ops(ref)
        assert ref.op_type == "add_n" and ops["sd__contrib_ROIAlign_v2"] == 4 and ops["assign_layer_fpn"] == 1
        scales = sorted(float(n.params["spatial_scale"]) for n in RS.find(ref, "sd__contrib_ROIAlign_v2"))
        assert scales == [1 / 32.0, 1 / 16.0, 1 / 8.0, 1 / 4.0]
        # fp16 graphs: 7x7 / 14x14 use the native fp16 op (no casts), other sizes keep the casts
        out16 = FB.FPNRoiAlign(_roi_param(fp16=True, out_size=14)).get_roi_feature(dict(feats), rois)
        n16 = RS.source(out16.inputs[0])
        assert n16.params["fp16"] == "True" and n16.params["pooled_size"] == "(14, 14)"
        assert _ops(out16)["Cast"] == 0
        out5 = FB.FPNRoiAlign(_roi_param(fp16=True, out_size=5)).get_roi_feature(dict(feats), rois)
        assert out5.op_type == "Cast" and out5.params["dtype"] == "float16" and _ops(out5)["Cast"] == 5


@pytest.mark.parametrize("late", [True, False])
def test_mxnext_wrappers_are_aliased_explicitly(late):
    """install() -> patch_mxnext(): the routing must not depend on mxnext looking mx.sym.* up at call
    time.  late=False builds an mxnext stand-in that captured the ORIGINAL mx.sym constructors when it
    was imported (early binding); the reference's RoiAlign / RpnHead / BboxHead code must still emit
    sd_* nodes."""
    with RS.reference_modules(late_binding=late) as R:
        mx = R.mx
        SB = importlib.import_module("symbol.builder")
        plug, _ = _install(R)
        wrappers = {"mxnext.roi_align", "mxnext.proposal_target", "mxnext.proposal", "mxnext.decode_bbox"}
        assert "mxnext.tvm.get_top_proposal.get_top_proposal" in plug._state["mxnext_patched"]
        if late:    # call-time lookup: the probes find sd_* nodes already, nothing to rebind
            assert not wrappers & set(plug._state["mxnext_patched"])
            assert all(v.startswith("late binding") for v in plug._state["mxnext_probe"].values())
        else:       # import-time binding: the probes find the native operators of the same names -> rebound
            assert wrappers <= set(plug._state["mxnext_patched"])

        class RoiParam:
            fp16 = False
            out_size = 7
            stride = 16
        out = SB.RoiAlign(RoiParam).get_roi_feature(mx.sym.var("c4"), mx.sym.var("rois"))
        ra = _one(out, "sd__contrib_ROIAlign_v2")
        assert ra.params == {"pooled_size": "(7, 7)", "spatial_scale": "0.0625"}
        top = importlib.import_module("mxnext.tvm.get_top_proposal").get_top_proposal(
            mx.symbol, bbox=mx.sym.var("b"), score=mx.sym.var("s"), top_n=2000, batch_size=2)
        # (bbox, score), as models/FPN/builder.py:345 unpacks it; each a single-output symbol (a
        # multi-output symbol must never reach `rois=` -- the stand-in's operators now refuse one)
        b, s = top
        assert RS.source(b).op_type == "sd_get_top_proposal" and RS.source(b) is RS.source(s)
        assert (b.nout, b.index, s.nout, s.index) == (1, 0, 1, 1)
        with pytest.raises(TypeError):
            mx.sym.ProposalTarget(rois=RS.source(b), gt_boxes=mx.sym.var("gt"), num_classes=81, batch_images=1,
                                  image_rois=64, fg_thresh=0.5, bg_thresh_hi=0.5, bg_thresh_lo=0.0)


def test_a_wrapper_that_builds_another_operator_is_left_alone():
    """mxnext is not vendored, so `X.proposal` may bind `_contrib_Proposal` (proposal.cu: other +1
    convention, clamp and min-size filter than proposal_v3.cu).  patch_mxnext() probes the saved original
    and rebinds only a wrapper that builds the operator the plugin replaces under the same name."""
    with RS.reference_modules(late_binding=False) as R:
        mx, X = R.mx, R.X
        frozen_proposal = lambda **kw: RS.Symbol("Proposal", [v for v in kw.values() if isinstance(v, RS.Symbol)],
                                                 {k: v for k, v in kw.items() if not isinstance(v, RS.Symbol)},
                                                 kw.get("name"), 2)
        X.proposal = frozen_proposal
        broken = lambda **kw: (_ for _ in ()).throw(RuntimeError("tvm runtime missing"))
        X.decode_bbox = broken
        plug, _ = _install(R)
        assert X.proposal is frozen_proposal and "left alone" in plug._state["mxnext_probe"]["proposal"]
        assert X.decode_bbox is broken and plug._state["mxnext_probe"]["decode_bbox"].startswith("probe failed")
        assert "mxnext.roi_align" in plug._state["mxnext_patched"]
        # a second install() keeps the FIRST originals
        first = X._sd_reference_roi_align
        plug2, _ = _install(R)
        assert X._sd_reference_roi_align is first and not getattr(first, "_sd_alias", False)
        assert not getattr(mx.sym.contrib._sd_reference_ROIAlign_v2, "_sd_alias", False)
