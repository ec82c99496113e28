"""MXNet CustomOp adapter (simpledet_amd/mxnet_plugin.py) driven through a stub of mx.operator:
CPU tests cover registration, the reference's argument/output names, visible-output counts,
parameter parsing and shape inference; the GPU test runs forward/backward through the adapter and
compares with the oracle."""
import numpy as np
import pytest

from . import mx_stub


@pytest.fixture()
def plugin():
    from simpledet_amd import mxnet_plugin
    mx = mx_stub.make_stub()
    props = mxnet_plugin.install(mx)
    return mx, props, mxnet_plugin


def test_registers_reference_operator_names(plugin):
    mx, props, _ = plugin
    assert set(props) == {"_contrib_ROIAlign_v2", "ROIPooling_v1", "ProposalTarget",
                          "_contrib_GenAnchor", "_contrib_NMS", "assign_layer_fpn",
                          "_contrib_DeformableConvolution", "fpn_roi_align", "_contrib_Proposal_v3",
                          "get_top_proposal", "_contrib_DecodeBBox", "ProposalTarget_v2", "ProposalMaskTarget"}
    for name in props:
        assert "sd_" + name in mx.registry
    # aliases on the symbol namespaces the reference graph uses
    for attr in ("ROIAlign_v2", "GenAnchor", "NMS", "DeformableConvolution"):
        assert callable(getattr(mx.sym.contrib, attr))
    assert callable(mx.sym.ROIPooling_v1) and callable(mx.sym.ProposalTarget)


def test_roi_align_prop_matches_reference_registration(plugin):
    _, props, _ = plugin
    p = props["_contrib_ROIAlign_v2"](pooled_size="(7, 7)", spatial_scale="0.0625")
    assert p.list_arguments() == ["data", "rois"]                     # roi_align_v2.cc:179-182
    assert p.list_outputs() == ["output", "maxidx_x", "maxidx_y"]     # :183-186
    assert p.num_visible_outputs == 1                                 # :175-178
    ins, outs = p.infer_shape([(2, 256, 50, 84), (2, 512, 4)])
    assert outs == [(2, 512, 256, 7, 7)] * 3                          # :195-208
    with pytest.raises(ValueError, match="3D tensor"):
        p.infer_shape([(2, 256, 50, 84), (1024, 5)])
    with pytest.raises(ValueError, match="pooled_size"):
        props["_contrib_ROIAlign_v2"](pooled_size="(0, 7)", spatial_scale="0.5")
    deps = p.declare_backward_dependency(["dy"], ["data", "rois"], ["out", "mx", "my"])
    assert deps == ["dy", "rois", "mx", "my"]                         # roi_align_v2-inl.h:206-218


def test_proposal_target_prop(plugin):
    _, props, _ = plugin
    kw = dict(num_classes="81", batch_images="2", image_rois="512", fg_thresh="0.5",
              bg_thresh_hi="0.5", bg_thresh_lo="0.0", fg_fraction="0.25",
              bbox_std="(0.1, 0.1, 0.2, 0.2)")
    p = props["ProposalTarget"](**kw)
    assert p.list_arguments() == ["rois", "gt_boxes"]
    assert p.list_outputs() == ["roi_output", "label", "bbox_target", "bbox_weight", "match_gt_iou"]
    assert p.num_visible_outputs == 4                                 # proposal_target-inl.h:297-303
    assert props["ProposalTarget"](output_iou="True", **kw).num_visible_outputs == 5
    _, outs = p.infer_shape([(2, 2000, 4), (2, 100, 5)])
    assert outs == [(2, 512, 4), (2, 512), (2, 512, 324), (2, 512, 324), (2, 512)]
    assert p.declare_backward_dependency([], [], []) == []


def test_other_props_shapes(plugin):
    _, props, _ = plugin
    p = props["ROIPooling_v1"](pooled_size="(7,7)", spatial_scale="0.0625")
    assert p.infer_shape([(2, 1024, 50, 84), (300, 5)])[1] == [(300, 1024, 7, 7)] * 2
    p = props["_contrib_GenAnchor"](scales="(8,)", ratios="(0.5, 1, 2)", feature_stride="4")
    assert p.infer_shape([(2, 6, 200, 334)])[1] == [(200 * 334 * 3, 4)]
    p = props["_contrib_NMS"](rpn_pre_nms_top_n="2000", rpn_post_nms_top_n="1000", threshold="0.7")
    assert p.list_outputs() == ["output", "score"] and p.num_visible_outputs == 1
    assert p.infer_shape([(2, 2000, 5)])[1] == [(2, 1000, 4), (2, 1000, 1)]
    # NMSProp::InferShape (nms-inl.h:90-103) declares (B, post, .) whatever the input count
    assert p.infer_shape([(2, 500, 5)])[1] == [(2, 1000, 4), (2, 1000, 1)]
    p = props["assign_layer_fpn"](rcnn_stride="(4, 8, 16, 32)", roi_canonical_scale="224",
                                  roi_canonical_level="4")
    assert p.list_outputs() == ["rois_s4", "rois_s8", "rois_s16", "rois_s32"]
    p = props["_contrib_DeformableConvolution"](kernel="(3,3)", num_filter="256", pad="(1,1)",
                                                num_deformable_group="4", no_bias="True")
    ins, outs = p.infer_shape([(2, 256, 50, 84), None, None])
    assert ins[1] == (2, 72, 50, 84) and ins[2] == (256, 256, 3, 3) and outs == [(2, 256, 50, 84)]
    assert p.list_arguments() == ["data", "offset", "weight"]
    # upstream's defaults: no_bias = False -> a fourth argument (models/RepPoints/builder.py:215-226);
    # num_group: weight (F, C / G, kh, kw) (models/sepc/sepc_dconv.py:12-16)
    p = props["_contrib_DeformableConvolution"](kernel="(3,3)", num_filter="8", pad="(1,1)", num_group="2")
    assert p.list_arguments() == ["data", "offset", "weight", "bias"]
    ins, outs = p.infer_shape([(2, 16, 10, 12), None, None, None])
    assert ins == [(2, 16, 10, 12), (2, 18, 10, 12), (8, 8, 3, 3), (8,)] and outs == [(2, 8, 10, 12)]
    assert p.declare_backward_dependency(["dy"], ["x", "o", "w", "b"], ["y"]) == ["dy", "x", "o", "w"]
    with pytest.raises(ValueError, match="group"):
        props["_contrib_DeformableConvolution"](kernel="(3,3)", num_filter="8", num_group="3").infer_shape(
            [(2, 16, 10, 12), None, None, None])
    # what the kernels do not take is refused by the prop (install()'s alias never gets here: it hands
    # such calls back to the native constructor, tests/test_reference_config_sweep.py)
    with pytest.raises(ValueError, match="square"):
        props["_contrib_DeformableConvolution"](kernel="(3,3)", num_filter="8", stride="(1,2)")
    sup = props["_contrib_DeformableConvolution"].sd_supports
    assert sup(dict(kernel="(3, 3)", num_filter="8", no_bias="False", num_group="4")) == ""
    assert "layout" in sup(dict(kernel="(3, 3)", num_filter="8", layout="NHWC"))
    assert "not one" in sup(dict(kernel="(3, 3)", num_filter="8", cudnn_tune="off"))


def test_proposal_v3_prop(plugin):
    _, props, _ = plugin
    p = props["_contrib_Proposal_v3"](rpn_pre_nms_top_n="2000", rpn_post_nms_top_n="2000",
                                      feature_stride="16", scales="(8,)", ratios="(0.5, 1, 2)",
                                      output_score="True", threshold="0.7", rpn_min_size="0")
    assert p.list_arguments() == ["cls_prob", "bbox_pred", "im_info"]   # proposal_v3-inl.h:255-257
    assert p.list_outputs() == ["output", "score"] and p.num_visible_outputs == 2
    ins, outs = p.infer_shape([(2, 6, 50, 84), None, None])
    assert ins[1] == (2, 12, 50, 84) and ins[2] == (2, 3)
    assert outs == [(2, 2000, 4), (2, 2000, 1)]
    t = props["get_top_proposal"](top_n="2000")
    assert t.infer_shape([(2, 10000, 4), (2, 10000, 1)])[1] == [(2, 2000, 4), (2, 2000, 1)]
    # ... and the registration of the reference's own class (models/FPN/get_top_proposal.py:46-72),
    # recorded by tests/golden/make_golden.py
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "top_proposal.npz"))
    assert t.list_arguments() == list(g["iface_arguments"])
    assert t.list_outputs() == list(g["iface_outputs"])
    ins, outs = t.infer_shape([(2, 10000, 4), (2, 10000, 1)])
    assert [list(x) for x in outs] == g["iface_infer_out"].tolist()
    assert [list(x) + [0] * (3 - len(x)) for x in ins] == g["iface_infer_in"].tolist()
    assert bool(t.need_top_grad_) == bool(g["iface_need_top_grad"])
    assert len(t.declare_backward_dependency([], [], [])) == int(g["iface_backward_dependency"])


def test_fused_fpn_roi_align_prop(plugin):
    _, props, _ = plugin
    p = props["fpn_roi_align"](rcnn_stride="(4, 8, 16, 32)")
    assert p.list_arguments() == ["data_s4", "data_s8", "data_s16", "data_s32", "rois"]
    assert p.num_visible_outputs == 1
    shapes = [(2, 256, 200, 334), (2, 256, 100, 167), (2, 256, 50, 84), (2, 256, 25, 42), (2, 512, 4)]
    # 7x7: output + the op's private state (one-byte arg-max, per-RoI coordinate / tap table, and the
    # backward's band lists / tap tables that the forward's single rois-only pre-pass builds)
    assert p.list_outputs() == ["output", "argmax", "coords", "plan"]
    out_shapes = p.infer_shape(shapes)[1]
    assert out_shapes[:3] == [(2, 512, 256, 7, 7), (2, 512, 256, 52), (2, 512, 126)]
    assert len(out_shapes[3]) == 1 and 1 << 20 < out_shapes[3][0] < 64 << 20 and out_shapes[3][0] % 16 == 0
    assert [np.dtype(t) for t in p.infer_type([np.float32] * 5)[1]] == [np.float32, np.uint8, np.float32, np.uint8]
    assert p.declare_backward_dependency(["dy"], ["a", "b", "c", "d", "rois"], ["o", "am", "co", "plan"]) == \
        ["dy", "rois", "am", "co", "plan"]
    q = props["fpn_roi_align"](rcnn_stride="(4, 8, 16, 32)", pooled_size="(5, 5)")
    assert q.list_outputs() == ["output", "maxidx_x", "maxidx_y"]
    assert q.infer_shape(shapes)[1] == [(2, 512, 256, 5, 5)] * 3


def test_fused_fpn_roi_align_prop_fp16(plugin):
    """fp16 = True: feature maps fp16, rois fp32 -> output fp16, the private state keeps its types"""
    _, props, _ = plugin
    p = props["fpn_roi_align"](rcnn_stride="(4, 8, 16, 32)", pooled_size="(14, 14)", fp16="True")
    tin, tout, _ = p.infer_type([np.float16] * 4 + [np.float32])
    assert [np.dtype(t) for t in tin] == [np.dtype(np.float16)] * 4 + [np.dtype(np.float32)]
    assert [np.dtype(t) for t in tout] == [np.dtype(np.float16), np.dtype(np.uint8), np.dtype(np.float32),
                                           np.dtype(np.uint8)]
    with pytest.raises(ValueError):
        props["fpn_roi_align"](rcnn_stride="(4, 8, 16, 32)", pooled_size="(5, 5)", fp16="True")


def test_symbol_alias_builds_custom_node_with_visible_outputs(plugin):
    mx, _, _ = plugin
    d, r = mx.sym.Variable("data"), mx.sym.Variable("rois")
    s = mx.sym.contrib.ROIAlign_v2(data=d, rois=r, pooled_size=(7, 7), spatial_scale=0.25,
                                   name="roi_align")
    assert s[0] == "out" and s[1].op_type == "sd__contrib_ROIAlign_v2" and s[2] == 0
    assert s[1].params == {"pooled_size": "(7, 7)", "spatial_scale": "0.25"}


def test_proposal_mask_target_alias_as_the_reference_graph_calls_it(plugin):
    """models/maskrcnn/builder.py:115-134 calls mx.sym.ProposalMaskTarget with three positional
    symbols and never passes num_args (MXNet's front end derives key_var_num_args from the inputs,
    proposal_mask_target.cc:485): the alias must build the node, and the six visible outputs."""
    mx, props, _ = plugin
    proposal, gt_bbox, gt_poly = (mx.sym.Variable(n) for n in ("proposal", "gt_bbox", "gt_poly"))
    outs = mx.sym.ProposalMaskTarget(
        proposal, gt_bbox, gt_poly, mask_size=28, num_classes=81, class_agnostic=False, batch_images=2,
        proposal_without_gt=False, image_rois=512, fg_fraction=0.25, fg_thresh=0.5, bg_thresh_hi=0.5,
        bg_thresh_lo=0.0, bbox_weight=(1.0, 1.0, 1.0, 1.0), bbox_mean=(0.0, 0.0, 0.0, 0.0),
        bbox_std=(0.1, 0.1, 0.2, 0.2), output_iou=True, name="subsample_proposal")
    node = outs if not isinstance(outs, list) else outs[0][1]
    assert node.op_type == "sd_ProposalMaskTarget" and len(node.inputs) == 3
    assert "num_args" not in node.params
    p = props["ProposalMaskTarget"](**node.params)
    assert p.num_args == 3 and p.list_arguments() == ["rois", "gt_boxes", "gt_polys"]
    assert len(p.list_outputs()) == 6
    with pytest.raises(ValueError):
        props["ProposalMaskTarget"](num_args="4", **node.params)


def test_proposal_mask_target_output_ratio_as_msrcnn_calls_it(plugin):
    """models/msrcnn/builder.py:219-237 unpacks seven outputs (output_iou=True, output_ratio=True):
    the seventh is mask_ratio (B, FG), proposal_mask_target-inl.h:453-456."""
    mx, props, _ = plugin
    proposal, gt_bbox, gt_poly = (mx.sym.Variable(n) for n in ("proposal", "gt_bbox", "gt_poly"))
    outs = mx.sym.ProposalMaskTarget(
        proposal, gt_bbox, gt_poly, mask_size=28, num_classes=81, class_agnostic=False, batch_images=2,
        proposal_without_gt=False, image_rois=512, fg_fraction=0.25, fg_thresh=0.5, bg_thresh_hi=0.5,
        bg_thresh_lo=0.0, bbox_weight=(1.0, 1.0, 1.0, 1.0), bbox_mean=(0.0, 0.0, 0.0, 0.0),
        bbox_std=(0.1, 0.1, 0.2, 0.2), output_iou=True, output_ratio=True, name="subsample_proposal")
    node = outs if not isinstance(outs, list) else outs[0][1]
    p = props["ProposalMaskTarget"](**node.params)
    assert p.list_outputs()[-1] == "mask_ratio" and len(p.list_outputs()) == 7 and p.num_visible_outputs == 7
    _, shapes = p.infer_shape([(2, 2000, 4), (2, 100, 5), (2, 100, 500)])
    assert shapes[5] == (2, 128, 28, 28) and shapes[6] == (2, 128)


def test_install_routes_fpn_extractor_to_the_fused_op_without_editing_the_reference(plugin):
    """models/FPN/builder.py:567-610 builds assign + 4 x roi_align + add_n; install() rebinds
    FPNRoiAlign.get_roi_feature so the same call emits one sd_fpn_roi_align node."""
    import sys
    import types
    mx, _, plug = plugin

    class FPNRoiAlign:  # stand-in with the reference class's name, attribute and method
        def __init__(self, p):
            self.p = p

        def get_roi_feature(self, conv_fpn_feat, proposal):
            return "reference subgraph"

    fake = types.ModuleType("models.FPN.builder")
    fake.FPNRoiAlign = FPNRoiAlign
    pkgs = {"models": types.ModuleType("models"), "models.FPN": types.ModuleType("models.FPN"),
            "models.FPN.builder": fake}
    old = {k: sys.modules.get(k) for k in pkgs}
    sys.modules.update(pkgs)
    try:
        assert plug.patch_fpn_roi_align(mx=mx) is True
        p = types.SimpleNamespace(stride=(4, 8, 16, 32), roi_canonical_scale=224, roi_canonical_level=4,
                                  out_size=7, fp16=False)
        feats = {"stride%d" % s: mx.sym.Variable("P%d" % s) for s in p.stride}
        rois = mx.sym.Variable("proposal")
        node = FPNRoiAlign(p).get_roi_feature(feats, rois)
        assert node[0] == "reshape" and node[2] == (-3, -2)
        _, custom, idx = node[1]
        assert idx == 0 and custom.op_type == "sd_fpn_roi_align"
        assert custom.inputs == [feats["stride4"], feats["stride8"], feats["stride16"], feats["stride32"], rois]
        assert custom.params == {"rcnn_stride": "(4, 8, 16, 32)", "pooled_size": "(7, 7)",
                                 "roi_canonical_scale": "224", "roi_canonical_level": "4"}
        assert FPNRoiAlign(p)._sd_reference_get_roi_feature(feats, rois) == "reference subgraph"
        # fp16 graphs, 7x7 / 14x14: the op reads and writes fp16 itself (sd_fpn_roi_align_fwd_packed_f16),
        # the cast nodes of builder.py:581-586, 607-608 are gone
        p.fp16 = True
        node = FPNRoiAlign(p).get_roi_feature(feats, rois)
        assert node[0] == "reshape"
        custom = node[1][1]
        assert custom.params["fp16"] == "True" and custom.inputs[0] is feats["stride4"]
        # other pooling sizes keep the reference's casts around the fp32 op
        p.out_size = 5
        node = FPNRoiAlign(p).get_roi_feature(feats, rois)
        assert node[0] == "cast" and node[2] == "float16" and node[1][0] == "reshape"
        assert node[1][1][1].inputs[0] == ("cast", feats["stride4"], "float32")
        assert "fp16" not in node[1][1][1].params
    finally:
        for k, v in old.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


@pytest.mark.gpu
def test_adapter_forward_backward_on_gpu(plugin, oracle):
    import torch
    from simpledet_amd import synth
    mx, props, _ = plugin
    w = mx_stub.wrap
    rs = np.random.RandomState(0)
    data = rs.standard_normal((2, 8, 25, 42)).astype(np.float32)
    rois = synth.random_rois(0, 2, 32)
    prop = props["_contrib_ROIAlign_v2"](pooled_size="(7,7)", spatial_scale=str(1 / 32.0))
    _, oshape = prop.infer_shape([data.shape, rois.shape])
    op = prop.create_operator(None, None, None)
    tin = [w(torch.from_numpy(data).cuda()), w(torch.from_numpy(rois).cuda())]
    tout = [w(torch.empty(s, device="cuda")) for s in oshape]
    op.forward(True, ["write"] * 3, tin, tout, [])
    want = oracle.roi_align_v2_fwd(data, rois, (7, 7), 1 / 32.0)
    for g, x in zip(tout, want):
        np.testing.assert_array_equal(g.t.cpu().numpy(), x)
    dy = rs.standard_normal(oshape[0]).astype(np.float32)
    grads = [w(torch.empty(data.shape, device="cuda")), w(torch.ones(rois.shape, device="cuda"))]
    op.backward(["write", "write"], [w(torch.from_numpy(dy).cuda())], tin, tout, grads, [])
    wdx = oracle.roi_align_v2_bwd(dy, want[1], want[2], data.shape)
    np.testing.assert_allclose(grads[0].t.cpu().numpy(), wdx, rtol=1e-4, atol=1e-4)
    assert float(grads[1].t.abs().max()) == 0
    # ProposalTarget through the adapter == oracle with the never-seeded libc stream
    prois, gt = synth.proposal_target_inputs(3, 2, 600, 40)
    pp = props["ProposalTarget"](num_classes="81", batch_images="2", image_rois="128",
                                 fg_thresh="0.5", bg_thresh_hi="0.5", bg_thresh_lo="0.0")
    op = pp.create_operator(None, None, None)
    _, oshape = pp.infer_shape([prois.shape, gt.shape])
    tout = [w(torch.empty(s, device="cuda")) for s in oshape]
    op.forward(True, ["write"] * 5, [w(torch.from_numpy(prois).cuda()), w(torch.from_numpy(gt).cuda())],
               tout, [])
    want = oracle.proposal_target(prois, gt, oracle.make_pt_param(81, 2, 128), rng=oracle.GlibcRand(1))
    np.testing.assert_array_equal(tout[0].t.cpu().numpy(), want[0])
    np.testing.assert_array_equal(tout[1].t.cpu().numpy(), want[1])
    # GenAnchor + NMS + assign through the adapter
    ga = props["_contrib_GenAnchor"](scales="(8,)", ratios="(0.5,1,2)", feature_stride="16")
    op = ga.create_operator(None, None, None)
    o = [w(torch.empty((50 * 84 * 3, 4), device="cuda"))]
    op.forward(False, ["write"], [w(torch.empty((1, 6, 50, 84), device="cuda"))], o, [])
    np.testing.assert_array_equal(o[0].t.cpu().numpy(), oracle.gen_anchor(50, 84, 16, [8], [0.5, 1, 2]))
    dets = np.stack([synth.nms_dets(1, 500), synth.nms_dets(2, 500)])
    nm = props["_contrib_NMS"](rpn_pre_nms_top_n="500", rpn_post_nms_top_n="100", threshold="0.7")
    op = nm.create_operator(None, None, None)
    o = [w(torch.empty((2, 100, 4), device="cuda")), w(torch.empty((2, 100, 1), device="cuda"))]
    op.forward(False, ["write"] * 2, [w(torch.from_numpy(dets).cuda())], o, [])
    wn = oracle.nms(dets, 500, 100, 0.7)
    np.testing.assert_array_equal(o[0].t.cpu().numpy(), wn[0])


@pytest.mark.gpu
@pytest.mark.parametrize("pooled", [(7, 7), (5, 5)])
def test_fused_fpn_roi_align_adapter_on_gpu(plugin, oracle, pooled):
    """The fused CustomOp (packed state for 7x7, float arg-max planes otherwise) against the oracle."""
    import torch
    from simpledet_amd import synth
    _, props, _ = plugin
    w = mx_stub.wrap
    strides = [4, 8, 16, 32]
    feats = synth.feature_maps(3, batch=2, channels=8)
    rois = synth.random_rois(3, 2, 40)
    want = oracle.fpn_roi_align_fwd(feats, rois, strides, pooled)
    prop = props["fpn_roi_align"](rcnn_stride=str(tuple(strides)), pooled_size=str(pooled))
    _, oshape = prop.infer_shape([f.shape for f in feats] + [rois.shape])
    _, otype, _ = prop.infer_type([np.float32] * 5)
    tmap = {np.dtype(np.float32): torch.float32, np.dtype(np.uint8): torch.uint8}
    op = prop.create_operator(None, None, None)
    tin = [w(torch.from_numpy(f).cuda()) for f in feats] + [w(torch.from_numpy(rois).cuda())]
    tout = [w(torch.empty(s, device="cuda", dtype=tmap[np.dtype(t)])) for s, t in zip(oshape, otype)]
    op.forward(True, ["write"] * 3, tin, tout, [])
    np.testing.assert_array_equal(tout[0].t.cpu().numpy(), want[0])
    dy = np.random.RandomState(1).standard_normal(want[0].shape).astype(np.float32)
    wd = oracle.fpn_roi_align_bwd(dy, rois, want[1], want[2], [f.shape for f in feats], strides)
    gin = [w(torch.empty_like(t.t)) for t in tin]
    op.backward(["write"] * 5, [w(torch.from_numpy(dy).cuda())], tin, tout, gin, [])
    for g, wv in zip(gin[:-1], wd):
        assert np.abs(g.t.cpu().numpy() - wv).max() <= 1e-4
    assert float(gin[-1].t.abs().max()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [dict(no_bias="True", G=1), dict(no_bias="False", G=1), dict(no_bias="False", G=2)])
def test_deformable_convolution_adapter_on_gpu(plugin, oracle, cfg):
    """The DeformableConvolution CustomOp as MXNet would drive it: arguments data, offset, weight[, bias],
    training forward (col kept) -> backward with the kept col, inference forward (col-free), req = add on
    the bias gradient; against the oracle's DeformableConvolutionOp restatement."""
    import torch
    _, props, _ = plugin
    w = mx_stub.wrap
    G, has_b = cfg["G"], cfg["no_bias"] == "False"
    rs = np.random.RandomState(5)
    N, C, H, W, F, dg = 2, 32, 12, 16, 16, 2
    x = rs.standard_normal((N, C, H, W)).astype(np.float32)
    off = (rs.standard_normal((N, dg * 18, H, W)) * 1.5).astype(np.float32)
    wt = (rs.standard_normal((F, C // G, 3, 3)) * 0.2).astype(np.float32)
    b = rs.standard_normal(F).astype(np.float32) if has_b else None
    prop = props["_contrib_DeformableConvolution"](kernel="(3, 3)", num_filter=str(F), pad="(1, 1)",
                                                   num_deformable_group=str(dg), num_group=str(G),
                                                   no_bias=cfg["no_bias"])
    ishape, oshape = prop.infer_shape([x.shape] + [None] * (3 if has_b else 2))
    assert ishape[1] == off.shape and ishape[2] == wt.shape
    op = prop.create_operator(None, None, None)
    tin = [w(torch.from_numpy(a).cuda()) for a in ((x, off, wt, b) if has_b else (x, off, wt))]
    want = oracle.deform_convolution_fwd(x, off, wt, b, pad=1, stride=1, dil=1, dgroup=dg, num_group=G)
    bar = lambda v: 1e-4 * max(1.0, float(np.abs(v).max()) / 32.0)
    for is_train in (False, True):
        tout = [w(torch.empty(oshape[0], device="cuda"))]
        op.forward(is_train, ["write"], tin, tout, [])
        assert float(np.abs(tout[0].t.cpu().numpy() - want).max()) <= bar(want), is_train
    assert op._fwd_ws is not None                      # training: the col matrix waits for the backward
    dy = rs.standard_normal(want.shape).astype(np.float32)
    wg = oracle.deform_convolution_bwd(dy, x, off, wt, bias=has_b, pad=1, stride=1, dil=1, dgroup=dg, num_group=G)
    grads = [w(torch.empty(a.shape, device="cuda")) for a in ((x, off, wt, b) if has_b else (x, off, wt))]
    req = ["write"] * 3
    if has_b:
        grads[3].t.fill_(1.5)
        req.append("add")
    op.backward(req, [w(torch.from_numpy(dy).cuda())], tin, tout, grads, [])
    assert op._fwd_ws is None
    for i, (g_, r_) in enumerate(zip(grads, wg)):
        r_ = r_ + 1.5 if i == 3 else r_
        assert float(np.abs(g_.t.cpu().numpy() - r_).max()) <= bar(r_), i


@pytest.mark.gpu
def test_adapter_ops_back_to_back_on_a_side_stream_without_the_per_op_synchronise(oracle):
    """VERDICT r5 "Next 10": `install(stream=..., sync=False)` -- the adapter launches on the stream it is given and
    does not synchronise per op.  Three operators issued back to back on a non-default torch stream (fused FPN
    RoIAlign forward, its backward, NMS), nothing waited for in between, the results read after ONE stream
    synchronise: equal to the oracle.  So the library itself needs no synchronisation and no NULL-stream
    ordering; the default's per-op synchronise is the CustomOp hook's contract, not the kernels'."""
    import torch
    from simpledet_amd import mxnet_plugin, synth
    from simpledet_amd._lib import lib
    mx = mx_stub.make_stub()
    side = torch.cuda.Stream()
    seen = []

    def cur():
        seen.append(torch.cuda.current_stream().cuda_stream)
        return seen[-1]
    props = mxnet_plugin.install(mx, stream=cur, sync=False)
    try:
        w = mx_stub.wrap
        strides = [4, 8, 16, 32]
        feats = synth.feature_maps(3, batch=2, channels=8)
        rois = synth.random_rois(3, 2, 40)
        dets = np.stack([synth.nms_dets(1, 500), synth.nms_dets(2, 500)])
        dy = np.random.RandomState(1).standard_normal((2, 40, 8, 7, 7)).astype(np.float32)
        prop = props["fpn_roi_align"](rcnn_stride=str(tuple(strides)), pooled_size="(7, 7)")
        _, oshape = prop.infer_shape([f.shape for f in feats] + [rois.shape])
        _, otype, _ = prop.infer_type([np.float32] * 5)
        tmap = {np.dtype(np.float32): torch.float32, np.dtype(np.uint8): torch.uint8}
        nm = props["_contrib_NMS"](rpn_pre_nms_top_n="500", rpn_post_nms_top_n="100", threshold="0.7")
        synced = []
        real = lib().cdll.sd_stream_synchronize
        with torch.cuda.stream(side):
            tin = [w(torch.from_numpy(f).cuda()) for f in feats] + [w(torch.from_numpy(rois).cuda())]
            tout = [w(torch.empty(s, device="cuda", dtype=tmap[np.dtype(t)])) for s, t in zip(oshape, otype)]
            gin = [w(torch.empty_like(t.t)) for t in tin]
            tdy, tdets = w(torch.from_numpy(dy).cuda()), w(torch.from_numpy(dets).cuda())
            o = [w(torch.empty((2, 100, 4), device="cuda")), w(torch.empty((2, 100, 1), device="cuda"))]
            op, op2 = prop.create_operator(None, None, None), nm.create_operator(None, None, None)
            n0 = len(seen)
            op.forward(True, ["write"] * 3, tin, tout, [])
            op.backward(["write"] * 5, [tdy], tin, tout, gin, [])
            op2.forward(False, ["write"] * 2, [tdets], o, [])
        assert len(seen) > n0 and all(s == side.cuda_stream for s in seen[n0:]) and side.cuda_stream != 0
        side.synchronize()
        want = oracle.fpn_roi_align_fwd(feats, rois, strides, (7, 7))
        np.testing.assert_array_equal(tout[0].t.cpu().numpy(), want[0])
        wd = oracle.fpn_roi_align_bwd(dy, rois, want[1], want[2], [f.shape for f in feats], strides)
        for g, wv in zip(gin[:-1], wd):
            assert np.abs(g.t.cpu().numpy() - wv).max() <= 1e-4
        np.testing.assert_array_equal(o[0].t.cpu().numpy(), oracle.nms(dets, 500, 100, 0.7)[0])
    finally:
        mxnet_plugin._state["stream"], mxnet_plugin._state["sync"] = None, True


def test_install_stream_and_sync_parameters_are_kept_and_default_to_the_null_stream():
    from simpledet_amd import mxnet_plugin
    mx = mx_stub.make_stub()
    mxnet_plugin.install(mx)
    assert mxnet_plugin._state["stream"] is None and mxnet_plugin._state["sync"] is True
    assert mxnet_plugin._stream() is None
    mxnet_plugin.install(mx, stream=0x1234, sync=False)
    try:
        assert mxnet_plugin._stream().value == 0x1234 and mxnet_plugin._state["sync"] is False
        mxnet_plugin.install(mx, stream=lambda: 77)
        assert mxnet_plugin._stream().value == 77 and mxnet_plugin._state["sync"] is True
    finally:
        mxnet_plugin.install(mx)
