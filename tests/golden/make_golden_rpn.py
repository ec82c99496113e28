#!/usr/bin/env python
"""Golden fixtures for the RPN anchor-target assignment, produced by THE REFERENCE'S OWN CLASSES.

Run in the build container (needs /root/reference and oracle/_ref/bbox*.so from oracle/build_ref.py):
    python tests/golden/make_golden_rpn.py
Executed from the reference, where it lies (functions/classes loaded by name with `ast`, nothing
copied): core/detection_input.py AnchorTarget2D (:345-565), models/FPN/input.py
PyramidAnchorTarget2DBase / PyramidAnchorTarget2D (:9-146), operator_py/bbox_transform.py
nonlinear_transform (:52-77), and the reference's compiled Cython bbox_overlaps_cython.
np.random is the real numpy generator, seeded per case; the fixture also stores the generator state
after the call, so a device replay must consume exactly the same number of MT19937 outputs.
-> tests/golden/rpn_target.npz (inputs are regenerated from seeds by tests/rpncases.py)
"""
import ast
import copy
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SIMPLEDET_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from tests import rpncases  # noqa: E402


def load_defs(path, names, env):
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert len(body) == len(names), (path, names, [n.name for n in body])
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), env)
    return env


def reference_classes():
    from oracle._ref import bbox as ref_bbox
    env = {"np": np, "copy": copy, "bbox_overlaps_cython": ref_bbox.bbox_overlaps_cython}
    load_defs(os.path.join(REF, "operator_py", "bbox_transform.py"), ["nonlinear_transform"], env)
    env["bbox_transform"] = env["nonlinear_transform"]  # core/detection_input.py:11
    load_defs(os.path.join(REF, "core", "detection_input.py"), ["DetectionAugmentation", "AnchorTarget2D"], env)
    load_defs(os.path.join(REF, "models", "FPN", "input.py"),
              ["PyramidAnchorTarget2DBase", "PyramidAnchorTarget2D"], env)
    return env


def make_param(cfg):
    """the AnchorTarget2DParam class of config/faster_r50v1_fpn_1x.py:212-232, parameterised"""
    gen = types.SimpleNamespace(stride=cfg["stride"], short=cfg["short"], long=cfg["long"],
                                scales=cfg["scales"], aspects=cfg["aspects"])
    assign = types.SimpleNamespace(allowed_border=cfg["allowed_border"], pos_thr=cfg["pos_thr"],
                                   neg_thr=cfg["neg_thr"], min_pos_thr=cfg["min_pos_thr"])
    sample = types.SimpleNamespace(image_anchor=cfg["image_anchor"], pos_fraction=cfg["pos_fraction"])
    return types.SimpleNamespace(generate=gen, assign=assign, sample=sample)


def main():
    env = reference_classes()
    out = {}
    for name, case in sorted(rpncases.CASES.items()):
        cfg = case["cfg"]
        p = make_param(cfg)
        pyramid = isinstance(cfg["stride"], (tuple, list))
        op = env["PyramidAnchorTarget2D"](p) if pyramid else env["AnchorTarget2D"](p)
        np.random.seed(case["seed"])
        res = []
        for im_info, gt in rpncases.inputs(case):
            rec = {"im_info": im_info, "gt_bbox": gt.copy()}
            res.append(op.apply(rec))
        state = np.random.get_state()
        for i, (lab, tgt, wgt) in enumerate(res):
            # compact storage: labels are -1/0/1 (int8); targets / weights are zero except at the
            # (<= image_anchor) foreground anchors: flat indices of weight != 0 + the values there
            lab, tgt, wgt = np.asarray(lab, np.float32), np.asarray(tgt, np.float32), np.asarray(wgt, np.float32)
            assert set(np.unique(lab)) <= {-1.0, 0.0, 1.0} and set(np.unique(wgt)) <= {0.0, 1.0}
            nz = np.flatnonzero(wgt.reshape(-1))
            assert not np.any(tgt.reshape(-1)[np.setdiff1d(np.arange(tgt.size), nz)])
            out["%s/%d/label" % (name, i)] = lab.astype(np.int8)
            out["%s/%d/shape" % (name, i)] = np.array(tgt.shape, np.int64)
            out["%s/%d/nz" % (name, i)] = nz.astype(np.int64)
            out["%s/%d/target_nz" % (name, i)] = tgt.reshape(-1)[nz]
        out["%s/mt_key" % name] = state[1].astype(np.uint32)
        out["%s/mt_pos" % name] = np.array([state[2]], np.int32)
        import hashlib
        for orient in ("h", "v"):
            a = np.ascontiguousarray(getattr(op, orient + "_all_anchor"), np.float64)
            out["%s/%s_all_anchor_sha256" % (name, orient)] = np.frombuffer(
                hashlib.sha256(a.tobytes()).digest(), np.uint8)
        nfg = [int((r[0] == 1).sum()) for r in res]
        nbg = [int((r[0] == 0).sum()) for r in res]
        print("%-24s images %d  fg %s  bg %s  label %s target %s" % (name, len(res), nfg, nbg, res[0][0].shape,
                                                                   res[0][1].shape))
    np.savez_compressed(os.path.join(HERE, "rpn_target.npz"), **out)


if __name__ == "__main__":
    main()
