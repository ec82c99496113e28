"""Seeded cases for the RPN anchor-target assignment (core/detection_input.py:345-565,
models/FPN/input.py:9-146).  tests/golden/make_golden_rpn.py runs them through the reference's own
classes; the numpy oracle (oracle/rpn_target.py) and the HIP op are compared with those fixtures."""
import numpy as np

from simpledet_amd import synth

FPN = dict(stride=(4, 8, 16, 32, 64), short=(200, 100, 50, 25, 13), long=(334, 167, 84, 42, 21),
           scales=(8,), aspects=(0.5, 1.0, 2.0), allowed_border=0, pos_thr=0.7, neg_thr=0.3,
           min_pos_thr=0.0, image_anchor=256, pos_fraction=0.5)      # config/faster_r50v1_fpn_1x.py:212-232
C4 = dict(stride=16, short=50, long=84, scales=(2, 4, 8, 16, 32), aspects=(0.5, 1.0, 2.0),
          allowed_border=0, pos_thr=0.7, neg_thr=0.3, min_pos_thr=0.0, image_anchor=256,
          pos_fraction=0.5)                                          # config/faster_r50v1c4_c5_512roi_1x.py

CASES = {
    # two landscape images through one generator state (the state carries from image to image)
    "fpn_landscape": dict(cfg=FPN, seed=0, images=[(800, 1333, 11), (768, 1280, 12)]),
    # portrait (h >= w selects v_all_anchor) and an image without gt boxes
    "fpn_portrait_nogt": dict(cfg=FPN, seed=1, images=[(1333, 800, 13), (800, 1200, None)]),
    "c4": dict(cfg=C4, seed=2, images=[(800, 1333, 14), (600, 1000, 15)]),
    # min_pos_thr > 0 and a border allowance; fewer anchors per image
    "fpn_thr": dict(cfg=dict(FPN, min_pos_thr=0.3, allowed_border=8, image_anchor=128, pos_fraction=0.25),
                    seed=3, images=[(800, 1333, 16)]),
    # a gt box no anchor overlaps (far outside the valid anchors): with min_pos_thr = 0 the reference's
    # `overlaps == gt_max_overlaps` then marks EVERY valid anchor positive (its own TODO, :466-472)
    "fpn_zero_overlap_gt": dict(cfg=FPN, seed=4, images=[(800, 1333, "far")]),
}


def inputs(case):
    """[(im_info float32 (3,), gt_bbox float32 (100, 5))] for the images of a case"""
    out = []
    for h, w, gseed in case["images"]:
        im_info = np.array([h, w, 1.0], np.float32)
        gt = -np.ones((100, 5), np.float32)
        if gseed == "far":
            g = synth.gt_boxes(99, 1, 100, img_h=h, img_w=w, min_n=3, max_n=6)[0]
            n = int((g[:, 4] != -1).sum())
            g[n] = [5000, 5000, 5100, 5100, 7]
            gt = g
        elif gseed is not None:
            gt = synth.gt_boxes(gseed, 1, 100, img_h=h, img_w=w, min_n=2, max_n=30)[0]
        out.append((im_info, gt))
    return out
