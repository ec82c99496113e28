"""Seeded synthetic workloads for tests and bench.py; numpy only.

Shapes come from config/faster_r50v1_fpn_1x.py of the reference: 800x1333 input, FPN strides
(4,8,16,32), C=256, 512 sampled RoIs per image, 7x7 RoIAlign, 2000 proposals, <=100 gt boxes.
"""
import numpy as np

IMG_H, IMG_W = 800, 1333
FPN_STRIDES = (4, 8, 16, 32)
FPN_SHAPES = ((200, 334), (100, 167), (50, 84), (25, 42))


def fpn_shapes(img_h=IMG_H, img_w=IMG_W, strides=FPN_STRIDES):
    return tuple((int(np.ceil(img_h / s)), int(np.ceil(img_w / s))) for s in strides)


def feature_maps(seed, batch=2, channels=256, shapes=FPN_SHAPES, relu=False):
    rs = np.random.RandomState(seed)
    feats = [rs.standard_normal((batch, channels, h, w)).astype(np.float32) for h, w in shapes]
    if relu:
        feats = [np.maximum(f, 0) for f in feats]
    return feats


def degenerate_rois(img_h=IMG_H, img_w=IMG_W):
    """Boxes that exercise is_empty, alpha=beta=0.5, clamping."""
    return np.array([
        [0, 0, 0, 0],                        # zero box (padding row)
        [100, 100, 100, 100],                # zero-size box inside the image
        [100, 100, 101, 101],                # 1-px box
        [-300, -200, -10, -10],              # completely outside (negative side)
        [img_w + 10, img_h + 10, img_w + 200, img_h + 300],  # completely outside (far side)
        [64, 32, 64 + 112, 32 + 112],        # integer-aligned on every stride
        [0, 0, img_w - 1, img_h - 1],        # whole image
        [-50, -50, img_w + 50, img_h + 50],  # larger than the image
        [200, 300, 100, 200],                # inverted box (x2<x1, y2<y1)
        [10.5, 20.25, 10.5 + 1e-3, 400.75],  # (almost) zero width
        [500, 790, 900, 799.5],              # thin strip at the bottom border
        [1320, 10, 1332, 700],               # thin strip at the right border
    ], dtype=np.float32)


def random_rois(seed, batch=2, num=512, img_h=IMG_H, img_w=IMG_W, degenerate=True,
                min_size=16.0, max_size=800.0):
    rs = np.random.RandomState(seed)
    out = np.empty((batch, num, 4), np.float32)
    for b in range(batch):
        cx = rs.uniform(0, img_w, num)
        cy = rs.uniform(0, img_h, num)
        s = np.exp(rs.uniform(np.log(min_size), np.log(max_size), num))
        ar = np.exp(rs.uniform(np.log(1 / 3.0), np.log(3.0), num))
        w = s * np.sqrt(ar)
        h = s / np.sqrt(ar)
        box = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
        box[:, 0::2] = np.clip(box[:, 0::2], 0, img_w - 1)
        box[:, 1::2] = np.clip(box[:, 1::2], 0, img_h - 1)
        out[b] = box.astype(np.float32)
        if degenerate:
            d = degenerate_rois(img_h, img_w)
            k = min(len(d), num)
            pos = rs.choice(num, k, replace=False)
            out[b, pos] = d[:k]
    return out


def level_balanced_rois(seed, batch=2, per_level=128, strides=FPN_STRIDES, img_h=IMG_H,
                        img_w=IMG_W):
    """per_level boxes for each FPN level by construction (sqrt(area) inside the level's range)."""
    rs = np.random.RandomState(seed)
    num = per_level * len(strides)
    out = np.empty((batch, num, 4), np.float32)
    for b in range(batch):
        boxes = []
        for li in range(len(strides)):
            lo = 224.0 * 2.0 ** (li - 2) * 1.05 if li > 0 else 16.0
            hi = 224.0 * 2.0 ** (li - 1) * 0.95 if li < len(strides) - 1 else 700.0
            s = np.exp(rs.uniform(np.log(lo), np.log(hi), per_level))
            ar = np.exp(rs.uniform(np.log(0.5), np.log(2.0), per_level))
            w, h = s * np.sqrt(ar), s / np.sqrt(ar)
            w = np.minimum(w, img_w - 2)
            h = np.minimum(h, img_h - 2)
            x1 = rs.uniform(0, img_w - 1 - w)
            y1 = rs.uniform(0, img_h - 1 - h)
            boxes.append(np.stack([x1, y1, x1 + w - 1, y1 + h - 1], 1))
        boxes = np.concatenate(boxes, 0)
        out[b] = boxes[rs.permutation(num)].astype(np.float32)
    return out


def gt_boxes(seed, batch=2, max_gt=100, num_classes=81, img_h=IMG_H, img_w=IMG_W, min_n=1,
             max_n=40):
    """(B, max_gt, 5) [x1,y1,x2,y2,cls], padded with -1 (core/detection_input.py:310-311)."""
    rs = np.random.RandomState(seed)
    out = -np.ones((batch, max_gt, 5), np.float32)
    for b in range(batch):
        n = rs.randint(min(min_n, max_gt), min(max_n, max_gt) + 1)
        w = np.exp(rs.uniform(np.log(16), np.log(600), n))
        h = np.exp(rs.uniform(np.log(16), np.log(500), n))
        x1 = rs.uniform(0, np.maximum(img_w - w, 1))
        y1 = rs.uniform(0, np.maximum(img_h - h, 1))
        x2 = np.minimum(x1 + w, img_w - 1)
        y2 = np.minimum(y1 + h, img_h - 1)
        out[b, :n, :4] = np.round(np.stack([x1, y1, x2, y2], 1), 1)
        out[b, :n, 4] = rs.randint(1, num_classes, n)
    return out


def proposals(seed, gt, num=2000, img_h=IMG_H, img_w=IMG_W, frac_near_gt=0.15, pad_rows=37):
    """(B, num, 4) proposals: jittered gt + random boxes, trailing all-zero padding rows."""
    rs = np.random.RandomState(seed)
    B = gt.shape[0]
    out = np.zeros((B, num, 4), np.float32)
    for b in range(B):
        g = gt[b][gt[b, :, 4] != -1][:, :4]
        n_real = num - pad_rows
        n_near = int(n_real * frac_near_gt)
        idx = rs.randint(0, len(g), n_near)
        base = g[idx]
        w = base[:, 2] - base[:, 0] + 1
        h = base[:, 3] - base[:, 1] + 1
        jit = rs.uniform(-0.3, 0.3, (n_near, 4)) * np.stack([w, h, w, h], 1)
        near = base + jit
        rnd = random_rois(rs.randint(1 << 30), 1, n_real - n_near, img_h, img_w,
                          degenerate=False)[0]
        boxes = np.concatenate([near, rnd], 0)
        boxes[:, 0::2] = np.clip(boxes[:, 0::2], 0, img_w - 1)
        boxes[:, 1::2] = np.clip(boxes[:, 1::2], 0, img_h - 1)
        x1 = np.minimum(boxes[:, 0], boxes[:, 2]); x2 = np.maximum(boxes[:, 0], boxes[:, 2])
        y1 = np.minimum(boxes[:, 1], boxes[:, 3]); y2 = np.maximum(boxes[:, 1], boxes[:, 3])
        boxes = np.stack([x1, y1, x2, np.maximum(y2, 1.0)], 1)
        out[b, :n_real] = boxes[rs.permutation(n_real)].astype(np.float32)
    return out


def nms_dets(seed, num=1000, img_h=IMG_H, img_w=IMG_W, mode="clustered"):
    """(num,5) [x1,y1,x2,y2,score]: clusters of 5-20 jittered boxes."""
    rs = np.random.RandomState(seed)
    if mode == "no_overlap":
        g = int(np.ceil(np.sqrt(num)))
        ix, iy = np.meshgrid(np.arange(g), np.arange(g))
        x1 = (ix.reshape(-1)[:num] * 40).astype(np.float32)
        y1 = (iy.reshape(-1)[:num] * 40).astype(np.float32)
        boxes = np.stack([x1, y1, x1 + 30, y1 + 30], 1)
    elif mode == "all_overlap":
        boxes = np.tile(np.array([[100, 100, 300, 300]], np.float32), (num, 1))
        boxes += rs.uniform(-2, 2, boxes.shape).astype(np.float32)
    else:
        boxes = []
        while sum(len(b) for b in boxes) < num:
            k = rs.randint(5, 21)
            w = np.exp(rs.uniform(np.log(20), np.log(400)))
            h = np.exp(rs.uniform(np.log(20), np.log(400)))
            cx, cy = rs.uniform(0, img_w), rs.uniform(0, img_h)
            c = np.array([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2])
            j = rs.uniform(-0.15, 0.15, (k, 4)) * np.array([w, h, w, h])
            boxes.append(c + j)
        boxes = np.concatenate(boxes, 0)[:num]
        boxes[:, 0::2] = np.clip(boxes[:, 0::2], 0, img_w - 1)
        boxes[:, 1::2] = np.clip(boxes[:, 1::2], 0, img_h - 1)
    scores = rs.uniform(0.05, 1.0, (num, 1))
    return np.concatenate([boxes, scores], 1).astype(np.float32)


def proposal_target_inputs(seed, batch=2, num=2000, max_gt=100, n_gt=None):
    """(rois (B,num,4), gt (B,max_gt,5)) for ProposalTarget; n_gt fixes the gt count per image."""
    if n_gt is None:
        gt = gt_boxes(seed, batch, max_gt)
    else:
        gt = np.concatenate([gt_boxes(seed + 101 * b, 1, max_gt, min_n=n, max_n=n)
                             for b, n in enumerate(n_gt)], 0)
    return proposals(seed + 1, gt, num, pad_rows=max(1, num // 50)), gt


def rpn_outputs(seed, batch=2, num_anchors=3, height=50, width=84, stride=16, img_h=IMG_H,
                img_w=IMG_W):
    """(cls_prob (B,2A,H,W) softmax pairs, bbox_pred (B,4A,H,W), im_info (B,3)) for Proposal_v3."""
    rs = np.random.RandomState(seed)
    logit = rs.standard_normal((batch, 2, num_anchors, height, width)).astype(np.float32) * 2
    e = np.exp(logit - logit.max(1, keepdims=True))
    prob = (e / e.sum(1, keepdims=True)).astype(np.float32)
    cls_prob = prob.reshape(batch, 2 * num_anchors, height, width)
    bbox_pred = (rs.standard_normal((batch, 4 * num_anchors, height, width)) * 0.3).astype(np.float32)
    bbox_pred[:, 2::4][rs.rand(batch, num_anchors, height, width) < 0.01] = 6.0  # dw above the clip
    im_info = np.array([[img_h, img_w, 1.0]] * batch, np.float32)
    if batch > 1:
        im_info[1] = [img_h - 64, img_w - 100, 1.5]
    return cls_prob, bbox_pred, im_info


def gt_polys(seed, gt, max_len=400):
    """gt_poly rows for ProposalMaskTarget (models/maskrcnn/input.py:100-127 layout): per gt box
    [category, n_seg, len_1..len_n, x0, y0, x1, y1, ...] padded with -1.  Every valid gt box gets 1-3
    closed polygons (jittered ellipses with 5-24 vertices, some concave) inside its box."""
    rs = np.random.RandomState(seed)
    B, M = gt.shape[:2]
    out = np.full((B, M, max_len), -1.0, np.float32)
    for b in range(B):
        for m in range(M):
            x1, y1, x2, y2, cls = gt[b, m]
            if cls == -1:
                continue
            nseg = int(rs.choice([1, 1, 1, 2, 3]))
            segs = []
            for _ in range(nseg):
                k = int(rs.randint(5, 25))
                cx = rs.uniform(x1 + 0.3 * (x2 - x1), x2 - 0.3 * (x2 - x1))
                cy = rs.uniform(y1 + 0.3 * (y2 - y1), y2 - 0.3 * (y2 - y1))
                rx = rs.uniform(0.15, 0.5) * (x2 - x1)
                ry = rs.uniform(0.15, 0.5) * (y2 - y1)
                ang = np.sort(rs.uniform(0, 2 * np.pi, k))
                rad = rs.uniform(0.5, 1.0, k)
                px = np.clip(cx + rx * rad * np.cos(ang), x1, x2)
                py = np.clip(cy + ry * rad * np.sin(ang), y1, y2)
                segs.append(np.stack([px, py], 1).reshape(-1))
            row = [float(cls), float(nseg)] + [float(len(sg)) for sg in segs]
            for sg in segs:
                row += [float(v) for v in sg]
            assert len(row) <= max_len
            out[b, m, :len(row)] = np.asarray(row, np.float32)
    return out
