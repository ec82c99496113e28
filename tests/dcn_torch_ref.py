"""TEST INFRASTRUCTURE -- an independent second statement of DeformableConvolution v1, in fp64 PyTorch,
whose three gradients come from AUTOGRAD instead of from a restatement of MXNet's hand-written
backward kernels.

Why it exists ("Missing 1"): the arithmetic the reference runs (`models/dcn/builder.py:14-17`
-> upstream MXNet 1.6 `src/operator/contrib/nn/deformable_im2col.cuh`) is not vendored, so
`oracle/deform_conv.c` cannot be pinned to a compiled twin.  This file shares NO code with the C
oracle: the forward is written from the paper's definition in *absolute* image coordinates with
explicit index gathers (the C oracle follows MXNet's patch-relative pointer arithmetic), and the
backward is whatever `torch.autograd` derives from it.  `tests/test_deform_conv_autograd.py` compares

    oracle.deform_im2col / deform_conv_fwd          with  dcn_forward() / dcn_col()
    oracle.deform_col2im       (MXNet's dX kernel)   with  d loss / d x
    oracle.deform_col2im_coord (MXNet's dOffset)     with  d loss / d offset
    dY . col^T                                       with  d loss / d weight

on full tensors, and enumerates the places where MXNet's backward is *not* literally the derivative
of its forward.

Forward semantics restated (the only facts taken from the published kernel):
  * sample position of (pixel p, tap (i, j), deformable group g):
        h = h_out * stride - pad + i * dil + offset[g, 2 * (i * kw + j)    ][p]
        w = w_out * stride - pad + j * dil + offset[g, 2 * (i * kw + j) + 1][p]
  * the sample is 0 unless 0 <= h < H and 0 <= w < W
  * bilinear interpolation between floor(h) and floor(h) + 1, except that from floor(h) >= H - 1 on
    both rows are H - 1 and the row fraction is 0 (the value of the last row, no fade-out); same for w
  * y[n, f, p] = sum_{c, i, j} weight[f, c, i, j] * sample(x[n, c], p, i, j, group of c)
"""
import torch


def _coords(offset, H, W, kh, kw, pad, stride, dil, dgroup):
    """sample coordinates (N, dgroup, kh*kw, Ho, Wo) in fp64, from offsets in their stored precision"""
    N, _, Ho, Wo = offset.shape
    off = offset.double().reshape(N, dgroup, kh * kw, 2, Ho, Wo)
    dev = offset.device
    hs = (torch.arange(Ho, device=dev, dtype=torch.float64) * stride - pad).reshape(1, 1, 1, Ho, 1)
    ws = (torch.arange(Wo, device=dev, dtype=torch.float64) * stride - pad).reshape(1, 1, 1, 1, Wo)
    ti = (torch.arange(kh * kw, device=dev) // kw).double().reshape(1, 1, kh * kw, 1, 1) * dil
    tj = (torch.arange(kh * kw, device=dev) % kw).double().reshape(1, 1, kh * kw, 1, 1) * dil
    return hs + ti + off[:, :, :, 0], ws + tj + off[:, :, :, 1]


def dcn_col(x, offset, kernel=(3, 3), pad=1, stride=1, dil=1, dgroup=1):
    """x (N,C,H,W), offset (N, dgroup*2*kh*kw, Ho, Wo) -> col (N, C, kh*kw, Ho, Wo), differentiable in
    x and offset (floor() has zero gradient, so d/d offset is the one-sided bilinear slope)."""
    kh, kw = kernel
    N, C, H, W = x.shape
    Ho, Wo = offset.shape[2:]
    cpg = C // dgroup
    h, w = _coords(offset, H, W, kh, kw, pad, stride, dil, dgroup)
    inside = (h >= 0) & (w >= 0) & (h < H) & (w < W)

    def axis(v, size):
        lo = torch.floor(v.detach())
        clamp = lo >= size - 1
        lo_i = torch.where(clamp, torch.full_like(lo, size - 1), lo).clamp(0, size - 1).long()
        hi_i = torch.where(clamp, lo_i, (lo_i + 1).clamp(max=size - 1))
        frac = torch.where(clamp, torch.zeros_like(v), v - lo)
        return lo_i, hi_i, frac

    hl, hh, lh = axis(h, H)
    wl, wh, lw = axis(w, W)
    xg = x.double().reshape(N, dgroup, cpg, H * W)

    def take(hi, wi):
        idx = (hi * W + wi).reshape(N, dgroup, 1, -1).expand(N, dgroup, cpg, -1)
        return torch.gather(xg, 3, idx).reshape(N, dgroup, cpg, kh * kw, Ho, Wo)

    v1, v2, v3, v4 = take(hl, wl), take(hl, wh), take(hh, wl), take(hh, wh)
    lh, lw = lh.unsqueeze(2), lw.unsqueeze(2)
    val = (1 - lh) * (1 - lw) * v1 + (1 - lh) * lw * v2 + lh * (1 - lw) * v3 + lh * lw * v4
    val = torch.where(inside.unsqueeze(2), val, torch.zeros_like(val))
    return val.reshape(N, C, kh * kw, Ho, Wo)


def dcn_forward(x, offset, weight, pad=1, stride=1, dil=1, dgroup=1, bias=None, num_group=1):
    """-> y (N, F, Ho, Wo) in fp64.  weight (F, C / num_group, kh, kw): filter block g sees the channel
    block g (the definition of a grouped convolution); bias (F) added per filter."""
    F, Cg, kh, kw = weight.shape
    N, C = x.shape[:2]
    col = dcn_col(x, offset, (kh, kw), pad, stride, dil, dgroup)
    Ho, Wo = col.shape[-2:]
    colg = col.reshape(N, num_group, Cg, kh * kw, Ho, Wo)
    wg = weight.double().reshape(num_group, F // num_group, Cg, kh * kw)
    y = torch.einsum("gfck,ngckhw->ngfhw", wg, colg).reshape(N, F, Ho, Wo)
    if bias is not None:
        y = y + bias.double().reshape(1, F, 1, 1)
    return y


def dcn_grads(x, offset, weight, dy, pad=1, stride=1, dil=1, dgroup=1, bias=None, num_group=1):
    """autograd gradients of <dcn_forward(x, offset, weight[, bias]), dy> -> (y, dx, doffset, dweight[, dbias]),
    fp64"""
    xr = x.double().clone().requires_grad_(True)
    orr = offset.double().clone().requires_grad_(True)
    wr = weight.double().clone().requires_grad_(True)
    br = bias.double().clone().requires_grad_(True) if bias is not None else None
    y = dcn_forward(xr, orr, wr, pad, stride, dil, dgroup, br, num_group)
    (y * dy.double()).sum().backward()
    out = (y.detach(), xr.grad, orr.grad, wr.grad)
    return out + (br.grad,) if bias is not None else out


def keep_off_the_kinks(offset, H, W, kernel=(3, 3), pad=1, stride=1, dil=1, dgroup=1, margin=2e-3):
    """Nudge fp32 offsets so that no sample coordinate lies within `margin` of an integer (where the
    bilinear surface has a kink and where fp32-vs-fp64 rounding of the coordinate could flip a
    floor() or the inside test).  Returns a new fp32 tensor."""
    kh, kw = kernel
    off = offset.clone().float()
    for _ in range(4):
        h, w = _coords(off, H, W, kh, kw, pad, stride, dil, dgroup)
        N, _, Ho, Wo = off.shape
        o = off.reshape(N, dgroup, kh * kw, 2, Ho, Wo)
        for k, v in ((0, h), (1, w)):
            near = (v - torch.round(v)).abs() < margin
            o[:, :, :, k][near] += 4 * margin
        off = o.reshape(off.shape)
    return off
