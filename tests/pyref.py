"""Independent pure-Python (numpy float32 scalar) restatements for SMALL cases.

A second implementation of the oracle's algorithms in a different language, used to cross-check
oracle/*.c typing decisions (float vs double sub-expressions).  Slow by design.
"""
import math

import numpy as np

f32 = np.float32


def _fmax(a, b):
    return a if a > b else b


def _fmin(a, b):
    return a if a < b else b


def roi_align_v2_fwd(data, rois, pooled, spatial_scale):
    """operator_cxx/contrib/roi_align_v2-inl.h:61-153"""
    B, C, H, W = data.shape
    R = rois.shape[1]
    PH, PW = pooled
    out = np.zeros((B, R, C, PH, PW), f32)
    ax = np.zeros_like(out)
    ay = np.zeros_like(out)
    ss = f32(spatial_scale)
    for n in range(B * R):
        b = n // R
        r = rois.reshape(-1, 4)[n].astype(f32)
        rsw, rsh, rew, reh = r[0] * ss, r[1] * ss, r[2] * ss, r[3] * ss
        rw, rh = f32(rew - rsw), f32(reh - rsh)
        bh, bw = f32(rh / f32(PH)), f32(rw / f32(PW))
        for ph in range(PH):
            for pw in range(PW):
                hs = _fmin(_fmax(f32(f32(ph) * bh) + rsh, f32(0)), f32(H - 1))
                he = _fmin(_fmax(f32(f32(ph + 1) * bh) + rsh, f32(0)), f32(H - 1))
                ws = _fmin(_fmax(f32(f32(pw) * bw) + rsw, f32(0)), f32(W - 1))
                we = _fmin(_fmax(f32(f32(pw + 1) * bw) + rsw, f32(0)), f32(W - 1))
                empty = (he <= hs) or (we <= ws)
                for c in range(C):
                    mv, mx, my = f32(0), f32(-1), f32(-1)
                    if not empty:
                        mv = f32(-3.4028234663852886e38)
                        hst = f32(np.float64(f32(he - hs)) / 3.0)
                        wst = f32(np.float64(f32(we - ws)) / 3.0)
                        h = f32(hs + hst)
                        while np.float64(h) <= np.float64(f32(he - hst)) + 0.01:
                            w = f32(ws + wst)
                            while np.float64(w) <= np.float64(f32(we - wst)) + 0.01:
                                hl = min(max(int(math.floor(h)), 0), H - 1)
                                hh = min(max(int(math.ceil(h)), 0), H - 1)
                                wl = min(max(int(math.floor(w)), 0), W - 1)
                                wr = min(max(int(math.ceil(w)), 0), W - 1)
                                al = f32(0.5) if hl == hh else f32(f32(h - f32(hl)) / f32(hh - hl))
                                be = f32(0.5) if wl == wr else f32(f32(w - f32(wl)) / f32(wr - wl))
                                one = f32(1)
                                p = data[b, c]
                                v = f32(f32(f32(one - al) * f32(one - be)) * p[hl, wl])
                                v = f32(v + f32(f32(al * f32(one - be)) * p[hh, wl]))
                                v = f32(v + f32(f32(f32(one - al) * be) * p[hl, wr]))
                                v = f32(v + f32(f32(al * be) * p[hh, wr]))
                                if v > mv:
                                    mv, mx, my = v, w, h
                                w = f32(w + _fmax(wst, f32(0.01)))
                            h = f32(h + _fmax(hst, f32(0.01)))
                    out[b, n % R, c, ph, pw] = mv
                    ax[b, n % R, c, ph, pw] = mx
                    ay[b, n % R, c, ph, pw] = my
    return out, ax, ay
