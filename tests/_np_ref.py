"""Independent numpy restatements used to cross-check oracle/nesie_oracle.c.

Written from SURVEY.md appendix A (the order-independent formulations), NOT by
transliterating the C oracle: FPS here reduces the 64-bit tie-break key with a
plain max, the C oracle simulates the reference's thread-striped scan and LDS
tree literally -- agreement of the two is the evidence for the tie rule.
"""
import math

import numpy as np

f32 = np.float32


# Form of the squared distance (the switch of oracle/nesie_oracle.c and include/nesie_ops.h):
# 0 = no contraction, 1 = fma(dz, dz, fma(dx, dx, dy*dy)), 2 = fma(dz, dz, fma(dy, dy, dx*dx))
FORM = 0


def fma32(a, b, c):
    """Correctly rounded float32 fused multiply-add, element-wise, without an fma instruction: the
    product of two float32 is exact in float64 (48 bits); the sum with c is formed in float64 with
    its exact error (TwoSum) and rounded TO ODD, which makes the final rounding to float32 immune to
    double rounding (53 >= 24 + 2 bits)."""
    a, b, c = (np.asarray(v, dtype=f32).astype(np.float64) for v in (a, b, c))
    p = a * b
    s = p + c
    bp = s - p
    e = (p - (s - bp)) + (c - bp)
    odd = (s.view(np.int64) & 1) == 1
    toward = np.where(e > 0, np.inf, -np.inf)
    s = np.where((e != 0) & ~odd, np.nextafter(s, toward), s)
    return s.astype(f32)


def sqdist(a, b):
    """The squared distance in float32 in the selected FORM, operand order a - b."""
    d = (a.astype(f32) - b.astype(f32)).astype(f32)
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
    if FORM == 1:
        return fma32(dz, dz, fma32(dx, dx, (dy * dy).astype(f32)))
    if FORM == 2:
        return fma32(dz, dz, fma32(dy, dy, (dx * dx).astype(f32)))
    xx = (dx * dx).astype(f32)
    yy = (dy * dy).astype(f32)
    zz = (dz * dz).astype(f32)
    return ((xx + yy).astype(f32) + zz).astype(f32)


def ref_block_size(n):
    pow_2 = int(math.log(float(n)) / math.log(2.0))
    return max(min(1 << pow_2, 1024), 1)


def bitrev(t, bits):
    r = 0
    for i in range(bits):
        r |= ((t >> i) & 1) << (bits - 1 - i)
    return r


def fps_key(xyz, m):
    """(N,3) -> (m,) indices; winner = max d2, ties by min (bitrev(k mod bs), k)."""
    n = xyz.shape[0]
    bs = ref_block_size(n)
    L = bs.bit_length() - 1
    k = np.arange(n)
    rb = np.array([bitrev(int(t), L) for t in (k % bs)], dtype=np.int64)
    tie = rb * (1 << 32) + k  # smaller is better
    temp = np.full(n, 1e10, dtype=f32)
    out = np.zeros(m, dtype=np.int32)
    old = 0
    for j in range(1, m):
        d = sqdist(xyz, xyz[old][None, :])
        temp = np.minimum(d, temp).astype(f32)
        best = temp.max()
        cand = np.nonzero(temp == best)[0]
        old = int(cand[np.argmin(tie[cand])])
        out[j] = old
    return out, temp


def ball_query(new_xyz, xyz, min_r, max_r, ns):
    """(M,3),(N,3) -> (M,ns) int32."""
    M = new_xyz.shape[0]
    max_r2 = f32(f32(max_r) * f32(max_r))
    min_r2 = f32(f32(min_r) * f32(min_r))
    out = np.zeros((M, ns), dtype=np.int32)
    for i in range(M):
        d2 = sqdist(new_xyz[i][None, :], xyz)
        hit = np.nonzero((d2 == 0) | ((d2 >= min_r2) & (d2 < max_r2)))[0]
        if hit.size:
            out[i, :] = hit[0]
            h = hit[:ns]
            out[i, :h.size] = h
    return out


def three_nn(unknown, known):
    """(n,3),(m,3) -> dist2 (n,3) f32, idx (n,3) i32; stable order = first seen wins."""
    n, m = unknown.shape[0], known.shape[0]
    d2 = np.full((n, 3), np.inf, dtype=f32)
    ix = np.zeros((n, 3), dtype=np.int32)
    if m == 0:
        return d2, ix
    for i in range(n):
        d = sqdist(unknown[i][None, :], known)
        order = np.argsort(d, kind="stable")[:3]
        d2[i, :order.size] = d[order]
        ix[i, :order.size] = order
    return d2, ix


def points_in_boxes(boxes, pts):
    """LiDAR-frame boxes (T,7), pts (M,3) -> (M,T) int32, mixed f32/f64 as the reference."""
    T, M = boxes.shape[0], pts.shape[0]
    out = np.zeros((M, T), dtype=np.int32)
    for k in range(T):
        cx, cy, cz, w, l, h, rz = [f32(v) for v in boxes[k]]
        czm = f32(np.float64(cz) + np.float64(h) / 2.0)
        rot = f32(np.float64(rz) + math.pi / 2)
        cosa = f32(math.cos(float(rot)))
        sina = f32(math.sin(float(rot)))
        x, y, z = pts[:, 0].astype(f32), pts[:, 1].astype(f32), pts[:, 2].astype(f32)
        zin = ~(np.abs((z - czm).astype(f32)).astype(np.float64) > np.float64(h) / 2.0)
        sx = (x - cx).astype(f32)
        sy = (y - cy).astype(f32)
        lx = ((sx * cosa).astype(f32) + (sy * f32(-sina)).astype(f32)).astype(f32)
        ly = ((sx * sina).astype(f32) + (sy * cosa).astype(f32)).astype(f32)
        hl, hw = np.float64(l) / 2.0, np.float64(w) / 2.0
        inside = zin & (lx.astype(np.float64) > -hl) & (lx.astype(np.float64) < hl) & \
            (ly.astype(np.float64) > -hw) & (ly.astype(np.float64) < hw)
        out[:, k] = inside.astype(np.int32)
    return out


def _sv_compare(x1, y1, x2, y2):
    E = 1e-8
    x1, y1, x2, y2 = f32(x1), f32(y1), f32(x2), f32(y2)
    if abs(f32(x1 - x2)) < E and abs(f32(y2 - y1)) < E:
        return False
    if y1 > 0 and y2 < 0:
        return True
    if y1 < 0 and y2 > 0:
        return False
    n1 = f32(np.float64(f32(f32(x1 * x1) + f32(y1 * y1))) + E)
    n2 = f32(np.float64(f32(f32(x2 * x2) + f32(y2 * y2))) + E)
    a = f32(f32(abs(x1) * x1) / n1)
    c = f32(f32(abs(x2) * x2) / n2)
    diff = np.float64(f32(a - c))
    if y1 > 0 and y2 > 0:
        return bool(diff > E)
    if y1 < 0 and y2 < 0:
        return bool(diff < E)
    return False


def sort_vertices(vertices, mask, num_valid):
    """(P,24,2),(P,24),(P,) -> (P,9) int32, following sort_vert_kernel.cu:42-134."""
    P, m = mask.shape
    out = np.zeros((P, 9), dtype=np.int32)
    for i in range(P):
        pad = m - 1
        for j in range(8, m):
            if not mask[i, j]:
                pad = j
                break
        nv = int(num_valid[i])
        if nv < 3:
            out[i, :] = pad
            continue
        o = [0] * 9
        for j in range(nv):
            x_min, y_min, take = f32(1), f32(-1e-8), 0
            for k in range(m):
                x, y = vertices[i, k]
                if not mask[i, k]:
                    continue
                if j == 0:
                    ok = _sv_compare(x, y, x_min, y_min)
                else:
                    x2, y2 = vertices[i, o[j - 1]]
                    ok = _sv_compare(x, y, x_min, y_min) and _sv_compare(x2, y2, x, y)
                if ok:
                    x_min, y_min, take = x, y, k
            o[j] = take
        o[nv] = o[0]
        for j in range(nv + 1, 9):
            o[j] = pad
        if nv == 8:
            counter = sum(1 for j in range(4) for k in range(4, 8) if o[k] == o[j])
            if counter == 4:
                o[4] = o[0]
                for j in range(5, 9):
                    o[j] = pad
        out[i] = o
    return out
