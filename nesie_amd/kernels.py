"""Kernel back end used by ``nesie_amd.mmdet3d_ops``.

The product back end is :class:`HipKernels` (libnesie_hip.so on the current HIP
stream).  ``use_backend`` lets TEST CODE and bench.py's cpu_baseline leg inject
another object with the same methods (the CPU oracle lives outside this
package, under ``oracle/``); nothing in nesie_amd ever installs one, and
without an injected back end a non-HIP tensor is an error, not a fallback.

Method names and argument order are those of the reference's extension
modules; every tensor must be contiguous; outputs are
pre-allocated by the caller and written in place.
"""
import contextlib
import os

import ctypes

import torch

from . import _lib

_injected = None


def _ptr(t):
    return t.data_ptr()


def _check(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                "nesie_amd kernels need HIP device tensors (got device "
                f"{t.device}); there is no CPU path in the product")
        if not t.is_contiguous():
            raise RuntimeError("nesie_amd kernels need contiguous tensors")


def _f32(*ts):
    for t in ts:
        if t.dtype != torch.float32:
            raise TypeError(f"expected float32, got {t.dtype}")


def _i32(*ts):
    for t in ts:
        if t.dtype != torch.int32:
            raise TypeError(f"expected int32, got {t.dtype}")


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


class _BNPoolMixin:
    def bn_relu_maxpool_forward(self, x, gamma, beta, running_mean, running_var, momentum, eps,
                                pooled, argmax, save_mean, save_invstd, fwd_coef):
        """x (B, C, M, ns) -> pooled (B, C, M) fp32, argmax (B, C, M) uint8."""
        _check(x, pooled, argmax, save_mean, save_invstd, fwd_coef); _f32(x, pooled)
        b, c, m, ns = x.shape
        assert tuple(pooled.shape) == (b, c, m) and argmax.dtype == torch.uint8
        with torch.cuda.device(x.device):
            ws, need = _bn_pool_ws(x)
            opt = lambda t: 0 if t is None else _ptr(t)  # noqa: E731
            _lib.call("nesie_bn_relu_maxpool_forward", b, c, m, ns, _ptr(x), opt(gamma),
                      opt(beta), opt(running_mean), opt(running_var), float(momentum),
                      float(eps), _ptr(pooled), _ptr(argmax), _ptr(save_mean),
                      _ptr(save_invstd), _ptr(fwd_coef), _ptr(ws), need, _stream(x))

    def bn_relu_maxpool_backward(self, grad_pooled, argmax, x, pooled, gamma, save_invstd,
                                 fwd_coef, dx, dgamma, dbeta):
        _check(grad_pooled, argmax, x, pooled, dx); _f32(grad_pooled, x, pooled, dx)
        b, c, m, ns = x.shape
        assert tuple(grad_pooled.shape) == (b, c, m) and dx.shape == x.shape
        with torch.cuda.device(x.device):
            ws, need = _bn_pool_ws(x)
            opt = lambda t: 0 if t is None else _ptr(t)  # noqa: E731
            _lib.call("nesie_bn_relu_maxpool_backward", b, c, m, ns, _ptr(grad_pooled),
                      _ptr(argmax), _ptr(x), _ptr(pooled), opt(gamma), opt(save_invstd),
                      _ptr(fwd_coef), _ptr(dx), opt(dgamma), opt(dbeta), _ptr(ws), need,
                      _stream(x))


def _bn_pool_ws(x):
    b, c, m, ns = x.shape
    need = _lib.load().nesie_bn_workspace_bytes(b, c, m * ns)
    return torch.empty(max(need, 16), dtype=torch.uint8, device=x.device), need


class HipKernels(_BNPoolMixin):
    """libnesie_hip.so, asynchronous on torch's current HIP stream."""

    name = "hip"

    # The bucket-pruned FPS kernel leaves the scene spatially sorted in its scratch; the ball
    # query that follows on the SAME coordinates (every set-abstraction module samples and then
    # groups one cloud) can use it as an index.  The hand-over is scoped: an entry is recorded
    # only inside ``spatial_index_scope()`` -- one set-abstraction forward -- and dropped when the
    # scope ends, so a ball query somewhere else can never meet the index of an earlier scene
    # (a buffer refilled by a kernel, a memcpy or a graph replay keeps its address AND its
    # version counter).  Inside the scope the entry keeps the coordinate tensor alive and still
    # records address, version, shape, stream and capture state.  One entry per device.
    _spatial_index = {}
    _index_scope_depth = 0

    @staticmethod
    def set_distance_form(form):
        """Process-wide form of the squared distance in FPS / ball query / 3-NN / grid taps
        (include/nesie_ops.h ``nesie_set_distance_form``): 0 = contraction-free (default, the
        tuned kernels), 1 / 2 = the fused forms an nvcc build of the reference may compute."""
        _lib.call("nesie_set_distance_form", int(form))

    @staticmethod
    def get_distance_form():
        return _lib.load().nesie_get_distance_form()

    @staticmethod
    @contextlib.contextmanager
    def cu_budget(n):
        """Inside: the persistent grids of the layer / weight-gradient kernels are sized for ``n``
        CUs instead of the chip's 256 (``nesie_set_cu_count``) -- for launches that run beside
        long-lived workgroups of another stream (the next batch's furthest point sampling holds
        one CU per scene and XCD).  Host-side sizing only: a captured graph keeps what it was
        captured with."""
        lib = _lib.load()
        before = lib.nesie_get_cu_count()
        _lib.call("nesie_set_cu_count", int(n))
        try:
            yield
        finally:
            _lib.call("nesie_set_cu_count", before)

    @classmethod
    def _index_for(cls, xyz, b, n):
        hit = cls._spatial_index.get(xyz.device)
        if hit is None:
            return None
        ref, version, shape, ws, stream, capturing = hit
        # an entry made while a graph was being captured describes that graph's static buffers:
        # replays rewrite them without touching the version counter, so eager code must not
        # trust it (and vice versa)
        same = (ref.data_ptr() == xyz.data_ptr() and ref._version == version
                and xyz._version == version and shape == (b, n) and ref.dtype == xyz.dtype
                and stream == _stream(xyz)
                and capturing == torch.cuda.is_current_stream_capturing())
        return ws if same else None

    def furthest_point_sampling_wrapper(self, b, n, m, xyz, temp, idx):
        _check(xyz, temp, idx); _f32(xyz, temp); _i32(idx)
        assert xyz.numel() == b * n * 3 and temp.numel() == b * n and idx.numel() == b * m
        lib = _lib.load()
        need = lib.nesie_fps_workspace_bytes(b, n)
        with torch.cuda.device(xyz.device):
            if need:
                # scratch for the bucket-pruned kernel, from torch's caching allocator
                ws = torch.empty(need, dtype=torch.uint8, device=xyz.device)
                _lib.call("nesie_furthest_point_sampling_ws", b, n, m, _ptr(xyz),
                          _ptr(temp), _ptr(idx), _ptr(ws), need, _stream(xyz))
                if HipKernels._index_scope_depth > 0 and lib.nesie_fps_leaves_index(b, n):
                    HipKernels._spatial_index[xyz.device] = (
                        xyz, xyz._version, (b, n), ws, _stream(xyz),
                        torch.cuda.is_current_stream_capturing())
            else:
                _lib.call("nesie_furthest_point_sampling_wrapper", b, n, m, _ptr(xyz),
                          _ptr(temp), _ptr(idx), _stream(xyz))

    def furthest_point_sampling_with_dist_wrapper(self, b, n, m, dist, temp, idx):
        _check(dist, temp, idx); _f32(dist, temp); _i32(idx)
        assert dist.numel() == b * n * n and temp.numel() == b * n and idx.numel() == b * m
        with torch.cuda.device(dist.device):
            _lib.call("nesie_furthest_point_sampling_with_dist_wrapper", b, n, m,
                      _ptr(dist), _ptr(temp), _ptr(idx), _stream(dist))

    def ball_query_wrapper(self, b, n, m, min_radius, max_radius, nsample, new_xyz, xyz,
                           idx):
        _check(new_xyz, xyz, idx); _f32(new_xyz, xyz); _i32(idx)
        assert new_xyz.numel() == b * m * 3 and xyz.numel() == b * n * 3
        assert idx.numel() == b * m * nsample
        with torch.cuda.device(xyz.device):
            ws = self._index_for(xyz, b, n) if nsample <= 64 else None
            if ws is not None:  # same results, only the buckets within max_radius are visited
                _lib.call("nesie_ball_query_indexed", b, n, m, float(min_radius),
                          float(max_radius), nsample, _ptr(new_xyz), _ptr(ws), ws.numel(),
                          _ptr(idx), _stream(xyz))
                return
            _lib.call("nesie_ball_query_wrapper", b, n, m, float(min_radius),
                      float(max_radius), nsample, _ptr(new_xyz), _ptr(xyz), _ptr(idx),
                      _stream(xyz))

    def group_points_forward(self, b, c, n, npoints, nsample, points, idx, out):
        _check(points, idx, out); _f32(points, out); _i32(idx)
        assert points.numel() == b * c * n and idx.numel() == b * npoints * nsample
        assert out.numel() == b * c * npoints * nsample
        with torch.cuda.device(points.device):
            _lib.call("nesie_group_points_forward", b, c, n, npoints, nsample,
                      _ptr(points), _ptr(idx), _ptr(out), _stream(points))

    def group_points_backward(self, b, c, n, npoints, nsample, grad_out, idx, grad_points):
        _check(grad_out, idx, grad_points); _f32(grad_out, grad_points); _i32(idx)
        assert grad_out.numel() == b * c * npoints * nsample
        assert idx.numel() == b * npoints * nsample and grad_points.numel() == b * c * n
        with torch.cuda.device(grad_out.device):
            _lib.call("nesie_group_points_backward", b, c, n, npoints, nsample,
                      _ptr(grad_out), _ptr(idx), _ptr(grad_points), _stream(grad_out))

    def gather_points_wrapper(self, b, c, n, npoints, points, idx, out):
        _check(points, idx, out); _f32(points, out); _i32(idx)
        assert points.numel() == b * c * n and idx.numel() == b * npoints
        assert out.numel() == b * c * npoints
        with torch.cuda.device(points.device):
            _lib.call("nesie_gather_points_wrapper", b, c, n, npoints, _ptr(points),
                      _ptr(idx), _ptr(out), _stream(points))

    def gather_points_grad_wrapper(self, b, c, n, npoints, grad_out, idx, grad_points):
        _check(grad_out, idx, grad_points); _f32(grad_out, grad_points); _i32(idx)
        assert grad_out.numel() == b * c * npoints and idx.numel() == b * npoints
        assert grad_points.numel() == b * c * n
        with torch.cuda.device(grad_out.device):
            _lib.call("nesie_gather_points_grad_wrapper", b, c, n, npoints,
                      _ptr(grad_out), _ptr(idx), _ptr(grad_points), _stream(grad_out))

    def query_and_group_forward(self, xyz, centres, features, idx, radius, out):
        """out (B, 3+C, M, ns) = cat[(xyz[idx] - centre) / radius, features[idx]]."""
        _check(xyz, centres, idx, out); _f32(xyz, centres, out); _i32(idx)
        b, n = xyz.shape[:2]
        m, ns = idx.shape[1], idx.shape[2]
        c = 0 if features is None else features.shape[1]
        if features is not None:
            _check(features); _f32(features)
            assert tuple(features.shape) == (b, c, n)
        assert tuple(out.shape) == (b, 3 + c, m, ns) and tuple(centres.shape) == (b, m, 3)
        with torch.cuda.device(xyz.device):
            _lib.call("nesie_query_and_group_forward", b, c, n, m, ns, _ptr(xyz), _ptr(centres),
                      0 if features is None else _ptr(features), _ptr(idx), float(radius),
                      _ptr(out), _stream(xyz))

    def query_and_group_backward(self, grad_out, idx, grad_features):
        """grad_features (B,C,N) zeroed += channels 3.. of grad_out (B, 3+C, M, ns)."""
        _check(grad_out, idx, grad_features); _f32(grad_out, grad_features); _i32(idx)
        b, c, n = grad_features.shape
        m, ns = idx.shape[1], idx.shape[2]
        assert tuple(grad_out.shape) == (b, 3 + c, m, ns)
        with torch.cuda.device(grad_out.device):
            _lib.call("nesie_query_and_group_backward", b, c, n, m, ns, _ptr(grad_out), _ptr(idx),
                      _ptr(grad_features), _stream(grad_out))

    def three_interpolate_grad_csr(self, grad_out, weight, order, sources, grad_points):
        """grad_points (B,C,m) = scatter of weight * grad_out (B,C,n) through (order, sources) =
        inverted_index(idx (B,n,3), m); WRITTEN in full (no zero fill needed).  grad_out may be a
        channel slice of a wider tensor (batch stride free, rows contiguous)."""
        _check(weight, order, sources, grad_points); _f32(grad_out, weight, grad_points)
        _i32(order, sources)
        b, c, n = grad_out.shape
        m = grad_points.shape[2]
        assert grad_out.is_cuda and grad_out.stride(2) == 1 and grad_out.stride(1) == n
        assert weight.numel() == b * n * 3 and tuple(order.shape) == (b, n * 3)
        with torch.cuda.device(grad_out.device):
            _lib.call("nesie_three_interpolate_grad_csr", b, c, n, m, _ptr(grad_out),
                      grad_out.stride(0) if b > 1 else c * n, _ptr(weight), _ptr(order), _ptr(sources),
                      _ptr(grad_points), _stream(grad_out))

    def inverted_index(self, idx, n):
        """idx (B, M, ns) int32 in [0, n) -> order, sources (B, M*ns) int32: the grouped columns
        sorted by source point (ascending column inside a point's run) and that point for each."""
        _check(idx); _i32(idx)
        b = idx.shape[0]
        e = idx.numel() // b
        order = torch.empty(b, e, dtype=torch.int32, device=idx.device)
        sources = torch.empty(b, e, dtype=torch.int32, device=idx.device)
        scratch = torch.empty(b, e, dtype=torch.int32, device=idx.device)
        with torch.cuda.device(idx.device):
            _lib.call("nesie_inverted_index", b, n, e, _ptr(idx), _ptr(order), _ptr(sources),
                      _ptr(scratch), _stream(idx))
        return order, sources

    def vote_finish_forward(self, raw, seed_points, seed_feats, normalise):
        """-> (vote_points (B, N, 3), vote_feats (B, C, N), inv_norm (B, N)) (nesie_vote_finish_forward)."""
        _check(raw, seed_points, seed_feats); _f32(raw, seed_points, seed_feats)
        b, c, n = seed_feats.shape
        assert tuple(raw.shape) == (b, 3 + c, n) and tuple(seed_points.shape) == (b, n, 3)
        vp, vf = torch.empty_like(seed_points), torch.empty_like(seed_feats)
        inv = torch.empty(b, n, dtype=torch.float32, device=raw.device)
        with torch.cuda.device(raw.device):
            _lib.call("nesie_vote_finish_forward", b, c, n, int(bool(normalise)), _ptr(raw), _ptr(seed_points),
                      _ptr(seed_feats), _ptr(vp), _ptr(vf), _ptr(inv), _stream(raw))
        return vp, vf, inv

    def vote_finish_backward(self, g_feats, g_points, vote_feats, inv_norm, normalise):
        """-> d_raw (B, 3 + C, N) (nesie_vote_finish_backward); either gradient may be None."""
        _check(vote_feats, inv_norm); _f32(vote_feats, inv_norm)
        b, c, n = vote_feats.shape
        for t, shape in ((g_feats, (b, c, n)), (g_points, (b, n, 3))):
            if t is not None:
                _check(t); _f32(t)
                assert tuple(t.shape) == shape
        d_raw = torch.empty(b, 3 + c, n, dtype=torch.float32, device=vote_feats.device)
        opt = lambda t: 0 if t is None else _ptr(t)  # noqa: E731
        with torch.cuda.device(vote_feats.device):
            _lib.call("nesie_vote_finish_backward", b, c, n, int(bool(normalise)), opt(g_feats), opt(g_points),
                      _ptr(vote_feats), _ptr(inv_norm), _ptr(d_raw), _stream(vote_feats))
        return d_raw

    def gather_rows3(self, xyz, sample):
        """xyz (B, N, 3), sample (B, M) int32 -> xyz[b, sample[b, m]] (B, M, 3)."""
        _check(xyz, sample); _f32(xyz); _i32(sample)
        b, n = xyz.shape[:2]
        m = sample.shape[1]
        out = torch.empty(b, m, 3, dtype=torch.float32, device=xyz.device)
        with torch.cuda.device(xyz.device):
            _lib.call("nesie_gather_rows3", b, n, m, _ptr(xyz), _ptr(sample), _ptr(out), _stream(xyz))
        return out

    def query_and_group_backward_xyz(self, grad_out, radius, order, sources, sample, d_centres, n):
        """Coordinate gradient (B, N, 3) of QueryAndGroup over network-computed coordinates whose
        centres are xyz[sample] (nesie_query_and_group_backward_xyz); grad_out (B, 3+C, M, ns),
        d_centres (B, M, 3) or None."""
        _check(grad_out, order, sources, sample); _f32(grad_out); _i32(order, sources, sample)
        b, c3, m, ns = grad_out.shape
        assert tuple(order.shape) == (b, m * ns) == tuple(sources.shape) and tuple(sample.shape) == (b, m)
        if d_centres is not None:
            _check(d_centres); _f32(d_centres)
            assert tuple(d_centres.shape) == (b, m, 3)
        d_xyz = torch.empty(b, n, 3, dtype=torch.float32, device=grad_out.device)
        with torch.cuda.device(grad_out.device):
            _lib.call("nesie_query_and_group_backward_xyz", b, c3 - 3, n, m, ns, float(radius),
                      _ptr(grad_out), _ptr(order), _ptr(sources), _ptr(sample),
                      0 if d_centres is None else _ptr(d_centres), _ptr(d_xyz), _stream(grad_out))
        return d_xyz

    def query_and_group_backward_csr(self, grad_out, idx_shape, order, offsets, grad_features):
        """grad_features (B,C,N) = scatter of channels 3.. of grad_out through (order, sources);
        WRITTEN in full (no zero fill needed)."""
        _check(grad_out, order, offsets, grad_features); _f32(grad_out, grad_features)
        _i32(order, offsets)
        b, c, n = grad_features.shape
        m, ns = idx_shape[1], idx_shape[2]
        assert tuple(grad_out.shape) == (b, 3 + c, m, ns)
        assert tuple(order.shape) == (b, m * ns) and tuple(offsets.shape) == (b, m * ns)
        with torch.cuda.device(grad_out.device):
            _lib.call("nesie_query_and_group_backward_csr", b, c, n, m, ns, _ptr(grad_out),
                      _ptr(order), _ptr(offsets), _ptr(grad_features), _stream(grad_out))

    def group_points_backward_csr(self, grad_out, order, sources, grad_points):
        """grad_points (B,C,N) = scatter of grad_out (B,C,M,ns) through (order, sources) =
        inverted_index(idx (B,M,ns), N): no float atomics, WRITTEN in full
        (nesie_group_points_backward_csr)."""
        _check(grad_out, order, sources, grad_points); _f32(grad_out, grad_points); _i32(order, sources)
        b, c, n = grad_points.shape
        m, ns = grad_out.shape[2], grad_out.shape[3]
        assert grad_out.shape[:2] == (b, c) and tuple(order.shape) == (b, m * ns) == tuple(sources.shape)
        with torch.cuda.device(grad_out.device):
            _lib.call("nesie_group_points_backward_csr", b, c, n, m, ns, _ptr(grad_out), _ptr(order),
                      _ptr(sources), _ptr(grad_points), _stream(grad_out))

    def three_nn_wrapper(self, b, n, m, unknown, known, dist2, idx):
        _check(unknown, known, dist2, idx); _f32(unknown, known, dist2); _i32(idx)
        assert unknown.numel() == b * n * 3 and known.numel() == b * m * 3
        assert dist2.numel() == b * n * 3 and idx.numel() == b * n * 3
        with torch.cuda.device(unknown.device):
            _lib.call("nesie_three_nn_wrapper", b, n, m, _ptr(unknown), _ptr(known),
                      _ptr(dist2), _ptr(idx), _stream(unknown))

    def three_interpolate_wrapper(self, b, c, m, n, points, idx, weight, out):
        _check(points, idx, weight, out); _f32(points, weight, out); _i32(idx)
        assert points.numel() == b * c * m and idx.numel() == b * n * 3
        assert weight.numel() == b * n * 3 and out.numel() == b * c * n
        with torch.cuda.device(points.device):
            _lib.call("nesie_three_interpolate_wrapper", b, c, m, n, _ptr(points),
                      _ptr(idx), _ptr(weight), _ptr(out), _stream(points))

    def side_decode_forward(self, reg, agg, scale, sign, probs, surface, bbox):
        _check(reg, agg, scale, sign, probs, surface, bbox); _f32(reg, agg, probs, surface, bbox)
        b, cch, k = reg.shape
        bins = (cch - 2) // 6
        assert tuple(probs.shape) == (b, 6, bins, k) and tuple(bbox.shape) == (b, k, 7)
        with torch.cuda.device(reg.device):
            _lib.call("nesie_side_decode_forward", b, k, bins, _ptr(reg), _ptr(agg), _ptr(scale),
                      _ptr(sign), _ptr(probs), _ptr(surface), _ptr(bbox), _stream(reg))

    def side_decode_backward(self, reg, probs, scale, sign, d_surface, d_bbox, d_reg, d_agg):
        _check(reg, probs, scale, sign, d_reg, d_agg)
        b, cch, k = reg.shape
        bins = (cch - 2) // 6
        opt = lambda t: 0 if t is None else _ptr(t)  # noqa: E731
        for t in (d_surface, d_bbox):
            if t is not None:
                _check(t); _f32(t)
        with torch.cuda.device(reg.device):
            _lib.call("nesie_side_decode_backward", b, k, bins, _ptr(reg), _ptr(probs),
                      _ptr(scale), _ptr(sign), opt(d_surface), opt(d_bbox), _ptr(d_reg),
                      _ptr(d_agg), _stream(reg))

    def bn_eval_coef(self, gamma, beta, running_mean, running_var, eps, coef):
        """coef (C,4) <- (gamma / sqrt(var + eps), beta - mean * scale, 0, 0); one launch."""
        _check(running_mean, running_var, coef); _f32(running_mean, running_var, coef)
        c = running_mean.numel()
        assert tuple(coef.shape) == (c, 4) and coef.is_contiguous()
        opt = lambda t: 0 if t is None else _ptr(t)  # noqa: E731
        with torch.cuda.device(coef.device):
            _lib.call("nesie_bn_eval_coef", c, opt(gamma), opt(beta), _ptr(running_mean),
                      _ptr(running_var), float(eps), _ptr(coef), _stream(coef))

    def affine_relu_forward(self, x, coef, relu, y, row_bias=None):
        """Evaluation-mode norm: y = relu?(coef[c,0] * (x + row_bias) + coef[c,1]); x, y (B,C,*)."""
        _check(x, coef, y); _f32(x, coef, y)
        b, c = x.shape[:2]
        p = x.numel() // (b * c) if b * c else 0
        assert tuple(coef.shape) == (c, 4) and y.shape == x.shape
        group = _row_bias_group(x, row_bias)
        with torch.cuda.device(x.device):
            _lib.call("nesie_affine_relu_forward", b, c, p, _ptr(x), _ptr(coef), int(bool(relu)),
                      0 if row_bias is None else _ptr(row_bias), group, _ptr(y), _stream(x))

    def affine_relu_maxpool_forward(self, x, coef, pooled, argmax):
        """x (B,C,M,ns) -> pooled (B,C,M) = max_ns relu(coef[c,0] * x + coef[c,1]), argmax u8."""
        _check(x, coef, pooled, argmax); _f32(x, coef, pooled)
        b, c, m, ns = x.shape
        assert tuple(coef.shape) == (c, 4) and tuple(pooled.shape) == (b, c, m)
        assert argmax.dtype == torch.uint8 and tuple(argmax.shape) == (b, c, m)
        with torch.cuda.device(x.device):
            _lib.call("nesie_affine_relu_maxpool_forward", b, c, m, ns, _ptr(x), _ptr(coef),
                      _ptr(pooled), _ptr(argmax), _stream(x))

    def mlp_stream_forward(self, x, w, y, stat_partial=None, in_coef=None, in_relu=False):
        """y[b] = W . act(x[b]) on the matrix cores, streaming form (Cin <= 64, Cout <= 128):
        x (B,Cin,P), w (Cout,Cin), y (B,Cout,P); stat_partial (parts, Cout, 2) receives the
        per-workgroup (sum, sum of squares) of y, parts = mlp_stream_parts(B, P)."""
        _check(x, w); _f32(x, w)
        b, cin, p = x.shape
        cout = w.shape[0]
        assert tuple(w.shape) == (cout, cin)
        if y is not None:       # (None: only the statistics of y are wanted)
            _check(y); _f32(y)
            assert tuple(y.shape) == (b, cout, p)
        else:
            assert stat_partial is not None
        if stat_partial is not None:
            _check(stat_partial); _f32(stat_partial)
            assert tuple(stat_partial.shape) == (self.mlp_stream_parts(b, p), cout, 2)
        with torch.cuda.device(x.device):
            _lib.call("nesie_mlp_layer_forward_stream", b, cin, cout, p, _ptr(x), cin * p,
                      _ptr(w), 0 if in_coef is None else _ptr(in_coef), int(bool(in_relu)),
                      0 if y is None else _ptr(y), 0 if stat_partial is None else _ptr(stat_partial),
                      _stream(x))

    @staticmethod
    def mlp_stream_parts(b, p):
        return int(_lib.load().nesie_mlp_stream_partials(b, p))

    def aligned_3d_nms(self, boxes, scores, classes, valid, thr, picks, count):
        """boxes (B,K,6), scores (B,K), classes (B,K) i32, valid (B,K) u8 or None ->
        picks (B,K) i32 (-1 padded, pick order), count (B) i32."""
        _check(boxes, scores, classes, picks, count); _f32(boxes, scores); _i32(classes, picks, count)
        b, k = scores.shape
        assert tuple(boxes.shape) == (b, k, 6) and tuple(classes.shape) == (b, k)
        assert tuple(picks.shape) == (b, k) and count.numel() == b
        if valid is not None:
            _check(valid)
            assert valid.dtype == torch.uint8 and tuple(valid.shape) == (b, k)
        with torch.cuda.device(boxes.device):
            _lib.call("nesie_aligned_3d_nms", b, k, _ptr(boxes), _ptr(scores), _ptr(classes),
                      0 if valid is None else _ptr(valid), float(thr), _ptr(picks), _ptr(count),
                      _stream(boxes))

    def points_in_boxes_count(self, boxes, pts, counts):
        """boxes (B,T,7) LiDAR frame, pts (B,M,3) -> counts (B,T) i32."""
        _check(boxes, pts, counts); _f32(boxes, pts); _i32(counts)
        b, t, _ = boxes.shape
        assert pts.shape[0] == b and pts.shape[2] == 3 and tuple(counts.shape) == (b, t)
        with torch.cuda.device(boxes.device):
            _lib.call("nesie_points_in_boxes_count", b, t, pts.shape[1], _ptr(boxes), _ptr(pts),
                      _ptr(counts), _stream(boxes))

    def boxes_overlap_bev(self, boxes_a, boxes_b, ans_overlap):
        """(N,5), (M,5) rotated BEV rectangles (x1,y1,x2,y2,angle) -> overlap areas (N,M)."""
        _check(boxes_a, boxes_b, ans_overlap); _f32(boxes_a, boxes_b, ans_overlap)
        n, m = boxes_a.shape[0], boxes_b.shape[0]
        assert boxes_a.shape[1] == 5 and boxes_b.shape[1] == 5
        assert tuple(ans_overlap.shape) == (n, m)
        with torch.cuda.device(boxes_a.device):
            _lib.call("nesie_boxes_overlap_bev", n, _ptr(boxes_a), m, _ptr(boxes_b),
                      _ptr(ans_overlap), _stream(boxes_a))

    def scene_assemble(self, pool, height, choices, xform, out):
        """pool (R,3), height (R), choices (B,n) i32 rows of the pool, xform (B,20) -> out (B,n,4)."""
        _check(pool, height, choices, xform, out); _f32(pool, height, xform, out); _i32(choices)
        b, n = choices.shape
        assert pool.dim() == 2 and pool.shape[1] == 3 and height.numel() == pool.shape[0]
        assert tuple(xform.shape) == (b, 20) and tuple(out.shape) == (b, n, 4)
        with torch.cuda.device(pool.device):
            _lib.call("nesie_scene_assemble", b, n, pool.shape[0], _ptr(pool), _ptr(height),
                      _ptr(choices), _ptr(xform), _ptr(out), _stream(pool))

    def grid_taps(self, centre, size, heading, mult, plane, known):
        """-> idx (B,K*gp,3) int32, weight, rel (B,K*gp,3) for the gp grid points per proposal."""
        _check(centre, size, heading, mult, plane, known)
        _f32(centre, size, heading, mult, plane, known)
        b, k = centre.shape[:2]
        gp, m = mult.shape[0], known.shape[1]
        assert tuple(size.shape) == (b, k, 3) and tuple(heading.shape) == (b, k)
        assert tuple(plane.shape) == (gp, 3) and known.shape[0] == b
        idx = torch.empty(b, k * gp, 3, dtype=torch.int32, device=centre.device)
        weight = torch.empty(b, k * gp, 3, dtype=torch.float32, device=centre.device)
        rel = torch.empty_like(weight)
        with torch.cuda.device(centre.device):
            _lib.call("nesie_grid_taps", b, k, gp, m, _ptr(centre), _ptr(size), _ptr(heading),
                      _ptr(mult), _ptr(plane), _ptr(known), _ptr(idx), _ptr(weight), _ptr(rel),
                      _stream(centre))
        return idx, weight, rel

    def blend_conv_forward(self, table, seg_off, idx, weight, rel, wx, out, segs, seg_len,
                           c, c_offset, stat_partial=None):
        """table (B, M, pitch) point-major; out (B, segs, c_total, n/segs): query (k, s, g) ->
        out[b, s, c_offset+ch, k*seg_len+g] = blend(table[..., s*seg_off+ch]) (+ wx[s,ch].rel)."""
        _check(table, idx, weight, out); _f32(table, weight, out); _i32(idx)
        b, m, pitch = table.shape
        n = idx.shape[1]
        assert idx.numel() == b * n * 3 and weight.numel() == b * n * 3
        assert n % (segs * seg_len) == 0 and out.dim() == 4
        assert out.shape[0] == b and out.shape[1] == segs and out.shape[3] * segs == n
        assert c_offset + c <= out.shape[2] and (rel is None) == (wx is None)
        if wx is not None:
            _check(rel, wx); _f32(rel, wx)
            assert rel.numel() == b * n * 3 and tuple(wx.shape) == (segs, c, 3)
        opt = lambda t: 0 if t is None else _ptr(t)  # noqa: E731
        with torch.cuda.device(table.device):
            if stat_partial is not None:
                _check(stat_partial); _f32(stat_partial)
                assert stat_partial.numel() == segs * c * (b * (n // segs // 64)) * 2
            _lib.call("nesie_blend_conv_forward", b, c, m, n, _ptr(table), pitch, seg_off,
                      _ptr(idx), _ptr(weight), opt(rel), opt(wx), _ptr(out), segs, seg_len,
                      int(out.shape[2]), c_offset, opt(stat_partial), _stream(table))

    # Deterministic backward -- THE DEFAULT since round 5 (NESIE_DETERMINISTIC=0 or
    # ``set_deterministic(False)`` restores the reference's atomicAdd scatters,
    # three_interpolate_cuda.cu:61-84, group_points_cuda.cu:10-31): every scatter-add of the backward
    # pass runs in an order fixed by the indices alone --
    #  * the blend backward stages its rows and gathers them per seed (nesie_blend_conv_backward_staged);
    #  * group_points / QueryAndGroup / gather_points / three_interpolate backward go through an
    #    inverted index of their indices (``scatter_index``), built on the spot when the caller did
    #    not hand one over (any number of source points);
    # so two runs of the same step give the same bits in every gradient (tests/test_parity_gpu.py).
    DETERMINISTIC = os.environ.get('NESIE_DETERMINISTIC', '1') != '0'

    @classmethod
    def set_deterministic(cls, on=True):
        """Bitwise-reproducible backward (see DETERMINISTIC); -> the previous setting."""
        prev, cls.DETERMINISTIC = cls.DETERMINISTIC, bool(on)
        return prev

    def scatter_index(self, idx, n):
        """(order, sources) = inverted_index(idx, n) when the deterministic backward is on and the
        caller has none, else None (-> the atomic kernels)."""
        if not self.DETERMINISTIC:
            return None
        return self.inverted_index(idx.contiguous(), n)

    def blend_backward_writes_table(self, c, n, segs, m):
        """True when ``blend_conv_backward`` WRITES d_table in full (the staged form): the caller
        then need not zero it."""
        return self.DETERMINISTIC and c % 64 == 0 and c <= 256 and (n // segs) % 64 == 0 and m + 1 <= 2048

    def blend_conv_backward(self, dy, seg_off, idx, weight, rel, d_table, d_wx, segs, seg_len,
                            bn_z=None, bnb=None):
        """dy (B, segs, c, n/segs) -> d_table (B, M, pitch) columns [s*seg_off, +c) and
        sum(dy x rel) WRITTEN to d_wx (segs, c, 3).  bn_z / bnb: dy is the gradient of
        relu(bn(bn_z)) and the norm backward runs on the tile load (bnb (segs*c, 8) from
        ``pw_bnb_coef``).  Staged form (``blend_backward_writes_table``): no float atomics, d_table
        written in full, bitwise reproducible (nesie_blend_conv_backward_staged); otherwise rows
        are ADDED into d_table (zeroed by the caller) with atomics."""
        _check(dy, idx, weight, d_table); _f32(dy, weight, d_table); _i32(idx)
        b, m, pitch = d_table.shape
        n, c = idx.shape[1], dy.shape[2]
        assert tuple(dy.shape) == (b, segs, c, n // segs)
        if d_wx is not None:
            _check(rel, d_wx); _f32(rel, d_wx)
            assert tuple(d_wx.shape) == (segs, c, 3)
        opt = lambda t: 0 if t is None else _ptr(t)  # noqa: E731
        with torch.cuda.device(dy.device):
            part = None
            if d_wx is not None:   # one partial per workgroup, summed here
                runs = _lib.load().nesie_blend_conv_runs(n, segs)
                part = torch.empty(b * runs, segs, c, 3, dtype=torch.float32, device=dy.device)
            if bnb is not None:
                _check(bn_z, bnb); _f32(bn_z, bnb)
                assert tuple(bn_z.shape) == tuple(dy.shape) and tuple(bnb.shape) == (segs * c, 8)
            if self.DETERMINISTIC and not self.blend_backward_writes_table(c, n, segs, m):
                raise RuntimeError(
                    f'blend_conv_backward: no fixed-order form for c={c}, {n // segs} queries per face, '
                    f'{m} seeds (needs c in 64..256 step 64, faces of whole 64-query tiles, <= 2047 '
                    'seeds); HipKernels.set_deterministic(False) selects the atomic scatter')
            if self.blend_backward_writes_table(c, n, segs, m):
                assert seg_off == c or segs == 1, 'staged form: one column block per face'
                need = _lib.load().nesie_blend_conv_backward_workspace_bytes(b, c, n, segs)
                ws = torch.empty(need, dtype=torch.uint8, device=dy.device)
                _lib.call("nesie_blend_conv_backward_staged", b, c, m, n, _ptr(dy), opt(bn_z), opt(bnb),
                          pitch, seg_off, _ptr(idx), _ptr(weight), opt(rel), _ptr(d_table), opt(part),
                          segs, seg_len, _ptr(ws), need, _stream(dy))
            elif bnb is not None:
                _lib.call("nesie_blend_conv_backward_bn", b, c, m, n, _ptr(dy), _ptr(bn_z), _ptr(bnb),
                          pitch, seg_off, _ptr(idx), _ptr(weight), opt(rel), _ptr(d_table), opt(part),
                          segs, seg_len, _stream(dy))
            else:
                _lib.call("nesie_blend_conv_backward", b, c, m, n, _ptr(dy), pitch, seg_off,
                          _ptr(idx), _ptr(weight), opt(rel), _ptr(d_table), opt(part), segs, seg_len,
                          _stream(dy))
            if part is not None:      # (d_wx arrives zero-filled or empty from its caller: the sum overwrites it)
                torch.sum(part, 0, out=d_wx)

    def blend_conv_bn_forward(self, table, idx, weight, rel, wx, gamma, beta, running_mean,
                              running_var, momentum, eps, out, save_mean, save_invstd, fwd_coef,
                              segs, seg_len):
        """out (B, segs, c, n/segs) = relu(bn(blend conv)); table (B, M, segs*c)."""
        _check(table, idx, weight, rel, wx, out, save_mean, save_invstd, fwd_coef)
        _f32(table, weight, rel, wx, out); _i32(idx)
        b, m, pitch = table.shape
        n, c = idx.shape[1], pitch // segs
        assert tuple(out.shape) == (b, segs, c, n // segs) and tuple(wx.shape) == (segs, c, 3)
        lib = _lib.load()
        need = lib.nesie_blend_conv_bn_workspace_bytes(b, c, n, segs)
        opt = lambda t: 0 if t is None else _ptr(t)  # noqa: E731
        with torch.cuda.device(table.device):
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=table.device)
            _lib.call("nesie_blend_conv_bn_forward", b, c, m, n, _ptr(table), pitch, c, _ptr(idx),
                      _ptr(weight), _ptr(rel), _ptr(wx), opt(gamma), opt(beta),
                      opt(running_mean), opt(running_var), float(momentum), float(eps), _ptr(out),
                      _ptr(save_mean), _ptr(save_invstd), _ptr(fwd_coef), _ptr(ws), need, segs,
                      seg_len, _stream(table))

    def blend_conv_bn_backward(self, dy, table, idx, weight, rel, wx, gamma, save_invstd,
                               fwd_coef, d_table, d_wx, dgamma, dbeta, segs, seg_len):
        """d_table (zeroed) += ..., d_wx (segs, c, 3) += ..., dgamma / dbeta [segs*c] written."""
        _check(dy, table, idx, weight, rel, wx, d_table, d_wx, dgamma, dbeta)
        _f32(dy, table, d_table); _i32(idx)
        b, m, pitch = table.shape
        n, c = idx.shape[1], pitch // segs
        assert tuple(dy.shape) == (b, segs, c, n // segs) and d_table.shape == table.shape
        lib = _lib.load()
        need = lib.nesie_blend_conv_bn_workspace_bytes(b, c, n, segs)
        runs = lib.nesie_blend_conv_runs(n, segs)
        opt = lambda t: 0 if t is None else _ptr(t)  # noqa: E731
        with torch.cuda.device(dy.device):
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dy.device)
            part = torch.empty(b * runs, segs, c, 3, dtype=torch.float32, device=dy.device)
            _lib.call("nesie_blend_conv_bn_backward", b, c, m, n, _ptr(dy), _ptr(table), pitch, c,
                      _ptr(idx), _ptr(weight), _ptr(rel), _ptr(wx), opt(gamma),
                      _ptr(save_invstd), _ptr(fwd_coef), _ptr(d_table), _ptr(part), _ptr(dgamma),
                      _ptr(dbeta), _ptr(ws), need, segs, seg_len, _stream(dy))
            d_wx += part.sum(0)

    def three_interpolate_grad_wrapper(self, b, c, n, m, grad_out, idx, weight,
                                       grad_points):
        _check(grad_out, idx, weight, grad_points); _f32(grad_out, weight, grad_points)
        _i32(idx)
        assert grad_out.numel() == b * c * n and idx.numel() == b * n * 3
        assert weight.numel() == b * n * 3 and grad_points.numel() == b * c * m
        with torch.cuda.device(grad_out.device):
            _lib.call("nesie_three_interpolate_grad_wrapper", b, c, n, m, _ptr(grad_out),
                      _ptr(idx), _ptr(weight), _ptr(grad_points), _stream(grad_out))

    def sort_vertices_forward(self, vertices, mask, num_valid, idx):
        _check(vertices, mask, num_valid, idx); _f32(vertices); _i32(num_valid, idx)
        if mask.dtype != torch.bool:
            raise TypeError("mask must be bool")
        b, n, m, two = vertices.shape
        assert two == 2 and tuple(mask.shape) == (b, n, m)
        assert tuple(num_valid.shape) == (b, n) and tuple(idx.shape) == (b, n, 9)
        with torch.cuda.device(vertices.device):
            _lib.call("nesie_sort_vertices_forward", b, n, m, _ptr(vertices), _ptr(mask),
                      _ptr(num_valid), _ptr(idx), _stream(vertices))

    def points_in_boxes_batch(self, boxes, pts, out):
        _check(boxes, pts, out); _f32(boxes, pts); _i32(out)
        b, t, seven = boxes.shape
        assert seven == 7 and pts.shape[0] == b and pts.shape[2] == 3
        m = pts.shape[1]
        assert tuple(out.shape) == (b, m, t)
        with torch.cuda.device(boxes.device):
            _lib.call("nesie_points_in_boxes_batch", b, t, m, _ptr(boxes), _ptr(pts),
                      _ptr(out), _stream(boxes))

    def vote_targets(self, points, gt_boxes, gt_count):
        """points (B,N,C>=3), depth-frame gt_boxes (B,T,7), gt_count (B) int64 -> vote targets
        (B,N,9) and masks (B,N) int64 of get_targets_single, one launch."""
        _check(points, gt_boxes, gt_count); _f32(points, gt_boxes)
        b, n, c = points.shape
        t = gt_boxes.shape[1]
        assert gt_boxes.shape == (b, t, 7) and gt_count.shape == (b,) and gt_count.dtype == torch.int64
        votes = points.new_empty(b, n, 9)
        masks = torch.empty(b, n, dtype=torch.int64, device=points.device)
        with torch.cuda.device(points.device):
            _lib.call("nesie_vote_targets", b, t, n, c, _ptr(gt_boxes) if t else 0,
                      _ptr(gt_count), _ptr(points), _ptr(votes), _ptr(masks), _stream(points))
        return votes, masks

    def group_max_pool_forward(self, x, out, argmax):
        """x (..., ns) -> out (...), argmax (...) uint8."""
        _check(x, out, argmax); _f32(x, out)
        ns = x.shape[-1]
        rows = x.numel() // ns
        assert out.numel() == rows and argmax.numel() == rows and argmax.dtype == torch.uint8
        with torch.cuda.device(x.device):
            _lib.call("nesie_group_max_pool_forward", rows, ns, _ptr(x), _ptr(out),
                      _ptr(argmax), _stream(x))

    def group_max_pool_backward(self, grad_out, argmax, grad_x):
        _check(grad_out, argmax, grad_x); _f32(grad_out, grad_x)
        ns = grad_x.shape[-1]
        rows = grad_x.numel() // ns
        assert grad_out.numel() == rows and argmax.numel() == rows
        with torch.cuda.device(grad_x.device):
            _lib.call("nesie_group_max_pool_backward", rows, ns, _ptr(grad_out),
                      _ptr(argmax), _ptr(grad_x), _stream(grad_x))


    def group_max_pool_backward_add(self, grad_out, argmax, grad_x):
        """grad_x[..., argmax] += grad_out, in place (grad_x (..., ns) contiguous)."""
        _check(grad_out, argmax, grad_x); _f32(grad_out, grad_x)
        ns = grad_x.shape[-1]
        rows = grad_x.numel() // ns
        assert grad_out.numel() == rows and argmax.numel() == rows
        with torch.cuda.device(grad_x.device):
            _lib.call("nesie_group_max_pool_backward_add", rows, ns, _ptr(grad_out),
                      _ptr(argmax), _ptr(grad_x), _stream(grad_x))

    def channel_sum(self, x, out=None, ng=1):
        """x (NB, C, P) (batch stride free, each x[n] (C, P) contiguous) -> (ng * C,) sums over the
        batch entries n = g (mod ng) and the positions, in a fixed order (nesie_channel_sum): a conv
        bias's gradient (ng > 1: S stacked nets whose batch axis runs (scene, net))."""
        _f32(x)
        nb, c, p = x.shape
        assert x.is_cuda and x.stride(2) == 1 and x.stride(1) == p and nb % ng == 0
        if out is None:
            out = torch.empty(ng * c, dtype=torch.float32, device=x.device)
        _check(out); _f32(out)
        assert out.numel() == ng * c
        with torch.cuda.device(x.device):
            _lib.call("nesie_channel_sum", nb, ng, c, p, _ptr(x), x.stride(0) if nb > 1 else c * p, _ptr(out),
                      _stream(x))
        return out

    def lhs_nms_samecls(self, boxes, thr, keep):
        """boxes (B,K,8) f32, keep (B,K) uint8."""
        _check(boxes, keep); _f32(boxes)
        b, k, eight = boxes.shape
        assert eight == 8 and tuple(keep.shape) == (b, k) and keep.dtype == torch.uint8
        with torch.cuda.device(boxes.device):
            _lib.call("nesie_lhs_nms_samecls", b, k, _ptr(boxes), float(thr), _ptr(keep),
                      _stream(boxes))

    def iou3d_forward(self, box1, box2, iou, jac):
        """box1, box2 (n,7); iou (n,); jac (n,7) or None."""
        _check(box1, box2, iou); _f32(box1, box2, iou)
        n = iou.numel()
        assert box1.numel() == n * 7 and box2.numel() == n * 7
        if jac is not None:
            _check(jac); _f32(jac)
            assert jac.numel() == n * 7
        with torch.cuda.device(box1.device):
            _lib.call("nesie_iou3d_forward", n, _ptr(box1), _ptr(box2), _ptr(iou),
                      0 if jac is None else _ptr(jac), _stream(box1))

    def conv_wgrad(self, dy, x, dw, x_coef=None, x_relu=False, bn_z=None, bnb=None):
        """dw (cout, cin) = sum_b dy[b] (cout, P) @ act(x[b]) (cin, P)^T on the matrix cores;
        dy and x may be batch-strided views (each dy[b], x[b] contiguous).  bn_z / bnb: dy is the
        gradient of relu(bn(bn_z)) and the norm backward runs on the load (nesie_conv_wgrad_bn)."""
        _f32(dy, x, dw)
        if not (dy.is_cuda and dw.is_contiguous()):
            raise ValueError("conv_wgrad: HIP tensors, dw contiguous")
        b, cout, p = dy.shape
        cin = x.shape[1]
        assert x.shape[0] == b and x.shape[2] == p and tuple(dw.shape) == (cout, cin)
        assert x.stride(2) == 1 and x.stride(1) == p, "each x[b] must be (cin, P) contiguous"
        assert dy.stride(2) == 1 and dy.stride(1) == p, "each dy[b] must be (cout, P) contiguous"
        lib = _lib.load()
        need = lib.nesie_conv_wgrad_workspace_bytes(b, cout, cin, p)
        with torch.cuda.device(dy.device):
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dy.device)
            if bnb is not None:
                _check(bn_z, bnb); _f32(bn_z, bnb)
                assert tuple(bn_z.shape) == tuple(dy.shape) and dy.is_contiguous() and tuple(bnb.shape) == (cout, 8)
                _lib.call("nesie_conv_wgrad_bn", b, cout, cin, p, _ptr(dy), _ptr(bn_z), cout * p, _ptr(bnb),
                          _ptr(x), x.stride(0) if b > 1 else cin * p,
                          0 if x_coef is None else _ptr(x_coef), int(bool(x_relu)), _ptr(dw), _ptr(ws),
                          need, _stream(dy))
                return
            _lib.call("nesie_conv_wgrad", b, cout, cin, p, _ptr(dy),
                      dy.stride(0) if b > 1 else cout * p, _ptr(x),
                      x.stride(0) if b > 1 else cin * p, 0 if x_coef is None else _ptr(x_coef),
                      int(bool(x_relu)), _ptr(dw), _ptr(ws), need, _stream(dy))

    def pw_wgrad_supported(self, co, ci, p):
        return bool(_lib.load().nesie_pw_wgrad_supported(int(co), int(ci), int(p)))

    def pw_wgrad_tiled(self, nb, ng, co, ci, p):
        """True when ``pw_wgrad`` serves this shape as ONE launch over 64 x 64 blocks of the product
        (wide layers over few positions: the 1-D chains)."""
        return bool(_lib.load().nesie_pw_wgrad_tiled(int(nb), int(ng), int(co), int(ci), int(p)))

    # ---- deferred weight-gradient reductions (nesie_pw_wgrad_deferred, include/nesie_ops.h) -----
    # Between ``begin_deferred_reductions()`` and ``flush_deferred_reductions()`` a weight gradient
    # whose destination is FINAL (a slot of the flat gradient vector: nothing reads it before the
    # optimiser) leaves the addition of its partials pending; the flush runs them all in one launch.
    # dp.FlatTrainState opens the window in begin() and flushes in collect().  NESIE_DEFER_WGRAD=0:
    # every reduction runs with its own launch (A/B switch).
    DEFER_WGRAD = os.environ.get('NESIE_DEFER_WGRAD', '1') != '0'
    _deferred = None       # None: no window open; else the workspaces (and destinations) kept alive

    @classmethod
    def begin_deferred_reductions(cls):
        if cls._deferred:      # a window left open by an aborted backward: its partials are stale
            _lib.call("nesie_pw_wgrad_drop_deferred")
        cls._deferred = [] if cls.DEFER_WGRAD else None

    @classmethod
    def flush_deferred_reductions(cls, device=None, close=False):
        """Finish every pending weight gradient (one launch on the current stream); -> the
        destinations that were written.  ``close``: end the window."""
        held, done = cls._deferred, []
        if held:
            dev = device if device is not None else held[0][1].device
            with torch.cuda.device(dev):
                _lib.call("nesie_pw_wgrad_flush_deferred", torch.cuda.current_stream(dev).cuda_stream)
            done = [dw for _, dw in held]
            # (under a graph capture the workspaces live in the graph's pool; in eager mode the caching
            # allocator hands a freed block to later launches of this stream only: both orders are safe)
            cls._deferred = []
        if close:
            cls._deferred = None
        return done

    def pw_wgrad(self, dy, x, dw, ng=1, x_coef=None, x_relu=True, final=False):
        """dw (ng, co, ci) = sum over n % ng == g and positions of dy[n] (co, P) act(x[n])^T
        (nesie_pw_wgrad); dy (NB, co, P), x (NB, ci, P) batch-strided views allowed;
        x_coef (ng*ci, 4) folded BatchNorm of x.  ``final``: dw is read by nobody before the
        deferred reductions are flushed (a slot of the flat gradient vector)."""
        _f32(dy, x, dw)
        nb, co, p = dy.shape
        ci = x.shape[1]
        assert dy.is_cuda and x.shape[0] == nb and x.shape[2] == p and nb % ng == 0
        assert dw.is_contiguous() and dw.numel() == ng * co * ci
        assert x.stride(2) == 1 and x.stride(1) == p and dy.stride(2) == 1 and dy.stride(1) == p
        if x_coef is not None:
            _check(x_coef); _f32(x_coef)
            assert tuple(x_coef.shape) == (ng * ci, 4)
        lib = _lib.load()
        need = lib.nesie_pw_wgrad_workspace_bytes(nb, ng, co, ci, p)
        defer = final and HipKernels._deferred is not None
        with torch.cuda.device(dy.device):
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=dy.device)
            _lib.call("nesie_pw_wgrad_deferred" if defer else "nesie_pw_wgrad", nb, ng, co, ci, p, _ptr(dy),
                      dy.stride(0) if nb > 1 else co * p, _ptr(x), x.stride(0) if nb > 1 else ci * p,
                      0 if x_coef is None else _ptr(x_coef), int(bool(x_relu)), _ptr(dw), _ptr(ws),
                      need, _stream(dy))
            if defer:
                HipKernels._deferred.append((ws, dw))

    def pw_bnb_coef(self, part, z_coef, gamma, count, dgamma, dbeta):
        """(channels, 8) reduction coefficients of a BatchNorm + ReLU backward from the partial sums
        part (channels, slots, 2) of ``pw_dgrad_bn_reduce`` (nesie_pw_bnb_coef); writes dgamma, dbeta."""
        _check(part, z_coef, dgamma, dbeta); _f32(part, z_coef, dgamma, dbeta)
        ch = part.shape[0]
        assert part.dim() == 3 and part.shape[2] == 2 and tuple(z_coef.shape) == (ch, 4)
        assert dgamma.numel() == ch == dbeta.numel()
        if gamma is not None:
            _check(gamma); _f32(gamma)
        with torch.cuda.device(part.device):
            bnb = torch.empty(ch, 8, dtype=torch.float32, device=part.device)
            _lib.call("nesie_pw_bnb_coef", ch, part.shape[1], float(count), _ptr(part), _ptr(z_coef),
                      0 if gamma is None else _ptr(gamma), _ptr(bnb), _ptr(dgamma), _ptr(dbeta),
                      _stream(part))
        return bnb

    def pw_wgrad_bn_supported(self, co, ci, p):
        return bool(_lib.load().nesie_pw_wgrad_bn_supported(int(co), int(ci), int(p)))

    def pw_wgrad_bn_backward(self, da, z, z_coef, gamma, part, x, dz, dw, dgamma, dbeta, ng=1,
                             x_coef=None, x_relu=True, d_row_bias=None, group=0, final=False):
        """The BatchNorm + ReLU backward's apply pass fused into the weight gradient it feeds
        (nesie_pw_wgrad_bn_backward): da (NB, co, P) gradient of relu(bn(z)), z raw conv output,
        z_coef (ng*co, 4), part (ng*co, slots, 2) from ``pw_dgrad_bn_reduce``; x (NB, ci, P) the
        layer's input with its own folded norm x_coef.  Writes dz (may be da), dw (ng, co, ci),
        dgamma, dbeta (ng*co); d_row_bias (NB, co, P / group), group 16 or 64: the gradient of the
        per-group row bias the forward added to z (zero-filled by the caller for group 64)."""
        _f32(da, z, x, dz, dw, dgamma, dbeta); _check(z_coef, part, dz); _f32(z_coef, part)
        nb, co, p = da.shape
        ci = x.shape[1]
        assert da.is_cuda and tuple(z.shape) == (nb, co, p) == tuple(dz.shape) and x.shape[0] == nb
        assert x.shape[2] == p and nb % ng == 0 and da.is_contiguous() and z.is_contiguous()
        assert dw.is_contiguous() and dw.numel() == ng * co * ci
        assert x.stride(2) == 1 and x.stride(1) == p
        assert tuple(z_coef.shape) == (ng * co, 4) and part.dim() == 3 and part.shape[0] == ng * co
        assert dgamma.numel() == ng * co == dbeta.numel()
        if d_row_bias is not None:
            _check(d_row_bias); _f32(d_row_bias)
            assert group in (16, 64) and d_row_bias.numel() == nb * co * (p // group)
        if gamma is not None:
            _check(gamma); _f32(gamma)
        if x_coef is not None:
            _check(x_coef); _f32(x_coef)
            assert tuple(x_coef.shape) == (ng * ci, 4)
        lib = _lib.load()
        need = lib.nesie_pw_wgrad_workspace_bytes(nb, ng, co, ci, p)
        defer = final and HipKernels._deferred is not None
        with torch.cuda.device(da.device):
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=da.device)
            cws = torch.empty(ng * co, 8, dtype=torch.float32, device=da.device)
            if defer:
                HipKernels._deferred.append((ws, dw))
            _lib.call("nesie_pw_wgrad_bn_backward_deferred" if defer else "nesie_pw_wgrad_bn_backward",
                      nb, ng, co, ci, p, _ptr(da), _ptr(z), co * p,
                      _ptr(z_coef), 0 if gamma is None else _ptr(gamma), _ptr(part), part.shape[1],
                      _ptr(x), x.stride(0) if nb > 1 else ci * p, 0 if x_coef is None else _ptr(x_coef),
                      int(bool(x_relu)), _ptr(dz), _ptr(dw), _ptr(dgamma), _ptr(dbeta), _ptr(cws),
                      0 if d_row_bias is None else _ptr(d_row_bias), int(group), _ptr(ws), need,
                      _stream(da))

    @staticmethod
    def conv_wgrad_supported(cout, cin):
        return (cout <= 128 and cin <= 288) or (cout <= 256 and cin <= 128)

    def bn_relu_forward(self, x, gamma, beta, running_mean, running_var, momentum, eps, relu,
                        y, save_mean, save_invstd, fwd_coef, row_bias=None, pre_partial=None):
        """x, y (B, C, *) fp32; per-channel vectors [C]; running stats updated in place.
        row_bias (B, C, K): added to x broadcast over the last axis of x (B, C, K, G)."""
        _check(x, y, save_mean, save_invstd); _f32(x, y, save_mean, save_invstd)
        b, c = x.shape[:2]
        p = x.numel() // (b * c) if b * c else 0
        group = _row_bias_group(x, row_bias)
        need = _lib.load().nesie_bn_workspace_bytes(b, c, p)
        with torch.cuda.device(x.device):
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
            opt = lambda t: 0 if t is None else _ptr(t)  # noqa: E731
            _lib.call("nesie_bn_relu_forward", b, c, p, _ptr(x), opt(gamma), opt(beta),
                      opt(running_mean), opt(running_var), float(momentum), float(eps),
                      int(bool(relu)), _ptr(y), _ptr(save_mean), _ptr(save_invstd),
                      _ptr(fwd_coef), opt(row_bias), group, opt(pre_partial),
                      0 if pre_partial is None else pre_partial.numel() // (2 * c),
                      _ptr(ws), need, _stream(x))

    def pw_supported(self, k, cout, p):
        return bool(_lib.load().nesie_pw_supported(int(k), int(cout), int(p)))

    def pw_stat_slots(self, nb, ng, k, cout, p):
        return int(_lib.load().nesie_pw_stat_slots(nb, ng, k, cout, p))

    def pw_layer_forward(self, x, w, ng=1, in_coef=None, in_relu=True, row_bias=None, rb_group=0,
                         bias=None, y=None, stat_part=None, pool_group=0, pool_min=False,
                         pool_out=None):
        """y[n] = W[n % ng] . act(x[n]) (+ row_bias) (+ bias) on the matrix cores
        (nesie_pw_layer_forward).  x (NB, K, P) with each x[n] contiguous (batch stride free);
        w (ng, Cout, K) as ANY strided view (a transposed view gives the input-gradient product);
        in_coef (ng*K, 4) folded BatchNorm of the operand; y (NB, Cout, P) or None;
        stat_part (ng, slots, Cout, 4); pool_out = (max, min | None, argmax, argmin | None), each
        (NB, Cout, P / pool_group)."""
        _f32(x, w)
        nb, k, p = x.shape
        assert w.dim() == 3 and w.shape[0] == ng and w.shape[2] == k and nb % ng == 0
        cout = w.shape[1]
        assert x.is_cuda and x.stride(2) == 1 and x.stride(1) == p, "each x[n] must be (K, P) contiguous"
        opt = lambda t: 0 if t is None else _ptr(t)  # noqa: E731
        if y is not None:
            _f32(y)
            assert y.is_cuda and tuple(y.shape) == (nb, cout, p) and y.stride(2) == 1 \
                and y.stride(1) == p, "each y[n] must be (Cout, P) contiguous"
        if stat_part is not None:
            _check(stat_part); _f32(stat_part)
            assert tuple(stat_part.shape) == (ng, self.pw_stat_slots(nb, ng, k, cout, p), cout, 4)
        if in_coef is not None:
            _check(in_coef); _f32(in_coef)
            assert tuple(in_coef.shape) == (ng * k, 4)
        if row_bias is not None:
            _check(row_bias); _f32(row_bias)
            assert row_bias.numel() == nb * cout * (p // rb_group)
        if bias is not None:
            _check(bias); _f32(bias)
            assert bias.numel() == ng * cout
        pmax = pmin = amax = amin = None
        if pool_group:
            pmax, pmin, amax, amin = pool_out
            for t in (pmax, pmin):
                if t is not None:
                    _check(t); _f32(t)
                    assert t.numel() == nb * cout * (p // pool_group)
            for t in (amax, amin):
                if t is not None:
                    _check(t)
                    assert t.dtype == torch.uint8 and t.numel() == nb * cout * (p // pool_group)
            assert (pmin is not None) == bool(pool_min)
        with torch.cuda.device(x.device):
            _lib.call("nesie_pw_layer_forward", nb, ng, k, cout, p, _ptr(x),
                      x.stride(0) if nb > 1 else k * p, _ptr(w), w.stride(0) if ng > 1 else 0,
                      w.stride(1), w.stride(2), opt(in_coef), int(bool(in_relu)), opt(row_bias),
                      int(rb_group), opt(bias), opt(y),
                      y.stride(0) if (y is not None and nb > 1) else cout * p, opt(stat_part),
                      int(pool_group),
                      int(bool(pool_min)), opt(pmax), opt(pmin), opt(amax), opt(amin), _stream(x))

    # ---- SA1's first layer without its output tensor (include/nesie_ops.h, "round 5: the first
    # shared-MLP layer ... WITHOUT its output tensor")
    @staticmethod
    def _k4_check(x4, w0):
        _f32(x4, w0); _check(w0)
        nb, c, p = x4.shape
        assert x4.is_cuda and c == 4 and x4.stride(2) == 1 and x4.stride(1) == p and tuple(w0.shape) == (64, 4)
        return nb, p

    def k4_supported(self, c0, c1, c2, p):
        """SA1's shape: 4 -> 64 -> 64 over whole 64-position tiles."""
        return c0 == 4 and c1 == 64 and c2 == 64 and p % 64 == 0 and self.pw_supported(64, 64, p) \
            and self.pw_wgrad_bn_supported(64, 64, p)

    def pw_layer_forward_k4(self, x4, w0, w, in_coef, y, stat_part):
        """y[n] = W . relu(bn(W0 . x4[n])) with the inner 64-row tensor rebuilt in the staging
        (nesie_pw_layer_forward_k4): x4 (NB, 4, P), w0 (64, 4), w (Cout = 64, 64) any strided view,
        in_coef (64, 4) the first layer's folded norm, y (NB, 64, P), stat_part as pw_layer_forward."""
        nb, p = self._k4_check(x4, w0)
        _f32(w, y); _check(in_coef, y, stat_part); _f32(in_coef, stat_part)
        cout = w.shape[0]
        assert tuple(w.shape) == (cout, 64) and tuple(y.shape) == (nb, cout, p) and tuple(in_coef.shape) == (64, 4)
        assert tuple(stat_part.shape) == (1, self.pw_stat_slots(nb, 1, 64, cout, p), cout, 4)
        with torch.cuda.device(x4.device):
            _lib.call("nesie_pw_layer_forward_k4", nb, cout, p, _ptr(x4), x4.stride(0) if nb > 1 else 4 * p,
                      _ptr(w0), _ptr(w), w.stride(0), w.stride(1), _ptr(in_coef), _ptr(y), cout * p,
                      _ptr(stat_part), _stream(x4))

    def pw_dgrad_bn_reduce_k4(self, dy, w, x4, w0, z_coef):
        """The reductions of ``pw_dgrad_bn_reduce`` for a layer whose input was relu(bn(W0 . x4)),
        WITHOUT the input gradient itself (nesie_pw_dgrad_bn_reduce_k4): dy (NB, K, P), w the
        transposed weight view (64, K) -> (part (64, slots, 2), g_part (64, slots, 4))."""
        nb, p = self._k4_check(x4, w0)
        _f32(dy, w); _check(z_coef); _f32(z_coef)
        k = dy.shape[1]
        assert tuple(dy.shape) == (nb, k, p) and dy.is_cuda and dy.stride(2) == 1 and dy.stride(1) == p
        assert tuple(w.shape) == (64, k) and tuple(z_coef.shape) == (64, 4)
        slots = self.pw_stat_slots(nb, 1, k, 64, p)
        part = torch.empty(64, slots, 2, dtype=torch.float32, device=dy.device)
        g_part = torch.empty(64, slots, 4, dtype=torch.float32, device=dy.device)
        with torch.cuda.device(dy.device):
            _lib.call("nesie_pw_dgrad_bn_reduce_k4", nb, k, p, _ptr(dy), dy.stride(0) if nb > 1 else k * p,
                      _ptr(w), w.stride(0), w.stride(1), _ptr(x4), x4.stride(0) if nb > 1 else 4 * p,
                      _ptr(w0), _ptr(z_coef), _ptr(part), _ptr(g_part), _stream(dy))
        return part, g_part

    def pw_wgrad_bn_backward_k4(self, da, z, z_coef, gamma, part, x4, w0, x_coef, dw, dgamma, dbeta,
                                final=False):
        """``pw_wgrad_bn_backward`` of a 64 x 64 layer whose X operand is relu(bn(W0 . x4)), rebuilt on
        the load (nesie_pw_wgrad_bn_backward_k4); dz is written over da."""
        nb, p = self._k4_check(x4, w0)
        _f32(da, z, dw, dgamma, dbeta); _check(da, z, z_coef, part, x_coef, dw); _f32(z_coef, part, x_coef)
        assert tuple(da.shape) == (nb, 64, p) == tuple(z.shape) and dw.numel() == 64 * 64
        assert tuple(z_coef.shape) == (64, 4) == tuple(x_coef.shape) and part.shape[0] == 64 and part.shape[2] == 2
        if gamma is not None:
            _check(gamma); _f32(gamma)
        need = _lib.load().nesie_pw_wgrad_workspace_bytes(nb, 1, 64, 64, p)
        defer = final and HipKernels._deferred is not None
        with torch.cuda.device(da.device):
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=da.device)
            cws = torch.empty(64, 8, dtype=torch.float32, device=da.device)
            if defer:
                HipKernels._deferred.append((ws, dw))
            _lib.call("nesie_pw_wgrad_bn_backward_k4", nb, p, _ptr(da), _ptr(z), 64 * p, _ptr(z_coef),
                      0 if gamma is None else _ptr(gamma), _ptr(part), part.shape[1], _ptr(x4),
                      x4.stride(0) if nb > 1 else 4 * p, _ptr(w0), _ptr(x_coef), _ptr(da), _ptr(dw),
                      _ptr(dgamma), _ptr(dbeta), _ptr(cws), _ptr(ws), need, int(defer), _stream(da))

    def pw_wgrad_bn_backward_k4_fused(self, da, z, z_coef, gamma, part, x4, w0, x_coef, w, dw, dgamma, dbeta,
                                      final=False):
        """``pw_wgrad_bn_backward_k4`` + ``pw_dgrad_bn_reduce_k4`` as one launch
        (nesie_pw_wgrad_bn_backward_k4_fused): w (64, 64) the layer's weight; da is NOT overwritten (dz is
        consumed inside the launch).  -> (in_part (64, slots, 2), in_gpart (64, slots, 4))."""
        nb, p = self._k4_check(x4, w0)
        _f32(da, z, dw, dgamma, dbeta, w); _check(da, z, z_coef, part, x_coef, dw, w); _f32(z_coef, part, x_coef)
        assert tuple(da.shape) == (nb, 64, p) == tuple(z.shape) and dw.numel() == 64 * 64 and tuple(w.shape) == (64, 64)
        assert tuple(z_coef.shape) == (64, 4) == tuple(x_coef.shape) and part.shape[0] == 64 and part.shape[2] == 2
        if gamma is not None:
            _check(gamma); _f32(gamma)
        lib = _lib.load()
        need = lib.nesie_pw_wgrad_workspace_bytes(nb, 1, 64, 64, p)
        slots = lib.nesie_pw_wgrad_bn_backward_k4_slots(nb, p)
        defer = final and HipKernels._deferred is not None
        with torch.cuda.device(da.device):
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=da.device)
            cws = torch.empty(64, 8, dtype=torch.float32, device=da.device)
            in_part = torch.empty(64, slots, 2, dtype=torch.float32, device=da.device)
            in_gpart = torch.empty(64, slots, 4, dtype=torch.float32, device=da.device)
            if defer:
                HipKernels._deferred.append((ws, dw))
            _lib.call("nesie_pw_wgrad_bn_backward_k4_fused", nb, p, _ptr(da), _ptr(z), 64 * p, _ptr(z_coef),
                      0 if gamma is None else _ptr(gamma), _ptr(part), part.shape[1], _ptr(x4),
                      x4.stride(0) if nb > 1 else 4 * p, _ptr(w0), _ptr(x_coef), _ptr(w), _ptr(dw),
                      _ptr(dgamma), _ptr(dbeta), _ptr(cws), _ptr(in_part), _ptr(in_gpart), _ptr(ws), need,
                      int(defer), _stream(da))
        return in_part, in_gpart

    def k4_moments(self, x4):
        """-> (256, 20) float64 per-workgroup partial sums of X4[j] and X4[j] X4[k] (nesie_k4_moments)."""
        _f32(x4)
        nb, c, p = x4.shape
        assert x4.is_cuda and c == 4 and x4.stride(2) == 1 and x4.stride(1) == p
        need = _lib.load().nesie_k4_moments_bytes()
        with torch.cuda.device(x4.device):
            mom = torch.empty(need // 160, 20, dtype=torch.float64, device=x4.device)
            _lib.call("nesie_k4_moments", nb, p, _ptr(x4), x4.stride(0) if nb > 1 else 4 * p, _ptr(mom), _stream(x4))
        return mom

    def k4_stat_finalize(self, mom, w0, gamma, beta, running_mean, running_var, momentum, eps, count, coef):
        """Training-mode BatchNorm coefficients of W0 . X4 from the moments of X4 (nesie_k4_stat_finalize)."""
        _check(mom, w0, coef); _f32(w0, coef)
        assert mom.dtype == torch.float64 and tuple(w0.shape) == (64, 4) and tuple(coef.shape) == (64, 4)
        opt = lambda t: 0 if t is None else _ptr(t)  # noqa: E731
        with torch.cuda.device(mom.device):
            _lib.call("nesie_k4_stat_finalize", float(count), _ptr(mom), _ptr(w0), opt(gamma), opt(beta),
                      opt(running_mean), opt(running_var), float(momentum), float(eps), _ptr(coef), _stream(mom))

    def k4_first_layer_wgrad(self, mom, w0, bnb, g_part, dw):
        """dW0 (64, 4) from the reductions (nesie_k4_first_layer_wgrad): mom from ``k4_moments``, bnb (64, 8)
        from ``pw_bnb_coef`` over the part of ``pw_dgrad_bn_reduce_k4``, g_part (64, slots, 4) of the same."""
        _check(mom, w0, bnb, g_part, dw); _f32(w0, bnb, g_part, dw)
        assert mom.dtype == torch.float64 and tuple(w0.shape) == (64, 4)
        assert tuple(bnb.shape) == (64, 8) and g_part.shape[0] == 64 and g_part.shape[2] == 4 and dw.numel() == 256
        with torch.cuda.device(mom.device):
            _lib.call("nesie_k4_first_layer_wgrad", _ptr(mom), _ptr(w0), _ptr(bnb), _ptr(g_part), g_part.shape[1],
                      _ptr(dw), _stream(mom))

    def pw_dgrad_bn_reduce(self, dy, w, z, z_coef, da, ng=1):
        """da[n] = W[n % ng] . dy[n] (w = the transposed weight view (ng, Cin, Cout)) plus the
        reduction of the backward of relu(bn(z)) that consumes da (nesie_pw_dgrad_bn_reduce):
        -> partials (ng*Cin, slots, 2) for ``bn_relu_backward_apply``.  z (NB, Cin, P) raw conv
        output, z_coef (ng*Cin, 4) its folded BatchNorm."""
        _f32(dy, w, z, da); _check(z_coef); _f32(z_coef)
        nb, k, p = dy.shape
        cout = w.shape[1]
        assert w.dim() == 3 and w.shape[0] == ng and w.shape[2] == k and nb % ng == 0
        for t in (dy, z, da):
            assert t.is_cuda and t.stride(2) == 1 and t.stride(1) == p
        assert tuple(z.shape) == (nb, cout, p) == tuple(da.shape) and tuple(z_coef.shape) == (ng * cout, 4)
        part = torch.empty(ng * cout, self.pw_stat_slots(nb, ng, k, cout, p), 2,
                           dtype=torch.float32, device=dy.device)
        bs = lambda t, rows: t.stride(0) if nb > 1 else rows * p  # noqa: E731
        with torch.cuda.device(dy.device):
            _lib.call("nesie_pw_dgrad_bn_reduce", nb, ng, k, cout, p, _ptr(dy), bs(dy, k), _ptr(w),
                      w.stride(0) if ng > 1 else 0, w.stride(1), w.stride(2), _ptr(da),
                      bs(da, cout), _ptr(z), bs(z, cout), _ptr(z_coef), _ptr(part), _stream(dy))
        return part

    def bn_relu_backward_apply(self, dy, x, gamma, save_invstd, fwd_coef, partial, dx, dgamma,
                               dbeta, d_row_bias=None, group=None):
        """Apply pass of the BatchNorm + ReLU backward from ready partials (C, nslice, 2); x is
        the raw conv output of the fused forward (nesie_bn_relu_backward_apply).  save_invstd None:
        column 3 of fwd_coef."""
        _check(dy, x, dx, fwd_coef, partial); _f32(dy, x, dx, partial)
        b, c = x.shape[:2]
        p = x.numel() // (b * c) if b * c else 0
        assert partial.dim() == 3 and partial.shape[0] == c and partial.shape[2] == 2
        if d_row_bias is not None:
            _check(d_row_bias); _f32(d_row_bias)
            assert group and d_row_bias.numel() == b * c * (p // group)
        opt = lambda t: 0 if t is None else _ptr(t)  # noqa: E731
        with torch.cuda.device(x.device):
            _lib.call("nesie_bn_relu_backward_apply", b, c, p, _ptr(dy), _ptr(x), opt(gamma),
                      opt(save_invstd), _ptr(fwd_coef), _ptr(partial), int(partial.shape[1]),
                      _ptr(dx), opt(dgamma), opt(dbeta), int(group or 1), opt(d_row_bias),
                      _stream(x))

    def vote_loss_forward(self, seed, vote, seed_idx, mask, targets, w_dst, ticket):
        """-> (loss scalar, sign (B,N,3), scale scalar) (nesie_vote_loss_forward)."""
        _check(seed, vote, seed_idx, mask, targets); _f32(seed, vote, targets)
        assert seed_idx.dtype == torch.int64 and mask.dtype == torch.int64
        b, n = seed.shape[:2]
        npts = mask.shape[1]
        gps = targets.shape[2] // 3
        assert tuple(vote.shape) == (b, n, 3) and tuple(targets.shape) == (b, npts, 3 * gps)
        dev = seed.device
        sign = torch.empty(b, n, 3, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        scale = torch.empty((), dtype=torch.float32, device=dev)
        partial = torch.empty(64, 2, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.call("nesie_vote_loss_forward", b, n, npts, gps, _ptr(seed), _ptr(vote),
                      _ptr(seed_idx), _ptr(mask), _ptr(targets), float(w_dst), _ptr(sign),
                      _ptr(loss), _ptr(scale), _ptr(partial), _ptr(ticket),
                      _stream(seed))
        return loss, sign, scale

    def vote_loss_backward(self, g, scale, sign):
        _check(g, scale, sign); _f32(g, scale, sign)
        d = torch.empty_like(sign)
        with torch.cuda.device(sign.device):
            _lib.call("nesie_vote_loss_backward", sign.numel(), _ptr(g), _ptr(scale), _ptr(sign),
                      _ptr(d), _stream(sign))
        return d

    def proposal_jitter(self, bbox, noise_c, noise_s, sigma, size_bias, zero_heading):
        """-> (centre_all (B,2K,3), size_all (B,2K,3), heading_all (B,2K), jitter_bbox (B,K,7))
        (nesie_proposal_jitter)."""
        _check(bbox, noise_c, noise_s); _f32(bbox, noise_c, noise_s)
        b, k, _ = bbox.shape
        dev = bbox.device
        f32 = lambda *s_: torch.empty(*s_, dtype=torch.float32, device=dev)  # noqa: E731
        ca, sa, ha, jb = f32(b, 2 * k, 3), f32(b, 2 * k, 3), f32(b, 2 * k), f32(b, k, 7)
        with torch.cuda.device(dev):
            _lib.call("nesie_proposal_jitter", b, k, _ptr(bbox), _ptr(noise_c), _ptr(noise_s),
                      float(sigma), float(size_bias), int(bool(zero_heading)), _ptr(ca), _ptr(sa),
                      _ptr(ha), _ptr(jb), _stream(bbox))
        return ca, sa, ha, jb

    def side_prob_stats(self, probs, copies):
        """probs (B, 6, bins, K) -> (6, B, bins + 5, copies*K): bins, top-4, unbiased variance per
        face (nesie_side_prob_stats)."""
        _check(probs); _f32(probs)
        b, six, bins, k = probs.shape
        assert six == 6 and bins >= 5
        out = torch.empty(6, b, bins + 5, copies * k, dtype=torch.float32, device=probs.device)
        with torch.cuda.device(probs.device):
            _lib.call("nesie_side_prob_stats", b, bins, k, int(copies), _ptr(probs), _ptr(out),
                      _stream(probs))
        return out

    def flat_adamw_step(self, param, grad, exp_avg, exp_avg_sq, step, lr, betas, eps, weight_decay,
                        max_norm, grad_norm_out=None, hyper=None):
        """clip_grad_norm_(max_norm) + AdamW over flat vectors, in place (nesie_flat_adamw_step);
        ``step`` is a float32 device scalar that the call increments.  ``hyper`` (2,) device
        float32 = [lr, weight_decay]: read at execution time instead of the host values
        (nesie_flat_adamw_step_dev: a captured launch follows the schedule)."""
        _check(param, grad, exp_avg, exp_avg_sq, step); _f32(param, grad, exp_avg, exp_avg_sq, step)
        n = param.numel()
        assert grad.numel() == n == exp_avg.numel() == exp_avg_sq.numel() and step.numel() == 1
        need = _lib.load().nesie_flat_adamw_workspace_bytes()
        ws = torch.empty(need, dtype=torch.uint8, device=param.device)
        if hyper is not None:
            _check(hyper); _f32(hyper)
            assert hyper.numel() == 2 and hyper.is_contiguous()
            with torch.cuda.device(param.device):
                _lib.call("nesie_flat_adamw_step_dev", n, _ptr(param), _ptr(grad), _ptr(exp_avg),
                          _ptr(exp_avg_sq), _ptr(step), _ptr(hyper), float(betas[0]),
                          float(betas[1]), float(eps), float(max_norm or 0.0),
                          0 if grad_norm_out is None else _ptr(grad_norm_out), _ptr(ws), need,
                          _stream(param))
            return
        with torch.cuda.device(param.device):
            _lib.call("nesie_flat_adamw_step", n, _ptr(param), _ptr(grad), _ptr(exp_avg),
                      _ptr(exp_avg_sq), _ptr(step), float(lr), float(betas[0]), float(betas[1]),
                      float(eps), float(weight_decay), float(max_norm or 0.0),
                      0 if grad_norm_out is None else _ptr(grad_norm_out), _ptr(ws), need,
                      _stream(param))

    # ---- Nesie head: targets and loss terms (include/nesie_head_ops.h) -----------------------
    def head_targets(self, agg, gt_boxes, gt_labels, gt_count, gt_valid, pos_thr, neg_thr):
        """-> dict(assignment, obj_targets, obj_weights, mask_targets, bbox_targets,
        center_targets, box_weights, valid_weights) (nesie_head_targets)."""
        _check(agg, gt_boxes, gt_labels, gt_count, gt_valid); _f32(agg, gt_boxes, gt_valid)
        assert gt_labels.dtype == torch.int64 and gt_count.dtype == torch.int64
        b, k = agg.shape[:2]
        t = gt_boxes.shape[1]
        dev = agg.device
        i64 = lambda *s_: torch.empty(*s_, dtype=torch.int64, device=dev)  # noqa: E731
        f32 = lambda *s_: torch.empty(*s_, dtype=torch.float32, device=dev)  # noqa: E731
        out = dict(assignment=i64(b, k), obj_targets=i64(b, k), obj_weights=f32(b, k),
                   mask_targets=i64(b, k), bbox_targets=f32(b, k, 7), center_targets=f32(b, t, 3),
                   box_weights=f32(b, k), valid_weights=f32(b, t))
        with torch.cuda.device(dev):
            _lib.call("nesie_head_targets", b, k, t, _ptr(agg), _ptr(gt_boxes), _ptr(gt_labels),
                      _ptr(gt_count), _ptr(gt_valid), float(pos_thr), float(neg_thr),
                      _ptr(out['assignment']), _ptr(out['obj_targets']), _ptr(out['obj_weights']),
                      _ptr(out['mask_targets']), _ptr(out['bbox_targets']),
                      _ptr(out['center_targets']), _ptr(out['box_weights']),
                      _ptr(out['valid_weights']), _stream(agg))
        return out

    def head_loss_forward(self, cls, bbox, surface, side, iou_s, iou, iou_j, tg, config, ticket,
                          quality=None, detach_sigma=False, sigma_mode=0):
        """The seven loss terms + their saved gradients (nesie_head_loss_forward).  cls
        (B,2+C,K), bbox (B,K,7), surface (B,K,6), side (6,B,C,2K), iou_s (B,2K,C), iou / iou_j
        (B*K); tg = head_targets(...); config = 11 python floats; ticket = a zeroed int32 scalar
        that lives outside any graph capture (the kernel leaves it zero).  -> (loss (7,), saved dict).
        quality (B*K, 6): the unsupervised variant (nesie_head_loss_forward_unsup; iou_j unused)."""
        _check(cls, bbox, surface, side, iou_s, iou, iou_j)
        _f32(cls, bbox, surface, side, iou_s, iou, iou_j)
        b, nc, k = cls.shape
        c = nc - 2
        t = tg['center_targets'].shape[1]
        assert tuple(side.shape) == (6, b, c, 2 * k) and tuple(bbox.shape) == (b, k, 7)
        assert tuple(iou_s.shape) == (b, 2 * k, c) and tuple(surface.shape) == (b, k, 6)
        assert iou.numel() == b * k == iou_j.numel() and len(config) == 11
        dev = cls.device
        f32 = lambda *s_: torch.empty(*s_, dtype=torch.float32, device=dev)  # noqa: E731
        i32 = lambda *s_: torch.empty(*s_, dtype=torch.int32, device=dev)  # noqa: E731
        sv = dict(cls=f32(b, nc, k), centre=f32(b * k, 3), surface=f32(b, k, 6), iou=f32(b * k),
                  iou_s=f32(b, 2 * k, c), side_surf=f32(b * k, 6), side_iou=f32(b * k, 6),
                  side_pred=f32(b * k, 6), sem_pick=i32(b * k))
        loss, kstar, dmin = f32(7), i32(b * t), f32(b * t)
        partial = f32((b * k + 63) // 64, 8)
        assert ticket.dtype == torch.int32 and ticket.is_cuda and ticket.numel() == 1
        cfg = (ctypes.c_float * 11)(*[float(v) for v in config])
        if quality is not None:
            _check(quality); _f32(quality)
            assert quality.numel() == b * k * 6
            head = ("nesie_head_loss_forward_unsup", b, k, t, c, _ptr(cls), _ptr(bbox), _ptr(surface),
                    _ptr(side), _ptr(iou_s), _ptr(iou), _ptr(quality), int(bool(detach_sigma)))
        elif sigma_mode:      # the SAQE head's supervised losses: constant (1) or no (2) uncertainties
            head = ("nesie_head_loss_forward_sigma", int(sigma_mode), b, k, t, c, _ptr(cls), _ptr(bbox),
                    _ptr(surface), _ptr(side), _ptr(iou_s), _ptr(iou), _ptr(iou_j))
        else:
            head = ("nesie_head_loss_forward", b, k, t, c, _ptr(cls), _ptr(bbox), _ptr(surface),
                    _ptr(side), _ptr(iou_s), _ptr(iou), _ptr(iou_j))
        with torch.cuda.device(dev):
            _lib.call(*head, _ptr(tg['obj_targets']),
                      _ptr(tg['mask_targets']), _ptr(tg['obj_weights']), _ptr(tg['box_weights']),
                      _ptr(tg['bbox_targets']), _ptr(tg['center_targets']),
                      _ptr(tg['valid_weights']), ctypes.cast(cfg, ctypes.c_void_p), _ptr(loss),
                      _ptr(sv['cls']), _ptr(sv['centre']), _ptr(sv['surface']), _ptr(sv['iou']),
                      _ptr(sv['iou_s']), _ptr(sv['side_surf']), _ptr(sv['side_iou']),
                      _ptr(sv['side_pred']), _ptr(sv['sem_pick']), _ptr(kstar), _ptr(dmin),
                      _ptr(partial), _ptr(ticket), _stream(cls))
        return loss, sv

    def saqe_extra_forward(self, robj, rot, bbox, bbox_t, jsurf, side, tg, sem_pick, config, sup,
                           ticket):
        """The SAQE head's additional supervised terms (nesie_saqe_extra_loss_forward): robj
        (B,2K,2), rot (B,2K,C), bbox / bbox_t (B,K,7), jsurf (B,K,6), side (6,B,C,2K); sem_pick from
        ``head_loss_forward``; config = 7 python floats.  -> (loss (4,), saved dict)."""
        _check(robj, rot, bbox, bbox_t, jsurf, side, sem_pick); _f32(robj, rot, bbox, bbox_t, jsurf, side)
        b, k2, c = rot.shape
        k = k2 // 2
        assert tuple(robj.shape) == (b, k2, 2) and tuple(side.shape) == (6, b, c, k2) and len(config) == 7
        assert bbox.numel() == b * k * 7 == bbox_t.numel() and jsurf.numel() == b * k * 6
        dev = rot.device
        f32 = lambda *s_: torch.empty(*s_, dtype=torch.float32, device=dev)  # noqa: E731
        sv = dict(robj=f32(b, k2, 2), angle=f32(b * k), rot=torch.zeros(b, k2, c, dtype=torch.float32, device=dev),
                  sidej=f32(b * k, 6))
        loss, partial = f32(4), f32((b * k + 63) // 64, 4)
        wmax = tg['box_weights'].max().reshape(1)
        cfg = (ctypes.c_float * 7)(*[float(v) for v in config])
        with torch.cuda.device(dev):
            _lib.call("nesie_saqe_extra_loss_forward", b, k, c, int(bool(sup)), _ptr(robj), _ptr(rot),
                      _ptr(bbox), _ptr(bbox_t), _ptr(jsurf), _ptr(side), _ptr(tg['obj_targets']),
                      _ptr(tg['mask_targets']), _ptr(tg['obj_weights']), _ptr(tg['box_weights']),
                      _ptr(wmax), _ptr(sem_pick), ctypes.cast(cfg, ctypes.c_void_p), _ptr(loss),
                      _ptr(sv['robj']), _ptr(sv['angle']), _ptr(sv['rot']), _ptr(sv['sidej']),
                      _ptr(partial), _ptr(ticket), _stream(rot))
        return loss, sv

    def saqe_extra_backward(self, g, label, sv, k):
        """g (4,) -> dict(robj, angle (B*K), rot, side (6,B,C,2K))."""
        _check(g, label); _f32(g)
        b, k2, c = sv['rot'].shape
        dev = g.device
        out = dict(robj=torch.empty_like(sv['robj']), angle=torch.empty_like(sv['angle']),
                   rot=torch.empty_like(sv['rot']),
                   side=torch.zeros(6, b, c, k2, dtype=torch.float32, device=dev))
        with torch.cuda.device(dev):
            _lib.call("nesie_saqe_extra_loss_backward", b, k, c, _ptr(g), _ptr(label), _ptr(sv['robj']),
                      _ptr(sv['angle']), _ptr(sv['rot']), _ptr(sv['sidej']), _ptr(out['robj']),
                      _ptr(out['angle']), _ptr(out['rot']), _ptr(out['side']), _stream(g))
        return out

    def head_loss_backward(self, g, label, sv, k):
        """g (7,) incoming gradients (device) -> dict(cls, bbox, surface, iou, iou_s, side) in the
        producers' layouts (nesie_head_loss_backward)."""
        _check(g, label); _f32(g)
        b, nc, _ = sv['cls'].shape
        c = nc - 2
        dev = g.device
        out = dict(cls=torch.empty_like(sv['cls']),
                   bbox=torch.empty(b, k, 7, dtype=torch.float32, device=dev),
                   surface=torch.empty_like(sv['surface']), iou=torch.empty_like(sv['iou']),
                   iou_s=torch.empty_like(sv['iou_s']),
                   side=torch.zeros(6, b, c, 2 * k, dtype=torch.float32, device=dev))
        with torch.cuda.device(dev):
            _lib.call("nesie_head_loss_backward", b, k, c, _ptr(g), _ptr(label), _ptr(sv['sem_pick']),
                      _ptr(sv['cls']), _ptr(sv['centre']), _ptr(sv['surface']), _ptr(sv['iou']),
                      _ptr(sv['iou_s']), _ptr(sv['side_surf']), _ptr(sv['side_iou']),
                      _ptr(sv['side_pred']), _ptr(out['cls']), _ptr(out['bbox']),
                      _ptr(out['surface']), _ptr(out['iou']), _ptr(out['iou_s']), _ptr(out['side']),
                      _stream(g))
        return out

    def pw_stats_finalize(self, stat_part, gamma, beta, running_mean, running_var, momentum, eps,
                          coef, chan_bias=None):
        """(ng, slots, Cout, 4) shifted partials -> coef (ng*Cout, 4) = (scale, bias, mean,
        invstd); running statistics (ng*Cout) updated in place (or None).  chan_bias (ng*Cout):
        bias of the convolution in front of the norm, added to the running mean only."""
        _check(stat_part, coef); _f32(stat_part, coef)
        ng, nslots, cout, _ = stat_part.shape
        assert tuple(coef.shape) == (ng * cout, 4)
        assert chan_bias is None or (chan_bias.numel() == ng * cout and chan_bias.is_contiguous())
        opt = lambda t: 0 if t is None else _ptr(t)  # noqa: E731
        with torch.cuda.device(coef.device):
            _lib.call("nesie_pw_stats_finalize", ng * cout, cout, nslots, _ptr(stat_part), opt(gamma),
                      opt(beta), opt(running_mean), opt(running_var), float(momentum), float(eps),
                      _ptr(coef), opt(chan_bias), _stream(coef))

    def pw_pool_finish(self, ng, p, group, pool_group, pool_out, coef, relu, pooled, argmax, zstar=None):
        """partial extrema (NB, C, P / pool_group) -> pooled (NB, C, P / group) float,
        argmax uint8 (position inside the group); zstar (optional, like pooled): the raw extremum."""
        pmax, pmin, amax, amin = pool_out
        _check(pmax, amax, pooled, argmax); _f32(pmax, pooled)
        nb, c = pooled.shape[:2]
        assert pooled.numel() == nb * c * (p // group) and argmax.dtype == torch.uint8
        opt = lambda t: 0 if t is None else _ptr(t)  # noqa: E731
        with torch.cuda.device(pooled.device):
            if zstar is not None:
                _check(zstar); _f32(zstar)
                assert zstar.numel() == pooled.numel()
                _lib.call("nesie_pw_pool_finish_z", nb, ng, c, p, group, pool_group, _ptr(pmax), opt(pmin),
                          _ptr(amax), opt(amin), opt(coef), int(bool(relu)), _ptr(pooled),
                          _ptr(argmax), _ptr(zstar), _stream(pooled))
                return
            _lib.call("nesie_pw_pool_finish", nb, ng, c, p, group, pool_group, _ptr(pmax), opt(pmin),
                      _ptr(amax), opt(amin), opt(coef), int(bool(relu)), _ptr(pooled),
                      _ptr(argmax), _stream(pooled))

    # ---- pooled tail without the dense pre-pool tensor (csrc/pool_tail.hip) ---------------------
    def pool_tail_supported(self, k, c, p, ns):
        return bool(_lib.load().nesie_pool_tail_supported(int(k), int(c), int(p), int(ns)))

    def pool_tail_backward(self, g, pooled, zstar, argmax, coef, gamma, w, z_prev, coef_prev, ns,
                           dgamma, dbeta, dw=None):
        """Backward of conv (k -> c) + BatchNorm + ReLU + max over groups of ``ns`` positions from
        the pooled gradient g (NB, C, M), what the forward kept (pooled, zstar, argmax (NB, C, M),
        coef (C, 4)) and the layer's operand in raw form (z_prev (NB, K, P) with its folded
        coefficients coef_prev (K, 4)).  Writes dgamma, dbeta (C,) and dw (C, K) (None: skipped);
        returns (da (NB, K, P), partials of the previous norm's backward (K, slots, 2))."""
        _check(g, pooled, zstar, argmax, coef, gamma, w, z_prev, coef_prev, dgamma, dbeta)
        _f32(g, pooled, zstar, coef, gamma, w, z_prev, coef_prev, dgamma, dbeta)
        nb, c, m = g.shape
        k, p = z_prev.shape[1], z_prev.shape[2]
        assert tuple(z_prev.shape) == (nb, k, p) and p == m * ns and tuple(w.shape) == (c, k)
        assert argmax.dtype == torch.uint8 and tuple(coef.shape) == (c, 4) and tuple(coef_prev.shape) == (k, 4)
        assert z_prev.stride(2) == 1 and z_prev.stride(1) == p
        dev = g.device
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
        sizes = (ctypes.c_int * 2)()
        _lib.call("nesie_pool_tail_sizes", nb, k, c, p, ctypes.addressof(sizes))
        nslots, nparts = int(sizes[0]), int(sizes[1])
        ab, ent, wcat, c0 = f(c, 4), f(nb, m, c, 2), f(k, c + k), f(k)
        da, part = f(nb, k, p), f(k, nslots, 2)
        zbs = z_prev.stride(0) if nb > 1 else k * p
        with torch.cuda.device(dev):
            st = _stream(g)
            _lib.call("nesie_pool_tail_prepare", nb, c, m, ns, k, _ptr(g), _ptr(pooled), _ptr(zstar),
                      _ptr(argmax), _ptr(coef), _ptr(gamma), _ptr(w), _ptr(dgamma), _ptr(dbeta),
                      _ptr(ab), _ptr(ent), _ptr(wcat), _ptr(c0), st)
            _lib.call("nesie_pool_tail_dgrad", nb, k, c, p, ns, _ptr(z_prev), zbs, _ptr(coef_prev),
                      _ptr(wcat), _ptr(c0), _ptr(ent), _ptr(da), k * p, _ptr(part), st)
            if dw is not None:
                _check(dw); _f32(dw)
                assert dw.numel() == c * k
                part_m, part_s, part_w = f(nparts, k, k), f(nparts, k), f(nparts, c, k)
                ms = torch.empty(k * k + k, dtype=torch.float64, device=dev)
                _lib.call("nesie_pool_tail_wgrad", nb, k, c, p, ns, _ptr(z_prev), zbs, _ptr(coef_prev),
                          _ptr(ent), _ptr(ab), _ptr(w), _ptr(part_m), _ptr(part_s), _ptr(part_w),
                          _ptr(ms), _ptr(dw), st)
        return da, part

    def mlp_stat_finalize(self, part, count, gamma, beta, running_mean, running_var, momentum, eps,
                          coef, channel_major=False):
        """(parts, C, 2) -- or, channel_major, (C, parts, 2) -- unshifted (sum, sum of squares)
        partials -> coef (C, 4); running statistics updated in place."""
        _check(part, coef); _f32(part, coef)
        if channel_major:
            c, nparts, _ = part.shape
        else:
            nparts, c, _ = part.shape
        opt = lambda t: 0 if t is None else _ptr(t)  # noqa: E731
        with torch.cuda.device(coef.device):
            _lib.call("nesie_mlp_stat_finalize", c, nparts, float(count), _ptr(part), opt(gamma),
                      opt(beta), opt(running_mean), opt(running_var), float(momentum), float(eps),
                      _ptr(coef), int(bool(channel_major)), _stream(coef))

    def bn_relu_backward(self, dy, x, y, gamma, beta, save_mean, save_invstd, fwd_coef, relu,
                         dx, dgamma, dbeta, row_bias=None, d_row_bias=None, group=None):
        """y None with relu: x is the raw conv output of a fused forward (mask re-derived from
        fwd_coef).  d_row_bias without row_bias (+ ``group``): per-group sums of dx.
        save_mean / save_invstd None: columns 2 / 3 of fwd_coef."""
        _check(dy, x, dx, fwd_coef); _f32(dy, x, dx)
        assert (save_mean is None) == (save_invstd is None)
        if save_mean is not None:
            _check(save_mean, save_invstd)
        b, c = x.shape[:2]
        p = x.numel() // (b * c) if b * c else 0
        if row_bias is None and d_row_bias is not None:
            _check(d_row_bias); _f32(d_row_bias)
            assert group and d_row_bias.numel() == b * c * (p // group)
        else:
            group = _row_bias_group(x, row_bias)
        if row_bias is not None and row_bias.dim() != 1:
            _check(d_row_bias); _f32(d_row_bias)
            assert d_row_bias.shape == row_bias.shape
        elif row_bias is not None:
            assert d_row_bias is None  # per-channel bias before a batch norm: gradient is zero
        need = _lib.load().nesie_bn_workspace_bytes(b, c, p)
        with torch.cuda.device(x.device):
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
            opt = lambda t: 0 if t is None else _ptr(t)  # noqa: E731
            _lib.call("nesie_bn_relu_backward", b, c, p, _ptr(dy), _ptr(x), opt(y), opt(gamma),
                      opt(beta), opt(save_mean), opt(save_invstd), _ptr(fwd_coef),
                      int(bool(relu)), _ptr(dx), opt(dgamma), opt(dbeta), opt(row_bias), group,
                      opt(d_row_bias), _ptr(ws), need, _stream(x))


def _row_bias_group(x, row_bias):
    if row_bias is None:
        return 1
    _check(row_bias); _f32(row_bias)
    if row_bias.dim() == 1:  # one value per channel (a conv bias folded into the norm): group 0
        assert row_bias.shape[0] == x.shape[1], (tuple(x.shape), tuple(row_bias.shape))
        return 0
    assert x.dim() == 4 and tuple(row_bias.shape) == tuple(x.shape[:3]), \
        (tuple(x.shape), tuple(row_bias.shape))
    return int(x.shape[3])


_hip = None


def backend_for(tensor):
    """The back end that serves ``tensor``: injected one, else HIP (or raise)."""
    global _hip
    if _injected is not None:
        return _injected
    if not tensor.is_cuda:
        raise RuntimeError(
            f"nesie_amd op called on a {tensor.device} tensor: the product path is "
            "HIP-only (libnesie_hip.so); there is no CPU fallback")
    if _hip is None:
        _lib.load()
        _hip = HipKernels()
    return _hip


def injected_backend():
    return _injected


@contextlib.contextmanager
def spatial_index_scope():
    """One sample-then-group pass over ONE coordinate tensor (a set-abstraction forward): the
    spatial index the FPS kernel leaves behind may serve the ball queries inside this block and
    nothing after it."""
    HipKernels._index_scope_depth += 1
    try:
        yield
    finally:
        HipKernels._index_scope_depth -= 1
        if HipKernels._index_scope_depth == 0:
            HipKernels._spatial_index.clear()


@contextlib.contextmanager
def use_backend(backend):
    """Test/bench hook: run the enclosed code with ``backend`` serving every op."""
    global _injected
    prev = _injected
    _injected = backend
    try:
        yield backend
    finally:
        _injected = prev
