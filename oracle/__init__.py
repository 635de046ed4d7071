"""CPU oracle for the nesie_amd kernels -- TEST INFRASTRUCTURE ONLY.

Loads ``oracle/liboracle.so`` (built from ``nesie_oracle.c`` by
``oracle/Makefile``; see that file's header for what it restates and its pin
status) and, when present, ``oracle/_ref/points_in_boxes_ref.so`` (the
reference's own CPU source compiled where it lies).  Only tests/,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of bench.py may
import this package; ``nesie_amd`` never does.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_PIB_PATH = os.path.join(_HERE, "_ref", "points_in_boxes_ref.so")

_P, _I, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_SIGS = {
    "oracle_fps_block_size": [_I],
    "oracle_furthest_point_sampling": [_I, _I, _I, _P, _P, _P],
    "oracle_furthest_point_sampling_with_dist": [_I, _I, _I, _P, _P, _P],
    "oracle_ball_query": [_I, _I, _I, _F, _F, _I, _P, _P, _P],
    "oracle_group_points": [_I, _I, _I, _I, _I, _P, _P, _P],
    "oracle_group_points_grad": [_I, _I, _I, _I, _I, _P, _P, _P],
    "oracle_gather_points": [_I, _I, _I, _I, _P, _P, _P],
    "oracle_gather_points_grad": [_I, _I, _I, _I, _P, _P, _P],
    "oracle_three_nn": [_I, _I, _I, _P, _P, _P, _P],
    "oracle_three_interpolate": [_I, _I, _I, _I, _P, _P, _P, _P],
    "oracle_three_interpolate_grad": [_I, _I, _I, _I, _P, _P, _P, _P],
    "oracle_sort_vertices": [_I, _I, _I, _P, _P, _P, _P],
    "oracle_points_in_boxes_batch": [_I, _I, _I, _P, _P, _P],
    "oracle_lhs_nms_samecls": [_I, _I, _P, _F, _P],
    "oracle_aligned_3d_nms": [_I, _I, _P, _P, _P, _P, _F, _P, _P],
    "oracle_points_in_boxes_count": [_I, _I, _I, _P, _P, _P],
    "oracle_boxes_overlap_bev": [_I, _P, _I, _P, _P],
    "oracle_scene_assemble": [_I, _I, ctypes.c_longlong, _P, _P, _P, _P, _P],
    "oracle_num_threads": [],
}
_lib = None
_ref = None


def build():
    """(Re)build liboracle.so and, if the reference tree exists, oracle/_ref."""
    subprocess.run(["make", "-C", _HERE, "all"], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = ctypes.CDLL(LIB_PATH)
        for name, at in _SIGS.items():
            f = getattr(L, name)
            f.argtypes = at
            f.restype = _I
        _lib = L
    return _lib


def ref_points_in_boxes_available():
    return os.path.exists(REF_PIB_PATH)


def ref_points_in_boxes_cpu(boxes, pts):
    """The REFERENCE's points_in_boxes_cpu (compiled from its own source).

    boxes (T,7) LiDAR frame, pts (M,3) -> int32 (T, M).
    """
    global _ref
    if _ref is None:
        _ref = ctypes.CDLL(REF_PIB_PATH)
        _ref.ref_points_in_boxes_cpu.argtypes = [_P, _I, _P, _I, _P]
        _ref.ref_points_in_boxes_cpu.restype = _I
    boxes = boxes.contiguous().float()
    pts = pts.contiguous().float()
    out = torch.zeros(boxes.shape[0], pts.shape[0], dtype=torch.int32)
    _ref.ref_points_in_boxes_cpu(boxes.data_ptr(), boxes.shape[0], pts.data_ptr(),
                                 pts.shape[0], out.data_ptr())
    return out


def _cpu(*ts):
    for t in ts:
        assert t.device.type == "cpu" and t.is_contiguous(), "oracle takes contiguous CPU tensors"


def set_distance_form(form):
    """0 (default): contraction-free squared distance; 1 / 2: the fused forms nvcc's -fmad=true
    could make of the reference's expression (nesie_oracle.c header).  Process-wide."""
    if lib().oracle_set_distance_form(int(form)) != 0:
        raise ValueError(f'distance form {form} (0, 1 or 2)')


def get_distance_form():
    return lib().oracle_get_distance_form()


class OracleKernels:
    """Same method surface as nesie_amd.kernels.HipKernels, on CPU tensors."""

    name = "oracle"

    def furthest_point_sampling_wrapper(self, b, n, m, xyz, temp, idx):
        _cpu(xyz, temp, idx)
        lib().oracle_furthest_point_sampling(b, n, m, xyz.data_ptr(), temp.data_ptr(),
                                             idx.data_ptr())

    def furthest_point_sampling_with_dist_wrapper(self, b, n, m, dist, temp, idx):
        _cpu(dist, temp, idx)
        lib().oracle_furthest_point_sampling_with_dist(b, n, m, dist.data_ptr(),
                                                       temp.data_ptr(), idx.data_ptr())

    def ball_query_wrapper(self, b, n, m, min_radius, max_radius, nsample, new_xyz, xyz,
                           idx):
        _cpu(new_xyz, xyz, idx)
        lib().oracle_ball_query(b, n, m, float(min_radius), float(max_radius), nsample,
                                new_xyz.data_ptr(), xyz.data_ptr(), idx.data_ptr())

    def group_points_forward(self, b, c, n, npoints, nsample, points, idx, out):
        _cpu(points, idx, out)
        lib().oracle_group_points(b, c, n, npoints, nsample, points.data_ptr(),
                                  idx.data_ptr(), out.data_ptr())

    def group_points_backward(self, b, c, n, npoints, nsample, grad_out, idx, grad_points):
        _cpu(grad_out, idx, grad_points)
        lib().oracle_group_points_grad(b, c, n, npoints, nsample, grad_out.data_ptr(),
                                       idx.data_ptr(), grad_points.data_ptr())

    def gather_points_wrapper(self, b, c, n, npoints, points, idx, out):
        _cpu(points, idx, out)
        lib().oracle_gather_points(b, c, n, npoints, points.data_ptr(), idx.data_ptr(),
                                   out.data_ptr())

    def gather_points_grad_wrapper(self, b, c, n, npoints, grad_out, idx, grad_points):
        _cpu(grad_out, idx, grad_points)
        lib().oracle_gather_points_grad(b, c, n, npoints, grad_out.data_ptr(),
                                        idx.data_ptr(), grad_points.data_ptr())

    def query_and_group_forward(self, xyz, centres, features, idx, radius, out):
        """The reference's order: transpose, group, subtract, divide, concatenate."""
        b, n = xyz.shape[:2]
        m, ns = idx.shape[1], idx.shape[2]
        xt = xyz.transpose(1, 2).contiguous()
        g = xyz.new_empty(b, 3, m, ns)
        self.group_points_forward(b, 3, n, m, ns, xt, idx, g)
        g = g - centres.transpose(1, 2).unsqueeze(-1)
        if radius > 0:
            g = g / radius
        out[:, :3] = g
        if features is not None:
            c = features.shape[1]
            gf = xyz.new_empty(b, c, m, ns)
            self.group_points_forward(b, c, n, m, ns, features.contiguous(), idx, gf)
            out[:, 3:] = gf

    def query_and_group_backward(self, grad_out, idx, grad_features):
        b, c, n = grad_features.shape
        m, ns = idx.shape[1], idx.shape[2]
        self.group_points_backward(b, c, n, m, ns, grad_out[:, 3:].contiguous(), idx,
                                   grad_features)

    def three_nn_wrapper(self, b, n, m, unknown, known, dist2, idx):
        _cpu(unknown, known, dist2, idx)
        lib().oracle_three_nn(b, n, m, unknown.data_ptr(), known.data_ptr(),
                              dist2.data_ptr(), idx.data_ptr())

    def three_interpolate_wrapper(self, b, c, m, n, points, idx, weight, out):
        _cpu(points, idx, weight, out)
        lib().oracle_three_interpolate(b, c, m, n, points.data_ptr(), idx.data_ptr(),
                                       weight.data_ptr(), out.data_ptr())

    def blend_conv_forward(self, table, seg_off, idx, weight, rel, wx, out, segs, seg_len,
                           c, c_offset):
        """Restated from the reference's pieces: per face, three_interpolate on the (C, M)
        table, the view/split order of side_pooling_module.py:226-243, 304-313, and the xyz
        term of the first conv as a plain matmul."""
        b, m, _ = table.shape
        n = idx.shape[1]
        k = n // (segs * seg_len)
        for s_ in range(segs):
            pts = table[:, :, s_ * seg_off:s_ * seg_off + c].transpose(1, 2).contiguous()
            flat = pts.new_empty(b, c, n)
            self.three_interpolate_wrapper(b, c, m, n, pts, idx, weight, flat)
            face = flat.view(b, c, k, segs, seg_len)[:, :, :, s_].reshape(b, c, k * seg_len)
            if wx is not None:
                r = rel.view(b, k, segs, seg_len, 3)[:, :, s_].reshape(b, k * seg_len, 3)
                face = face + torch.matmul(wx[s_].unsqueeze(0), r.transpose(1, 2))
            out[:, s_, c_offset:c_offset + c] = face

    def blend_conv_backward(self, dy, seg_off, idx, weight, rel, d_table, d_wx, segs, seg_len):
        b, m, _ = d_table.shape
        n, c = idx.shape[1], dy.shape[2]
        k = n // (segs * seg_len)
        full = dy.view(b, segs, c, k, seg_len).permute(0, 2, 3, 1, 4).reshape(b, c, n)
        for face in range(segs):
            only = torch.zeros_like(full).view(b, c, k, segs, seg_len)
            only[:, :, :, face] = full.view(b, c, k, segs, seg_len)[:, :, :, face]
            gp = dy.new_zeros(b, c, m)
            self.three_interpolate_grad_wrapper(b, c, n, m, only.view(b, c, n).contiguous(),
                                                idx, weight, gp)
            d_table[:, :, face * seg_off:face * seg_off + c] += gp.transpose(1, 2)
            if d_wx is not None:
                r = rel.view(b, k, segs, seg_len, 3)[:, :, face].reshape(b, k * seg_len, 3)
                d_wx[face] = torch.matmul(dy[:, face], r).sum(0)      # (written, as the HIP wrapper does)

    def three_interpolate_grad_wrapper(self, b, c, n, m, grad_out, idx, weight,
                                       grad_points):
        _cpu(grad_out, idx, weight, grad_points)
        lib().oracle_three_interpolate_grad(b, c, n, m, grad_out.data_ptr(),
                                            idx.data_ptr(), weight.data_ptr(),
                                            grad_points.data_ptr())

    def sort_vertices_forward(self, vertices, mask, num_valid, idx):
        _cpu(vertices, mask, num_valid, idx)
        b, n, m, _ = vertices.shape
        assert mask.dtype == torch.bool
        lib().oracle_sort_vertices(b, n, m, vertices.data_ptr(), mask.data_ptr(),
                                   num_valid.data_ptr(), idx.data_ptr())

    def rotated_iou_3d(self, box_a, box_b):
        """the reference's differentiable torch chain around sort_vertices (oracle/rotated_iou.py)"""
        from . import rotated_iou

        def order(vertices, mask, num_valid):
            v = vertices.float().contiguous()
            idx = torch.empty(v.shape[0], v.shape[1], 9, dtype=torch.int32)
            self.sort_vertices_forward(v, mask.contiguous(), num_valid.contiguous(), idx)
            return idx
        return rotated_iou.rotated_iou_3d(box_a, box_b, order)

    def points_in_boxes_batch(self, boxes, pts, out):
        _cpu(boxes, pts, out)
        b, t, _ = boxes.shape
        lib().oracle_points_in_boxes_batch(b, t, pts.shape[1], boxes.data_ptr(),
                                           pts.data_ptr(), out.data_ptr())


    def lhs_nms_samecls(self, boxes, thr, keep):
        _cpu(boxes, keep)
        b, k, _ = boxes.shape
        lib().oracle_lhs_nms_samecls(b, k, boxes.data_ptr(), float(thr), keep.data_ptr())

    def aligned_3d_nms(self, boxes, scores, classes, valid, thr, picks, count):
        _cpu(boxes, scores, classes, picks, count)
        b, k = scores.shape
        if valid is not None:
            _cpu(valid)
        lib().oracle_aligned_3d_nms(b, k, boxes.data_ptr(), scores.data_ptr(), classes.data_ptr(),
                                    None if valid is None else valid.data_ptr(), float(thr),
                                    picks.data_ptr(), count.data_ptr())

    def points_in_boxes_count(self, boxes, pts, counts):
        _cpu(boxes, pts, counts)
        b, t, _ = boxes.shape
        lib().oracle_points_in_boxes_count(b, t, pts.shape[1], boxes.data_ptr(), pts.data_ptr(),
                                           counts.data_ptr())

    def boxes_overlap_bev(self, boxes_a, boxes_b, ans_overlap):
        _cpu(boxes_a, boxes_b, ans_overlap)
        lib().oracle_boxes_overlap_bev(boxes_a.shape[0], boxes_a.data_ptr(), boxes_b.shape[0],
                                       boxes_b.data_ptr(), ans_overlap.data_ptr())

    def scene_assemble(self, pool, height, choices, xform, out):
        _cpu(pool, height, choices, xform, out)
        b, n = choices.shape
        lib().oracle_scene_assemble(b, n, pool.shape[0], pool.data_ptr(), height.data_ptr(),
                                    choices.data_ptr(), xform.data_ptr(), out.data_ptr())

    # dense-op stand-ins of the CPU path (PyTorch-CPU, first-index tie rule like ATen)
    def group_max_pool_forward(self, x, out, argmax):
        _cpu(x, out, argmax)
        v, i = torch.max(x, dim=-1)
        # torch.max(dim) returns an arbitrary index on ties; recompute the first one
        first = (x == v.unsqueeze(-1)).to(torch.uint8).argmax(dim=-1)
        out.copy_(v)
        argmax.copy_(first.to(torch.uint8))

    def group_max_pool_backward(self, grad_out, argmax, grad_x):
        _cpu(grad_out, argmax, grad_x)
        grad_x.zero_()
        grad_x.scatter_(-1, argmax.long().unsqueeze(-1), grad_out.unsqueeze(-1))

    def group_max_pool_backward_add(self, grad_out, argmax, grad_x):
        flat = grad_x.view(-1, grad_x.shape[-1])
        flat.scatter_add_(1, argmax.reshape(-1, 1).long(), grad_out.reshape(-1, 1))



def num_threads():
    return lib().oracle_num_threads()
