"""Input side of the training step with the data set resident in HBM.

The reference prepares every sample on CPU workers (configs/Nesie/*-pretrain-*.py:151-196):
``LoadPointsFromFile(shift_height)`` -> ``GlobalAlignment`` -> ``IndoorPointSample(40000)`` ->
``RandomFlip3D`` -> ``GlobalRotScaleTrans`` -> collate -> host-to-device copy.  ScanNet's 1 201
training scenes of <= 50 000 points are 0.7 GB of xyz: here they are uploaded ONCE
(``ResidentScenes``), the per-scene constants are fixed at load time (shifted height, axis
alignment, bottom-origin boxes) and a step's batch is one gather kernel
(``nesie_scene_assemble``) plus a handful of small tensor ops for the boxes -- no per-step
host work except drawing ~10 random numbers per scene.

Random numbers: ``draw_like_reference`` consumes a numpy generator in exactly the order the
reference pipeline does, so a seeded run reproduces the reference's samples; ``draw_on_device``
is the production form (device RNG, no host round trip).
"""
import math

import numpy as np
import torch

from .kernels import backend_for
from .votenet.boxes import DepthInstance3DBoxes


def load_points_bin(path, load_dim=6, use_dim=(0, 1, 2)):
    """``LoadPointsFromFile._load_points`` + column selection (loading.py:379-414): a flat
    float32 file of ``load_dim`` columns."""
    pts = np.fromfile(path, dtype=np.float32).reshape(-1, load_dim)
    return pts[:, list(use_dim)]


class AugmentDraws:
    """The random decisions of one sample: rows to keep, the two flips, rotation angle, scale
    factor, translation."""

    def __init__(self, choices, flip_h, flip_v, angle, scale, trans):
        self.choices = np.asarray(choices, dtype=np.int64)
        self.flip_h, self.flip_v = bool(flip_h), bool(flip_v)
        self.angle, self.scale = float(angle), float(scale)
        self.trans = np.asarray(trans, dtype=np.float64).reshape(3)


def draw_like_reference(rng, num_raw, num_points=40000, flip_ratio_h=0.5, flip_ratio_v=0.5,
                        rot_range=(-0.087266, 0.087266), scale_range=(1.0, 1.0),
                        translation_std=(0, 0, 0)):
    """One sample's draws from ``rng`` (``numpy.random`` itself or a ``RandomState``) in the
    order the reference pipeline makes them:
      IndoorPointSample   ``choice(N, num, replace=N < num)``          transforms_3d.py:857-860
      RandomFlip3D        mmdet ``RandomFlip.__call__`` first draws
                          ``choice([direction, None], p=[r, 1 - r])`` (third party, mmdet
                          2.19 transforms.py; its result only concerns images), then
                          ``rand() < r_h``, ``rand() < r_v``             :142-150
      GlobalRotScaleTrans ``uniform(rot)``, ``uniform(scale)``,
                          ``normal(scale=std, size=3)``                 :572, 620-621, 553"""
    choices = rng.choice(num_raw, num_points, replace=num_raw < num_points)
    rng.choice(['horizontal', None], p=[flip_ratio_h, 1 - flip_ratio_h])
    flip_h = rng.rand() < flip_ratio_h
    flip_v = rng.rand() < flip_ratio_v
    angle = rng.uniform(rot_range[0], rot_range[1])
    scale = rng.uniform(scale_range[0], scale_range[1])
    trans = rng.normal(scale=np.array(translation_std, dtype=np.float32), size=3).T
    return AugmentDraws(choices, flip_h, flip_v, angle, scale, trans)


class ResidentScenes:
    """A data set held on the device: xyz pool, shifted-height column, per-scene alignment and
    ground truth."""

    def __init__(self, device, with_yaw=False):
        self.device = torch.device(device)
        self.with_yaw = with_yaw          # ScanNet boxes are axis-aligned (scannet_dataset.py:97-101)
        self._xyz, self._height, self._align, self._boxes, self._labels = [], [], [], [], []
        self.pool = self.height = self.offsets = self.counts = self.align = None

    def add_scene(self, points, axis_align_matrix=None, gt_boxes=None, gt_labels=None):
        """points (N, >=3) raw xyz; axis_align_matrix (4,4) or None; gt_boxes (T, 6|7)
        gravity-centre boxes as the info files hold them (``gt_boxes_upright_depth``)."""
        xyz = np.ascontiguousarray(np.asarray(points, dtype=np.float32)[:, :3])
        # loading.py:424-430: float64 percentile, float64 subtraction, float32 on boxing
        floor = np.percentile(xyz[:, 2], 0.99)
        self._xyz.append(torch.from_numpy(xyz))
        self._height.append(torch.from_numpy((xyz[:, 2] - floor).astype(np.float32)))
        a = np.eye(4) if axis_align_matrix is None else np.asarray(axis_align_matrix)
        assert a.shape == (4, 4)
        self._align.append(torch.tensor(np.concatenate([a[:3, :3].reshape(-1), a[:3, 3]]),
                                        dtype=torch.float32))
        if gt_boxes is None:
            gt_boxes = np.zeros((0, 7 if self.with_yaw else 6), np.float32)
        gt_boxes = np.asarray(gt_boxes, dtype=np.float32)
        self._boxes.append(DepthInstance3DBoxes(gt_boxes, box_dim=gt_boxes.shape[-1],
                                                origin=(0.5, 0.5, 0.5)).tensor)
        self._labels.append(torch.as_tensor(np.asarray(
            gt_labels if gt_labels is not None else np.zeros((0,), np.int64))).long())

    def finalize(self):
        counts = torch.tensor([x.shape[0] for x in self._xyz], dtype=torch.int64)
        self.counts = counts
        self.offsets = torch.cumsum(counts, 0) - counts
        self.pool = torch.cat(self._xyz).to(self.device).contiguous()
        self.height = torch.cat(self._height).to(self.device).contiguous()
        self.align = torch.stack(self._align).to(self.device)
        self._boxes = [b.to(self.device) for b in self._boxes]
        self._labels = [l.to(self.device) for l in self._labels]
        self._counts_dev = counts.to(self.device)
        self._offsets_dev = self.offsets.to(self.device)
        return self

    def __len__(self):
        return len(self._xyz)

    def nbytes(self):
        return self.pool.numel() * 4 + self.height.numel() * 4

    # ---- random decisions ---------------------------------------------------------------
    def draw_on_device(self, scene_ids, num_points=40000, generator=None, flip_ratio_h=0.5,
                       flip_ratio_v=0.5, rot_range=(-0.087266, 0.087266),
                       scale_range=(1.0, 1.0), translation_std=(0, 0, 0)):
        """Same distributions as the reference, drawn with the device generator:
        -> choices (B,n) int32 pool rows, xform (B,20), and the (B,) flip / angle / scale /
        (B,3) trans tensors the boxes need.  Without replacement when the scene has at least
        ``num_points`` points (random keys, smallest n), with replacement otherwise."""
        ids = torch.as_tensor(scene_ids, device=self.device)
        B = ids.numel()
        cnt, off = self._counts_dev[ids], self._offsets_dev[ids]
        nmax = int(self.counts[torch.as_tensor(scene_ids)].max())
        rand = lambda *s: torch.rand(*s, device=self.device, generator=generator)  # noqa: E731
        keys = rand(B, nmax)
        keys = torch.where(torch.arange(nmax, device=self.device)[None] < cnt[:, None], keys,
                           keys.new_full((), 2.0))
        if nmax >= num_points:
            local = keys.topk(num_points, dim=1, largest=False, sorted=False)[1]
        else:
            local = torch.zeros(B, num_points, dtype=torch.int64, device=self.device)
        short = cnt < num_points                              # with replacement
        refill = (rand(B, num_points) * cnt[:, None]).long().clamp_(max=int(self.counts.max()) - 1)
        refill = torch.minimum(refill, cnt[:, None] - 1)
        local = torch.where(short[:, None], refill, local)
        choices = (local + off[:, None]).to(torch.int32)
        flip_h, flip_v = rand(B) < flip_ratio_h, rand(B) < flip_ratio_v
        angle = rot_range[0] + (rot_range[1] - rot_range[0]) * rand(B)
        scale = scale_range[0] + (scale_range[1] - scale_range[0]) * rand(B)
        trans = torch.randn(B, 3, device=self.device, generator=generator) \
            * torch.tensor(translation_std, dtype=torch.float32, device=self.device)
        return choices, self._xform(ids, flip_h, flip_v, angle, scale, trans), \
            (flip_h, flip_v, angle, scale, trans)

    def _xform(self, ids, flip_h, flip_v, angle, scale, trans):
        one = torch.ones_like(angle)
        return torch.cat([self.align[ids], torch.where(flip_h, -one, one)[:, None],
                          torch.where(flip_v, -one, one)[:, None], torch.cos(angle)[:, None],
                          torch.sin(angle)[:, None], scale[:, None], trans], dim=1).contiguous()

    def _from_draws(self, scene_ids, draws):
        f32 = dict(dtype=torch.float32, device=self.device)
        ids = torch.as_tensor(scene_ids, device=self.device)
        choices = torch.stack([torch.from_numpy(d.choices + int(self.offsets[s]))
                               for s, d in zip(scene_ids, draws)]).to(self.device, torch.int32)
        flip_h = torch.tensor([d.flip_h for d in draws], device=self.device)
        flip_v = torch.tensor([d.flip_v for d in draws], device=self.device)
        # the reference boxes the python floats as float32 tensors before sin / cos / multiply
        angle = torch.tensor([d.angle for d in draws], **f32)
        scale = torch.tensor([d.scale for d in draws], **f32)
        trans = torch.tensor(np.stack([d.trans for d in draws]), **f32)
        return choices, self._xform(ids, flip_h, flip_v, angle, scale, trans), \
            (flip_h, flip_v, angle, scale, trans)

    # ---- batch assembly -----------------------------------------------------------------
    def assemble(self, scene_ids, draws=None, num_points=40000, generator=None, **ranges):
        """-> points (B, n, 4) = (x, y, z, height), list of (T_i, 7) bottom-origin boxes, list
        of (T_i,) labels, all on the device.  ``draws``: list of ``AugmentDraws`` (reference
        order); None = draw on the device."""
        if draws is None:
            choices, xform, dec = self.draw_on_device(scene_ids, num_points, generator, **ranges)
        else:
            choices, xform, dec = self._from_draws(scene_ids, draws)
        out = torch.empty(choices.shape[0], choices.shape[1], 4, dtype=torch.float32,
                          device=self.device)
        backend_for(self.pool).scene_assemble(self.pool, self.height, choices.contiguous(),
                                              xform, out)
        boxes = [self._augment_boxes(self._boxes[s], *(t[i] for t in dec))
                 for i, s in enumerate(scene_ids)]
        return out, boxes, [self._labels[s] for s in scene_ids]

    def _augment_boxes(self, boxes, flip_h, flip_v, angle, scale, trans):
        """RandomFlip3D + GlobalRotScaleTrans on the boxes of one scene
        (depth_box3d.py:118-214, base_box3d.py:149-157, 215-222); tensor ops only, no host
        synchronisation (the flips are applied as signs)."""
        b = boxes.clone()
        if b.shape[0] == 0:
            return b
        sx = torch.where(flip_h, -1.0, 1.0).to(b.dtype)
        sy = torch.where(flip_v, -1.0, 1.0).to(b.dtype)
        b[:, 0] = b[:, 0] * sx
        b[:, 1] = b[:, 1] * sy
        if self.with_yaw:
            # horizontal: yaw = -yaw + pi; vertical: yaw = -yaw
            yaw = torch.where(flip_h, -b[:, 6] + math.pi, b[:, 6])
            b[:, 6] = torch.where(flip_v, -yaw, yaw)
        sin, cos = torch.sin(angle), torch.cos(angle)
        zero, one = torch.zeros_like(sin), torch.ones_like(sin)
        rot_t = torch.stack([torch.stack([cos, sin, zero]), torch.stack([-sin, cos, zero]),
                             torch.stack([zero, zero, one])])          # [[c,-s,0],[s,c,0],[0,0,1]].T
        b[:, 0:3] = b[:, 0:3] @ rot_t
        if self.with_yaw:
            b[:, 6] -= angle
        else:
            holder = DepthInstance3DBoxes.__new__(DepthInstance3DBoxes)
            holder.tensor = b
            corners = holder.corners @ rot_t
            b[:, 3] = corners[..., 0].max(dim=1)[0] - corners[..., 0].min(dim=1)[0]
            b[:, 4] = corners[..., 1].max(dim=1)[0] - corners[..., 1].min(dim=1)[0]
        b[:, :6] *= scale
        b[:, :3] += trans
        return b
