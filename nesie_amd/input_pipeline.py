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
        # ground truth padded to the widest scene, in the head's GTBatch form: padding columns
        # are zero-size boxes far away; an empty scene keeps the reference's all-zero fake box
        S, T = len(self._boxes), max(1, max(b.shape[0] for b in self._boxes))
        box_pad = torch.zeros(S, T, 7)
        box_pad[:, :, :3] = 1e6
        label_pad = torch.zeros(S, T, dtype=torch.long)
        valid = torch.zeros(S, T)
        for i, (b, l) in enumerate(zip(self._boxes, self._labels)):
            n = b.shape[0]
            if n == 0:
                box_pad[i, 0] = 0.0
            else:
                box_pad[i, :n], label_pad[i, :n], valid[i, :n] = b, l, 1.0
        self.box_counts = [b.shape[0] for b in self._boxes]
        self.box_pad, self.label_pad = box_pad.to(self.device), label_pad.to(self.device)
        self.box_valid = valid.to(self.device)
        self.box_count_dev = torch.tensor([max(n, 1) for n in self.box_counts], device=self.device)
        self._counts_dev = counts.to(self.device)
        self._offsets_dev = self.offsets.to(self.device)
        self._max_count = int(counts.max())
        return self

    def __len__(self):
        return len(self._xyz)

    def nbytes(self):
        return self.pool.numel() * 4 + self.height.numel() * 4

    # ---- random decisions ---------------------------------------------------------------
    def new_noise(self, batch, num_points=40000):
        """Static buffers for the uniform / normal variates one batch consumes: refresh them
        with ``refresh_noise`` (a few in-place launches, outside any captured graph) and pass
        them to ``draw_on_device`` / ``assemble_batch`` -- the rest of the assembly is then
        free of random-number calls and can be replayed as a hipGraph next to other graphs
        that use the default generator."""
        dev = self.device
        return dict(keys=torch.empty(batch, self._max_count, device=dev),
                    refill=torch.empty(batch, num_points, device=dev),
                    u=torch.empty(batch, 4, device=dev), normal=torch.empty(batch, 3, device=dev))

    @staticmethod
    def refresh_noise(noise, generator=None):
        for k in ('keys', 'refill', 'u'):
            noise[k].uniform_(generator=generator)
        noise['normal'].normal_(generator=generator)
        return noise

    def draw_on_device(self, scene_ids, num_points=40000, generator=None, flip_ratio_h=0.5,
                       flip_ratio_v=0.5, rot_range=(-0.087266, 0.087266),
                       scale_range=(1.0, 1.0), translation_std=(0, 0, 0), noise=None):
        """Same distributions as the reference, drawn with the device generator:
        -> choices (B,n) int32 pool rows, xform (B,20), and the (B,) flip / angle / scale /
        (B,3) trans tensors the boxes need.  Without replacement when the scene has at least
        ``num_points`` points (random keys, smallest n), with replacement otherwise.
        ``noise``: pre-drawn variates (``new_noise``) instead of drawing here."""
        ids = torch.as_tensor(scene_ids, device=self.device)
        B = ids.numel()
        cnt, off = self._counts_dev[ids], self._offsets_dev[ids]
        nmax = self._max_count            # static: the draw is capturable in a hipGraph
        if noise is None:
            noise = self.refresh_noise(self.new_noise(B, num_points), generator)
        keys, u = noise['keys'], noise['u']
        keys = torch.where(torch.arange(nmax, device=self.device)[None] < cnt[:, None], keys,
                           keys.new_full((), 2.0))
        if nmax >= num_points:
            local = torch.argsort(keys, dim=1)[:, :num_points]   # the n smallest random keys
        else:
            local = torch.zeros(B, num_points, dtype=torch.int64, device=self.device)
        short = cnt < num_points                              # with replacement
        refill = torch.minimum((noise['refill'] * cnt[:, None]).long(), cnt[:, None] - 1)
        local = torch.where(short[:, None], refill, local)
        choices = (local + off[:, None]).to(torch.int32)
        flip_h, flip_v = u[:, 0] < flip_ratio_h, u[:, 1] < flip_ratio_v
        angle = rot_range[0] + (rot_range[1] - rot_range[0]) * u[:, 2]
        scale = scale_range[0] + (scale_range[1] - scale_range[0]) * u[:, 3]
        trans = torch.stack([noise['normal'][:, k] * float(translation_std[k]) for k in range(3)],
                            dim=1)
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
    # ---- host-drawn decisions, staged through ONE pinned buffer -------------------------------
    def new_staging(self, batch, num_points=40000):
        """Buffers for a batch whose decisions are drawn on the HOST in the reference's order
        (``draw_like_reference``, numpy): B x num_points pool rows (int32), the B x 20 point
        transforms and the B x T x 7 augmented boxes (float32), packed in one int32 array so that
        one DMA copy brings them over.
        -> dict(host=[pinned array, pinned array] (double buffer), dev=device array, ...)."""
        T = self.box_pad.shape[1]
        words = batch * (num_points + 20 + T * 7)
        pin = self.device.type == 'cuda'
        return dict(host=[torch.empty(words, dtype=torch.int32, pin_memory=pin) for _ in range(2)],
                    dev=torch.empty(words, dtype=torch.int32, device=self.device), n=num_points,
                    batch=batch, boxes=T)

    def _staging_views(self, buf, staging):
        B, n, T = staging['batch'], staging['n'], staging['boxes']
        rows = buf[:B * n].view(B, n)
        xform = buf[B * n:B * (n + 20)].view(torch.float32).view(B, 20)
        boxes = buf[B * (n + 20):].view(torch.float32).view(B, T, 7)
        return rows, xform, boxes

    def stage_draws(self, staging, which, scene_ids, rng, **ranges):
        """Draw one batch's decisions for ``scene_ids`` (host ints) with ``rng`` exactly as the
        reference's pipeline would (IndoorPointSample, RandomFlip3D, GlobalRotScaleTrans), apply
        them to the boxes and fold them into the point transform -- all on the host, as the
        reference's data-loader workers do -- and write the result into pinned buffer ``which``."""
        rows, xform, boxes = self._staging_views(staging['host'][which], staging)
        host = self._host_tables()
        draws = [draw_like_reference(rng, int(self.counts[sid]), staging['n'], **ranges) for sid in scene_ids]
        for i, (sid, d) in enumerate(zip(scene_ids, draws)):
            rows[i] = torch.from_numpy(d.choices + int(self.offsets[sid]))
        ids = torch.as_tensor(list(scene_ids))
        f32 = dict(dtype=torch.float32)
        flip_h, flip_v = torch.tensor([d.flip_h for d in draws]), torch.tensor([d.flip_v for d in draws])
        angle, scale = torch.tensor([d.angle for d in draws], **f32), torch.tensor([d.scale for d in draws], **f32)
        trans = torch.tensor(np.stack([d.trans for d in draws]), **f32)
        one = torch.ones_like(angle)
        xform.copy_(torch.cat([host['align'][ids], torch.where(flip_h, -one, one)[:, None],
                               torch.where(flip_v, -one, one)[:, None], torch.cos(angle)[:, None],
                               torch.sin(angle)[:, None], scale[:, None], trans], dim=1))
        boxes.copy_(self._augment_boxes(host['box_pad'][ids], host['box_valid'][ids], flip_h, flip_v,
                                        angle, scale, trans))

    def _host_tables(self):
        if getattr(self, '_host', None) is None:
            self._host = dict(align=self.align.cpu(), box_pad=self.box_pad.cpu(), box_valid=self.box_valid.cpu())
        return self._host

    def assemble_batch(self, scene_ids, draws=None, num_points=40000, generator=None,
                       noise=None, staging=None, **ranges):
        """-> points (B, n, 4) = (x, y, z, height) and the ground truth as the head's GTBatch
        (boxes (B,T,7) bottom-origin, labels, count, valid), all on the device, no host
        synchronisation.  ``scene_ids``: list or device tensor; ``draws``: list of
        ``AugmentDraws`` (reference order) or None = draw on the device; with ``noise``
        (``new_noise``) and ``scene_ids`` a static device tensor the call makes no random-number
        and no host-side call and is capturable in a hipGraph; so it is with ``staging``
        (``new_staging`` / ``stage_draws``: the decisions drawn AND applied to boxes / transforms on
        the host in the reference's order and brought over by one copy -- the device runs the
        gather kernel and three small index gathers, nothing else)."""
        from .votenet.nesie_head import GTBatch
        if staging is not None:     # drawn and applied on the host, already copied into staging['dev']
            ids = torch.as_tensor(scene_ids, device=self.device)
            choices, xform, boxes = self._staging_views(staging['dev'], staging)
            out = torch.empty(choices.shape[0], choices.shape[1], 4, dtype=torch.float32, device=self.device)
            backend_for(self.pool).scene_assemble(self.pool, self.height, choices, xform, out)
            return out, GTBatch(boxes, self.label_pad[ids], self.box_count_dev[ids], self.box_valid[ids])
        if draws is None:
            choices, xform, dec = self.draw_on_device(scene_ids, num_points, generator,
                                                      noise=noise, **ranges)
        else:
            choices, xform, dec = self._from_draws(scene_ids, draws)
        ids = torch.as_tensor(scene_ids, device=self.device)
        out = torch.empty(choices.shape[0], choices.shape[1], 4, dtype=torch.float32,
                          device=self.device)
        backend_for(self.pool).scene_assemble(self.pool, self.height, choices.contiguous(),
                                              xform, out)
        valid = self.box_valid[ids]
        boxes = self._augment_boxes(self.box_pad[ids], valid, *dec)
        return out, GTBatch(boxes, self.label_pad[ids], self.box_count_dev[ids], valid)

    def assemble(self, scene_ids, draws=None, num_points=40000, generator=None, **ranges):
        """List form: points, [(T_i,7) boxes], [(T_i,) labels] (host-known box counts)."""
        pts, gt = self.assemble_batch(list(scene_ids), draws, num_points, generator, **ranges)
        n = [self.box_counts[s] for s in scene_ids]
        return pts, [gt.boxes[i, :k] for i, k in enumerate(n)], \
            [gt.labels[i, :k] for i, k in enumerate(n)]

    def _augment_boxes(self, boxes, valid, flip_h, flip_v, angle, scale, trans):
        """RandomFlip3D + GlobalRotScaleTrans on the padded boxes of a batch (B,T,7)
        (depth_box3d.py:118-214, base_box3d.py:149-157, 215-222): tensor ops only, the flips
        applied as signs; padding and fake boxes pass through unchanged."""
        b = boxes.clone()
        B, T = b.shape[:2]
        sx = torch.where(flip_h, -1.0, 1.0).to(b.dtype).view(B, 1)
        sy = torch.where(flip_v, -1.0, 1.0).to(b.dtype).view(B, 1)
        b[..., 0] = b[..., 0] * sx
        b[..., 1] = b[..., 1] * sy
        if self.with_yaw:
            # horizontal: yaw = -yaw + pi; vertical: yaw = -yaw
            yaw = torch.where(flip_h.view(B, 1), -b[..., 6] + math.pi, b[..., 6])
            b[..., 6] = torch.where(flip_v.view(B, 1), -yaw, yaw)
        sin, cos = torch.sin(angle), torch.cos(angle)
        zero, one = torch.zeros_like(sin), torch.ones_like(sin)
        rot_t = torch.stack([torch.stack([cos, sin, zero], -1), torch.stack([-sin, cos, zero], -1),
                             torch.stack([zero, zero, one], -1)], 1)   # (B,3,3) = [[c,-s,0],[s,c,0],[0,0,1]].T
        b[..., 0:3] = b[..., 0:3] @ rot_t
        if self.with_yaw:
            b[..., 6] -= angle.view(B, 1)
        else:
            holder = DepthInstance3DBoxes.__new__(DepthInstance3DBoxes)
            holder.tensor = b.reshape(B * T, 7)
            corners = holder.corners.view(B, T * 8, 3) @ rot_t
            corners = corners.view(B, T, 8, 3)
            b[..., 3] = corners[..., 0].max(dim=2)[0] - corners[..., 0].min(dim=2)[0]
            b[..., 4] = corners[..., 1].max(dim=2)[0] - corners[..., 1].min(dim=2)[0]
        b[..., :6] *= scale.view(B, 1, 1)
        b[..., :3] += trans.view(B, 1, 3)
        return torch.where(valid.unsqueeze(-1) > 0, b, boxes)
