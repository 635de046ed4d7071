"""The slice of ``mmdet3d/core/bbox/structures/depth_box3d.py`` (+ ``base_box3d.py``) the
training step and the test path touch: gravity centres (:42-48), ``points_in_boxes``
(:251-277) with the depth->LiDAR frame change of ``box_3d_mode.py:124-127``, ``corners``
(:50-89), indexing, and the 3-D ``overlaps`` of evaluation (base_box3d.py:355-438)."""
import torch

from ..kernels import injected_backend
from ..mmdet3d_ops import boxes_overlap_bev, points_in_boxes_batch


def depth_to_lidar_points(points):
    """(…,3) depth-frame xyz -> LiDAR frame (y, -x, z)  (depth_box3d.py:263-266)."""
    return torch.stack([points[..., 1], -points[..., 0], points[..., 2]], dim=-1)


def depth_to_lidar_boxes(boxes):
    """(…,7) depth boxes (x,y,z_bottom,dx,dy,dz,yaw) -> LiDAR (y,-x,z,dy,dx,dz,yaw):
    xyz @ [[0,1,0],[-1,0,0],[0,0,1]]^T and sizes swapped (box_3d_mode.py:124-127,134-143)."""
    return torch.stack([boxes[..., 1], -boxes[..., 0], boxes[..., 2], boxes[..., 4],
                        boxes[..., 3], boxes[..., 5], boxes[..., 6]], dim=-1)


class DepthInstance3DBoxes:
    """(T,7) boxes in the depth frame, (x, y, z_bottom, dx, dy, dz, yaw)."""

    def __init__(self, tensor, box_dim=7, with_yaw=True, origin=(0.5, 0.5, 0)):
        tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((0, box_dim))
        assert tensor.dim() == 2 and tensor.size(-1) == box_dim
        if box_dim == 6:
            tensor = torch.cat([tensor, tensor.new_zeros(tensor.shape[0], 1)], dim=-1)
        self.tensor = tensor.clone()
        if origin != (0.5, 0.5, 0):
            dst = self.tensor.new_tensor((0.5, 0.5, 0))
            src = self.tensor.new_tensor(origin)
            self.tensor[:, :3] += self.tensor[:, 3:6] * (dst - src)

    def __len__(self):
        return self.tensor.shape[0]

    def to(self, device):
        out = DepthInstance3DBoxes.__new__(DepthInstance3DBoxes)
        out.tensor = self.tensor.to(device)
        return out

    def new_box(self, data):
        return DepthInstance3DBoxes(torch.as_tensor(data, dtype=torch.float32,
                                                    device=self.tensor.device))

    @property
    def bottom_center(self):
        return self.tensor[:, :3]

    @property
    def dims(self):
        return self.tensor[:, 3:6]

    @property
    def gravity_center(self):
        bc = self.tensor[:, :3]
        return torch.cat([bc[:, :2], (bc[:, 2] + self.tensor[:, 5] * 0.5).unsqueeze(1)], dim=1)

    @property
    def volume(self):
        return self.tensor[:, 3] * self.tensor[:, 4] * self.tensor[:, 5]

    @property
    def bottom_height(self):
        return self.tensor[:, 2]

    @property
    def top_height(self):
        return self.bottom_height + self.tensor[:, 5]

    @property
    def bev(self):
        """(n,5) XYWHR (depth_box3d.py:92-95)."""
        return self.tensor[:, [0, 1, 3, 4, 6]]

    @property
    def corners(self):
        """(n,8,3) corners, order (x0y0z0, x0y0z1, x0y1z1, x0y1z0, x1y0z0, x1y0z1, x1y1z1,
        x1y1z0) about the relative origin (0.5, 0.5, 0), rotated about z by
        rotation_3d_in_axis (utils.py:36-61: x' = c x + s y, y' = -s x + c y), then shifted
        to the bottom centre (depth_box3d.py:50-89)."""
        assert len(self.tensor) != 0
        t = self.tensor
        # corner i = (x, y, z) bits (i>>2, (i>>1)&1, (i&1)^y): the order above; built with device
        # arithmetic (no host-to-device copy, so the property can run under graph capture)
        i = torch.arange(8, device=t.device)
        yb = (i >> 1) & 1
        norm = torch.stack([(i >> 2).to(t.dtype) - 0.5, yb.to(t.dtype) - 0.5,
                            ((i & 1) ^ yb).to(t.dtype)], dim=1)
        c = t[:, 3:6].view(-1, 1, 3) * norm.view(1, 8, 3)
        sin, cos = torch.sin(t[:, 6]).view(-1, 1), torch.cos(t[:, 6]).view(-1, 1)
        x, y = c[..., 0], c[..., 1]
        out = torch.stack([x * cos + y * sin, x * (-sin) + y * cos, c[..., 2]], dim=-1)
        return out + t[:, :3].view(-1, 1, 3)

    def __getitem__(self, item):
        out = DepthInstance3DBoxes.__new__(DepthInstance3DBoxes)
        if isinstance(item, int):
            out.tensor = self.tensor[item].view(1, -1)
        else:
            out.tensor = self.tensor[item]
            assert out.tensor.dim() == 2, f'Indexing on Boxes with {item} failed to return a matrix!'
        return out

    def convert_to(self, dst, rt_mat=None):
        """Box3DMode.DEPTH -> DEPTH is the identity (box_3d_mode.py:78-79); other modes are
        outside the indoor path."""
        if dst not in (None, 'DEPTH', 2) and getattr(dst, 'name', None) != 'DEPTH':
            raise NotImplementedError(f'only the depth frame is built, got {dst!r}')
        return self

    @classmethod
    def height_overlaps(cls, boxes1, boxes2):
        top = torch.min(boxes1.top_height.view(-1, 1), boxes2.top_height.view(1, -1))
        bottom = torch.max(boxes1.bottom_height.view(-1, 1), boxes2.bottom_height.view(1, -1))
        return torch.clamp(top - bottom, min=0)

    @classmethod
    def overlaps(cls, boxes1, boxes2, mode='iou'):
        """3-D IoU / IoF of every pair (base_box3d.py:387-438): rotated BEV overlap
        (ops/iou3d) x height overlap over the union volume.  The reference moves the operands
        to the GPU for the BEV op and back; here the result stays where the HIP kernel left
        it unless the inputs live on the host."""
        assert mode in ('iou', 'iof')
        rows, cols = len(boxes1), len(boxes2)
        if rows * cols == 0:
            return boxes1.tensor.new_empty(rows, cols)
        src = boxes1.tensor.device
        # host boxes (evaluation holds them on the CPU) visit the GPU for the BEV op, as in
        # the reference; an injected test back end serves them where they are
        dev = src if src.type == 'cuda' or injected_backend() is not None else torch.device('cuda')
        a, b = boxes1.to(dev), boxes2.to(dev)

        def xyxyr(bev):  # xywhr2xyxyr, structures/utils.py:64-82
            half_w, half_h = bev[:, 2] / 2, bev[:, 3] / 2
            return torch.stack([bev[:, 0] - half_w, bev[:, 1] - half_h, bev[:, 0] + half_w,
                                bev[:, 1] + half_h, bev[:, 4]], dim=1)
        overlaps_3d = boxes_overlap_bev(xyxyr(a.bev), xyxyr(b.bev)) * cls.height_overlaps(a, b)
        v1, v2 = a.volume.view(-1, 1), b.volume.view(1, -1)
        if mode == 'iou':
            iou = overlaps_3d / torch.clamp(v1 + v2 - overlaps_3d, min=1e-8)
        else:
            iou = overlaps_3d / torch.clamp(v1, min=1e-8)
        return iou.to(src)

    def points_in_boxes(self, points):
        """(M,3+) depth-frame points -> (M,T) int32 membership."""
        pts = depth_to_lidar_points(points[..., :3]).unsqueeze(0).contiguous()
        bx = depth_to_lidar_boxes(self.tensor.to(points.device)).unsqueeze(0).contiguous()
        return points_in_boxes_batch(pts, bx).squeeze(0)
