"""The slice of ``mmdet3d/core/bbox/structures/depth_box3d.py`` the training step
touches: gravity centres (:42-48) and ``points_in_boxes`` (:251-277) with the
depth->LiDAR frame change of ``box_3d_mode.py:124-127``."""
import torch

from ..mmdet3d_ops import points_in_boxes_batch


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

    def points_in_boxes(self, points):
        """(M,3+) depth-frame points -> (M,T) int32 membership."""
        pts = depth_to_lidar_points(points[..., :3]).unsqueeze(0).contiguous()
        bx = depth_to_lidar_boxes(self.tensor.to(points.device)).unsqueeze(0).contiguous()
        return points_in_boxes_batch(pts, bx).squeeze(0)
