"""Seeded synthetic ScanNet-shaped scenes: there is no dataset
offline, so benchmarks and parity tests use rooms generated here.

A scene is an 8 x 8 x 3 m room: floor, four walls and K in [3, 20] axis-aligned
furniture boxes standing on the floor.  P in U{25000..80000} surface points are
sampled area-uniformly with 5 mm Gaussian noise, then resampled to exactly
``num_points`` the way ``IndoorPointSample`` does
(transforms_3d.py:821,857: with replacement iff P < num_points, so exact duplicate
points are normal), and the 4th channel is the height above the 0.99th-percentile
floor (loading.py:418-420).  GT boxes are depth-frame (x, y, z_bottom, dx, dy, dz, 0).
"""
import numpy as np
import torch


def _box_surface_points(rng, centre, size, n):
    """n points uniform on the 5 visible faces (no bottom) of an axis-aligned box."""
    dx, dy, dz = size
    areas = np.array([dx * dy, dx * dz, dx * dz, dy * dz, dy * dz])
    face = rng.choice(5, size=n, p=areas / areas.sum())
    u, v = rng.random(n) - 0.5, rng.random(n) - 0.5
    p = np.zeros((n, 3))
    top = face == 0
    p[top] = np.stack([u[top] * dx, v[top] * dy, np.full(top.sum(), dz / 2)], 1)
    for f, sgn in ((1, -1.0), (2, 1.0)):
        m = face == f
        p[m] = np.stack([u[m] * dx, np.full(m.sum(), sgn * dy / 2), v[m] * dz], 1)
    for f, sgn in ((3, -1.0), (4, 1.0)):
        m = face == f
        p[m] = np.stack([np.full(m.sum(), sgn * dx / 2), u[m] * dy, v[m] * dz], 1)
    return p + centre


def make_scene(seed, num_points=40000, num_classes=18, room=(8.0, 8.0, 3.0)):
    """-> points (num_points,4) f32, gt_boxes (K,7) f32, gt_labels (K,) i64."""
    rng = np.random.default_rng(seed)
    rx, ry, rz = room
    K = int(rng.integers(3, 21))
    sizes = np.stack([0.4 + rng.random(K) * 1.6, 0.4 + rng.random(K) * 1.6,
                      0.4 + rng.random(K) * 1.4], 1)
    cxy = np.stack([(rng.random(K) - 0.5) * (rx - 2.0), (rng.random(K) - 0.5) * (ry - 2.0)], 1)
    labels = rng.integers(0, num_classes, K)
    P = int(rng.integers(25000, 80001))
    # area-proportional split between floor, walls and furniture
    box_area = (sizes[:, 0] * sizes[:, 1] + 2 * sizes[:, 2] * (sizes[:, 0] + sizes[:, 1]))
    parts = np.concatenate([[rx * ry], [rx * rz, rx * rz, ry * rz, ry * rz], box_area * 3.0])
    counts = rng.multinomial(P, parts / parts.sum())
    pts = [np.stack([(rng.random(counts[0]) - 0.5) * rx, (rng.random(counts[0]) - 0.5) * ry,
                     np.zeros(counts[0])], 1)]
    for w, (axis, sgn) in enumerate(((1, -1), (1, 1), (0, -1), (0, 1))):
        n = counts[1 + w]
        a = (rng.random(n) - 0.5) * (rx if axis == 1 else ry)
        z = rng.random(n) * rz
        fixed = np.full(n, sgn * (ry if axis == 1 else rx) / 2)
        pts.append(np.stack([a, fixed, z], 1) if axis == 1 else np.stack([fixed, a, z], 1))
    for k in range(K):
        c = np.array([cxy[k, 0], cxy[k, 1], sizes[k, 2] / 2])
        pts.append(_box_surface_points(rng, c, sizes[k], counts[5 + k]))
    pts = np.concatenate(pts, 0) + rng.normal(0, 0.005, (P, 3))
    choice = rng.choice(P, num_points, replace=P < num_points)
    pts = pts[choice]
    floor = np.percentile(pts[:, 2], 0.99)
    points = np.concatenate([pts, (pts[:, 2] - floor)[:, None]], 1).astype(np.float32)
    boxes = np.concatenate([cxy, np.zeros((K, 1)), sizes, np.zeros((K, 1))], 1).astype(np.float32)
    return (torch.from_numpy(points), torch.from_numpy(boxes),
            torch.from_numpy(labels.astype(np.int64)))


def make_batch(first_seed, batch, num_points=40000, num_classes=18):
    """-> points (B,N,4), list of (K_i,7) boxes, list of (K_i,) labels."""
    scenes = [make_scene(first_seed + i, num_points, num_classes) for i in range(batch)]
    return (torch.stack([s[0] for s in scenes]), [s[1] for s in scenes],
            [s[2] for s in scenes])
