"""Seeded input generators shared by the CPU and GPU parity tests."""
import numpy as np
import torch


def cloud(seed, b, n, dup_frac=0.0, grid=False):
    """(b,n,3) float32 points in a 4x4x2.5 m box; optional exact duplicates / lattice."""
    rng = np.random.default_rng(seed)
    if grid:  # lattice: masses of exactly equal distances
        side = int(np.ceil(n ** (1 / 3)))
        g = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3)
        pts = np.stack([g[rng.permutation(len(g))[:n]] for _ in range(b)]).astype(np.float32)
        pts *= 0.25
    else:
        pts = (rng.random((b, n, 3)) * np.array([4.0, 4.0, 2.5])).astype(np.float32)
    if dup_frac > 0:
        k = int(n * dup_frac)
        for i in range(b):
            src = rng.integers(0, n, k)
            dst = rng.permutation(n)[:k]
            pts[i, dst] = pts[i, src]
    return torch.from_numpy(np.ascontiguousarray(pts))


def boxes_lidar(seed, b, t, yaw=True):
    rng = np.random.default_rng(seed)
    c = rng.random((b, t, 3)) * np.array([4.0, 4.0, 1.0])
    s = 0.3 + rng.random((b, t, 3)) * 1.5
    r = (rng.random((b, t, 1)) - 0.5) * (2 * np.pi if yaw else 0.0)
    return torch.from_numpy(np.concatenate([c, s, r], -1).astype(np.float32))


def box_pairs(seed, n, mode="random"):
    """(1,n,7) box pairs (x,y,z,dx,dy,dz,yaw) for the rotated-IoU chain."""
    rng = np.random.default_rng(seed)
    a = np.concatenate([rng.random((n, 3)) * 2, 0.5 + rng.random((n, 3)) * 1.5,
                        (rng.random((n, 1)) - 0.5) * np.pi], -1)
    if mode == "identical":
        b = a.copy()
    elif mode == "disjoint":
        b = a.copy(); b[:, :2] += 10.0
    elif mode == "aligned":
        a[:, 6] = 0; b = a.copy(); b[:, :3] += (rng.random((n, 3)) - 0.5) * 0.6
        b[:, 3:6] *= 0.7 + rng.random((n, 3)) * 0.6
    else:
        b = a.copy(); b[:, :3] += (rng.random((n, 3)) - 0.5) * 0.8
        b[:, 3:6] *= 0.7 + rng.random((n, 3)) * 0.6
        b[:, 6] += (rng.random(n) - 0.5) * 1.0
    return (torch.from_numpy(a.astype(np.float32))[None].contiguous(),
            torch.from_numpy(b.astype(np.float32))[None].contiguous())
