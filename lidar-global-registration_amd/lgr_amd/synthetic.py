"""Synthetic scan pairs (SURVEY.md section 8d / BASELINE.md): height-field + boxes scene, random SE(3), noise.

The reference ships no point clouds (its .gitignore excludes them), so every config runs on this generator.
numpy only -- this module is used by tests, bench.py and the oracle-side baseline alike.
"""
import numpy as np

SEED = 566  # SEED of include/common.h:25


def make_points(xyz, intensity=1.0):
    """n x 12 float32 in pcl::PointXYZINormal layout {x,y,z,1 | nx,ny,nz,0 | intensity,curvature,pad,pad}."""
    xyz = np.asarray(xyz, dtype=np.float32)
    p = np.zeros((xyz.shape[0], 12), dtype=np.float32)
    p[:, 0:3] = xyz
    p[:, 3] = 1.0
    p[:, 8] = intensity
    return p


class Scene:
    """z = sum_k A_k sin(a_k x + b_k y + phi_k) over [0, 24 s] x [0, 16 s] plus axis-aligned boxes resting on it."""

    def __init__(self, rng, scale=1.0, n_waves=12, n_boxes=40):
        self.scale = float(scale)
        self.lx, self.ly = 24.0 * self.scale, 16.0 * self.scale
        self.A = rng.uniform(0.05, 0.6, n_waves) * min(1.0, self.scale * 2.0)
        wl = rng.uniform(0.8, 6.0, n_waves) * min(1.0, max(self.scale, 0.25))
        ang = rng.uniform(0, 2 * np.pi, n_waves)
        self.a = 2 * np.pi / wl * np.cos(ang)
        self.b = 2 * np.pi / wl * np.sin(ang)
        self.phi = rng.uniform(0, 2 * np.pi, n_waves)
        nb = n_boxes
        size = rng.uniform(0.5, 2.0, (nb, 3)) * min(1.0, max(self.scale, 0.25))
        cx = rng.uniform(0, self.lx, nb)
        cy = rng.uniform(0, self.ly, nb)
        self.box = np.stack([cx - size[:, 0] / 2, cx + size[:, 0] / 2, cy - size[:, 1] / 2, cy + size[:, 1] / 2], 1)
        self.box_h = size[:, 2]

    def height(self, x, y):
        z = np.zeros_like(x)
        for k in range(len(self.A)):
            z += self.A[k] * np.sin(self.a[k] * x + self.b[k] * y + self.phi[k])
        for i in range(self.box.shape[0]):
            x0, x1, y0, y1 = self.box[i]
            inside = (x >= x0) & (x <= x1) & (y >= y0) & (y <= y1)
            if inside.any():
                base = 0.0
                for k in range(len(self.A)):
                    base += self.A[k] * np.sin(self.a[k] * 0.5 * (x0 + x1) + self.b[k] * 0.5 * (y0 + y1) + self.phi[k])
                z = np.where(inside, np.maximum(z, base + self.box_h[i]), z)
        return z

    def sample(self, rng, n, x_lo, x_hi):
        x = rng.uniform(x_lo * self.lx, x_hi * self.lx, n)
        y = rng.uniform(0, self.ly, n)
        return np.stack([x, y, self.height(x, y)], 1)


def random_se3(rng, t_range=5.0):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = rng.uniform(-t_range, t_range, 3)
    return T


def make_pair(n_points=1_000_000, seed=SEED, noise=0.005, overlap=0.5, constant_density=True):
    """Returns dict(src, tgt [n x 12 float32], T_gt [4x4: maps src frame -> tgt frame], vp_src, vp_tgt).

    src covers x in [0, 2/3] of the scene, tgt covers [1/3, 1] (50 % overlap), both in the scene frame; tgt is then
    moved by a random rigid T (so T_gt = T).  With constant_density the scene shrinks with sqrt(n / 1e6) so the
    point density (and therefore the FPFH neighbourhood size at r = 0.25 m) is that of the 1M-point configuration.
    """
    rng = np.random.default_rng(seed)
    scale = np.sqrt(n_points / 1.0e6) if constant_density else 1.0
    scene = Scene(rng, scale)
    lo_t = (1.0 - overlap) * (2.0 / 3.0)
    src = scene.sample(rng, n_points, 0.0, 2.0 / 3.0)
    tgt = scene.sample(rng, n_points, lo_t, lo_t + 2.0 / 3.0)
    src += rng.normal(0, noise, src.shape)
    tgt += rng.normal(0, noise, tgt.shape)
    T = random_se3(rng)
    vp_scene = np.array([0.0, 0.0, 10.0])
    tgt_m = tgt @ T[:3, :3].T + T[:3, 3]
    vp_tgt = T[:3, :3] @ vp_scene + T[:3, 3]
    return dict(src=make_points(src), tgt=make_points(tgt_m), T_gt=T.astype(np.float64),
                vp_src=vp_scene.astype(np.float32), vp_tgt=vp_tgt.astype(np.float32), scale=scale)


def make_correspondence_problem(n_pts=20000, c=5000, inlier_frac=0.4, sigma=0.01, thr=0.05, seed=SEED, extent=10.0):
    """Directly synthesised correspondences for RANSAC stress (BASELINE config 4):
    inlier_frac of the c correspondences are true (t = T s + N(0, sigma)), the rest point at random targets."""
    rng = np.random.default_rng(seed)
    src = rng.uniform(-extent, extent, (n_pts, 3))
    src[:, 2] *= 0.2
    T = random_se3(rng)
    tgt = src @ T[:3, :3].T + T[:3, 3] + rng.normal(0, sigma, src.shape)
    perm = rng.permutation(n_pts)[:c]
    match = perm.copy()
    n_out = c - int(round(inlier_frac * c))
    out_idx = rng.permutation(c)[:n_out]
    match[out_idx] = rng.integers(0, n_pts, n_out)
    corr = np.zeros(c, dtype=[("index_query", "<i4"), ("index_match", "<i4"), ("distance", "<f4"), ("threshold", "<f4")])
    corr["index_query"] = perm
    corr["index_match"] = match
    corr["distance"] = rng.uniform(0, 50, c).astype(np.float32)
    corr["threshold"] = thr
    return dict(src=make_points(src), tgt=make_points(tgt), corr=corr, T_gt=T)
