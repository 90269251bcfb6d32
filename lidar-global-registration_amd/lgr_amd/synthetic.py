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


class PlanarScene:
    """A man-made scene of PLANES -- the shape of the reference's real configs (data/tests.yaml: WHU-TLS SubwayStation / Residence / ..., kizhi, arch,
    office): a flat ground rectangle, `n_boxes` buildings (four walls and a roof each: 5 n_boxes + 1 rectangles), and a few Gaussian blobs of
    clutter ("vegetation").  plane_frac of the points lie exactly on rectangles (before noise), sampled uniformly in area."""

    def __init__(self, rng, scale=1.0, n_boxes=8, plane_frac=0.85):
        self.lx, self.ly = 16.0 * scale, 10.0 * scale
        self.plane_frac = plane_frac
        w = rng.uniform(1.5, 3.0, (n_boxes, 2)) * min(1.0, max(scale, 0.25))
        h = rng.uniform(1.5, 3.0, n_boxes) * min(1.0, max(scale, 0.25))
        cx = rng.uniform(0.1 * self.lx, 0.9 * self.lx, n_boxes)
        cy = rng.uniform(0.15 * self.ly, 0.85 * self.ly, n_boxes)
        rects = [(np.array([0.0, 0.0, 0.0]), np.array([self.lx, 0.0, 0.0]), np.array([0.0, self.ly, 0.0]))]   # origin, edge u, edge v
        for i in range(n_boxes):
            x0, x1, y0, y1 = cx[i] - w[i, 0] / 2, cx[i] + w[i, 0] / 2, cy[i] - w[i, 1] / 2, cy[i] + w[i, 1] / 2
            z = np.array([0.0, 0.0, h[i]])
            rects += [(np.array([x0, y0, 0.0]), np.array([x1 - x0, 0.0, 0.0]), z), (np.array([x0, y1, 0.0]), np.array([x1 - x0, 0.0, 0.0]), z),
                      (np.array([x0, y0, 0.0]), np.array([0.0, y1 - y0, 0.0]), z), (np.array([x1, y0, 0.0]), np.array([0.0, y1 - y0, 0.0]), z),
                      (np.array([x0, y0, h[i]]), np.array([x1 - x0, 0.0, 0.0]), np.array([0.0, y1 - y0, 0.0]))]
        self.rects = rects
        self.area = np.array([np.linalg.norm(np.cross(u, v)) for _, u, v in rects])
        self.blob_c = np.stack([rng.uniform(0, self.lx, 12), rng.uniform(0, self.ly, 12), rng.uniform(0.3, 1.5, 12)], 1)
        self.blob_s = rng.uniform(0.15, 0.5, 12) * min(1.0, max(scale, 0.25))

    def n_planes(self):
        return len(self.rects)

    def sample(self, rng, n, x_lo, x_hi):
        out = np.empty((0, 3))
        lo, hi = x_lo * self.lx, x_hi * self.lx
        while out.shape[0] < n:
            m = int((n - out.shape[0]) * 1.8) + 1024
            on_plane = rng.uniform(size=m) < self.plane_frac
            r = rng.choice(len(self.rects), m, p=self.area / self.area.sum())
            O = np.stack([self.rects[k][0] for k in range(len(self.rects))])[r]
            U = np.stack([self.rects[k][1] for k in range(len(self.rects))])[r]
            V = np.stack([self.rects[k][2] for k in range(len(self.rects))])[r]
            p = O + U * rng.uniform(size=(m, 1)) + V * rng.uniform(size=(m, 1))
            b = rng.integers(0, len(self.blob_s), m)
            q = self.blob_c[b] + rng.normal(size=(m, 3)) * self.blob_s[b, None]
            q[:, 2] = np.abs(q[:, 2])
            p = np.where(on_plane[:, None], p, q)
            p = p[(p[:, 0] >= lo) & (p[:, 0] <= hi)]
            out = np.concatenate([out, p])
        return out[:n]


def make_planar_pair(n_points=1_000_000, seed=SEED, noise=0.005, overlap=0.5, plane_frac=0.85, n_boxes=8):
    """make_pair's layout on a PlanarScene: >= 70 % of the points on >= 20 planes; returns the same dict + n_planes, plane_frac"""
    rng = np.random.default_rng(seed)
    scale = np.sqrt(n_points / 1.0e6)
    scene = PlanarScene(rng, scale, n_boxes=n_boxes, plane_frac=plane_frac)
    lo_t = (1.0 - overlap) * (2.0 / 3.0)
    src = scene.sample(rng, n_points, 0.0, 2.0 / 3.0)
    tgt = scene.sample(rng, n_points, lo_t, lo_t + 2.0 / 3.0)
    src += rng.normal(0, noise, src.shape)
    tgt += rng.normal(0, noise, tgt.shape)
    T = random_se3(rng)
    vp_scene = np.array([0.5 * scene.lx, 0.5 * scene.ly, 10.0])
    tgt_m = tgt @ T[:3, :3].T + T[:3, 3]
    vp_tgt = T[:3, :3] @ vp_scene + T[:3, 3]
    return dict(src=make_points(src), tgt=make_points(tgt_m), T_gt=T.astype(np.float64), vp_src=vp_scene.astype(np.float32), vp_tgt=vp_tgt.astype(np.float32),
                scale=scale, n_planes=scene.n_planes(), plane_frac=plane_frac)


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
