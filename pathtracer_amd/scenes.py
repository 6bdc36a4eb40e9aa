"""Seeded synthetic inputs for the path-tracing hot path.

Every asset of the reference is absent from /root/reference (assets/sky_15.raw is a missing
large blob, models/ is empty), so all scenes are generated here.  The constant parts follow
the reference's own scene assembly:
  * default_scene()      — the 7 spheres of src/scene.rs:98-160
  * wrap_mesh()          — mesh + light triangle + two floor triangles, src/scene.rs:35-96
  * LIGHT_COLOR          — src/material.rs:20-27
Everything returns float32 / uint32 numpy arrays laid out as include/ptcore.h expects.
"""
from __future__ import annotations

import numpy as np

MATERIAL_DTYPE = np.dtype([("kind", "<u4"), ("a", "<f4"), ("b", "<f4"), ("rgb", "<f4", (3,))])
MAT_DIFFUSE, MAT_DIELECTRIC, MAT_EMISSIVE = 0, 1, 2

# src/material.rs:20-27 (computed in f32 like the Rust constant expression)
LIGHT_COLOR = (
    float(np.float32(8.5) * np.float32(0.2)),
    float(np.float32(8.5) * np.float32(0.2)),
    float(np.float32(7.0) * np.float32(0.2)),
)

SKY_W, SKY_H = 2500, 1250  # src/scene.rs:163-165

# camera defaults, src/camera.rs:94-99
CAMERA_ORIGIN = (-1.6, 0.0, -1.3)
CAMERA_TARGET = (0.7, 0.0, 0.6)
LENS_SIZE = 0.04


def materials(rows) -> np.ndarray:
    """rows: iterable of (kind, a, b, (r, g, b))."""
    out = np.zeros(len(rows), dtype=MATERIAL_DTYPE)
    for i, (kind, a, b, rgb) in enumerate(rows):
        out[i] = (kind, a, b, rgb)
    return out


def diffuse(speculaty, rgb):
    return (MAT_DIFFUSE, speculaty, 0.0, rgb)


def dielectric(n1, n2, rgb):
    return (MAT_DIELECTRIC, n1, n2, rgb)


def emissive(rgb=LIGHT_COLOR):
    return (MAT_EMISSIVE, 0.0, 0.0, rgb)


# --------------------------------------------------------------------------------- sky
def make_sky(seed: int = 15, w: int = SKY_W, h: int = SKY_H) -> np.ndarray:
    """Procedural stand-in for assets/sky_15.raw: (h, w, 3) float32, values in [0, 16].

    Vertical gradient + bright horizon band + a few Gaussian 'suns'; row v = 0 is straight up
    (v = acos(d.y)/pi, src/scene.rs:164).
    """
    rng = np.random.default_rng(seed)
    v = (np.arange(h, dtype=np.float64) + 0.5) / h  # 0 = zenith, 1 = nadir
    u = (np.arange(w, dtype=np.float64) + 0.5) / w
    zen = np.array([0.25, 0.45, 0.95])
    hor = np.array([1.10, 1.00, 0.85])
    gnd = np.array([0.18, 0.16, 0.14])
    t_up = np.clip(v / 0.5, 0, 1)[:, None]
    t_dn = np.clip((v - 0.5) / 0.5, 0, 1)[:, None]
    col = np.where(v[:, None] < 0.5, zen + (hor - zen) * t_up ** 2.0, hor + (gnd - hor) * np.sqrt(t_dn))
    sky = np.repeat(col[:, None, :], w, axis=1)
    band = 0.6 * np.exp(-((v - 0.5) / 0.03) ** 2)
    sky += band[:, None, None] * np.array([1.0, 0.9, 0.7])
    for _ in range(4):
        cu, cv = rng.uniform(0, 1), rng.uniform(0.12, 0.42)
        sig = rng.uniform(0.008, 0.03)
        amp = rng.uniform(4.0, 14.0)
        tint = np.array([1.0, rng.uniform(0.8, 1.0), rng.uniform(0.55, 0.95)])
        du = np.minimum(np.abs(u - cu), 1 - np.abs(u - cu))
        g = np.exp(-(du[None, :] ** 2 + ((v - cv)[:, None] * 0.5) ** 2) / (2 * sig * sig))
        sky += amp * g[:, :, None] * tint
    return np.clip(sky, 0.0, 16.0).astype(np.float32)


# --------------------------------------------------------------------------------- spheres
def default_scene():
    """Scene::default_scene, src/scene.rs:98-160.  Returns (cr[N,4], mat_id[N], materials)."""
    mats = materials([
        emissive(),                            # 0  Sphere::light
        diffuse(0.8, (1.0, 0.2, 0.2)),         # 1
        dielectric(1.0, 1.3, (0.1, 1.0, 0.1)),  # 2
        diffuse(0.8, (0.2, 0.2, 1.0)),         # 3
        diffuse(0.0, (1.0, 1.0, 1.0)),         # 4
    ])
    cr = np.array([
        [2.7, 1.7, -0.5, 0.3],
        [-0.8, 0.0, -2.0, 0.3],
        [0.0, 0.0, -2.0, 0.3],
        [0.8, 0.0, -2.0, 0.3],
        [-0.8, -0.8, -2.0, 0.5],
        [-0.0, -0.8, -2.0, 0.5],
        [0.8, -0.8, -2.0, 0.5],
    ], dtype=np.float32)
    mat_id = np.array([0, 1, 2, 3, 4, 4, 4], dtype=np.uint32)
    return cr, mat_id, mats


# camera that actually looks at the spheres (the reference default looks away from z = -2)
SPHERE_CAMERA_ORIGIN = (0.0, 0.2, 1.0)
SPHERE_CAMERA_TARGET = (0.0, -0.2, -2.0)


# --------------------------------------------------------------------------------- triangle scenes
def wrap_mesh(pos, nrm, mat_id, mats_rows, floor_speculaty=0.5):
    """Scene::scene, src/scene.rs:35-96: mesh triangles, then the light, then the two floor
    triangles.  mats_rows are the mesh's materials; light and floor materials are appended."""
    rows = list(mats_rows)
    light_id = len(rows)
    rows.append(emissive())
    floor_id = len(rows)
    rows.append(diffuse(floor_speculaty, (0.9, 0.9, 0.9)))
    extra_pos = np.array([
        [2.0, 2.0, 2.0, 1.0, 2.0, 2.0, 2.0, 2.0, 1.0],
        [200.0, -0.3, 200.0, 200.0, -0.3, -200.0, -200.0, -0.3, 200.0],
        [-200.0, -0.3, -200.0, -200.0, -0.3, 200.0, 200.0, -0.3, -200.0],
    ], dtype=np.float32)
    extra_nrm = np.array([
        [0, -1, 0] * 3,
        [0, 1, 0] * 3,
        [0, 1, 0] * 3,
    ], dtype=np.float32)
    pos = np.concatenate([np.asarray(pos, np.float32).reshape(-1, 9), extra_pos])
    nrm = np.concatenate([np.asarray(nrm, np.float32).reshape(-1, 9), extra_nrm])
    mat_id = np.concatenate([np.asarray(mat_id, np.uint32), np.array([light_id, floor_id, floor_id], np.uint32)])
    return np.ascontiguousarray(pos), np.ascontiguousarray(nrm), np.ascontiguousarray(mat_id), materials(rows)


# backup glass of Scene::scene (src/scene.rs:36-40) and the green glass of the cube screenshots
GLASS_BLUE = dielectric(1.0, 1.5, (0.01, 0.01, 0.9))
GLASS_GREEN = dielectric(1.0, 1.5, (0.01, 0.9, 0.01))


def _flat_normals(pos):
    p = pos.reshape(-1, 3, 3).astype(np.float64)
    n = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    return np.repeat(n[:, None, :], 3, axis=1).reshape(-1, 9).astype(np.float32)


def cube_mesh(centre_xz=(0.7, 0.6), side=0.6, base_y=-0.3, lift=0.0):
    """12 outward-wound triangles of an axis-aligned cube standing on y = base_y + lift."""
    h = side / 2
    cx, cz = centre_xz
    y0, y1 = base_y + lift, base_y + lift + side
    v = np.array([[cx - h, y0, cz - h], [cx + h, y0, cz - h], [cx + h, y1, cz - h], [cx - h, y1, cz - h],
                  [cx - h, y0, cz + h], [cx + h, y0, cz + h], [cx + h, y1, cz + h], [cx - h, y1, cz + h]])
    quads = [(0, 3, 2, 1), (4, 5, 6, 7), (0, 4, 7, 3), (1, 2, 6, 5), (3, 7, 6, 2), (0, 1, 5, 4)]
    tris = []
    for a, b, c, d in quads:
        tris += [(a, b, c), (a, c, d)]
    pos = v[np.array(tris)].reshape(-1, 9).astype(np.float32)
    ctr = np.array([cx, (y0 + y1) / 2, cz])
    p = pos.reshape(-1, 3, 3).astype(np.float64)
    n = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    assert np.all(np.einsum("ij,ij->i", n, p.mean(axis=1) - ctr) > 0), "cube winding must face outward"
    return pos, _flat_normals(pos)


def glass_cube_scene(floor_speculaty=0.0, lift=0.0):
    """C2: green glass cube on the diffuse floor under the reference light."""
    pos, nrm = cube_mesh(lift=lift)
    return wrap_mesh(pos, nrm, np.zeros(len(pos), np.uint32), [GLASS_GREEN], floor_speculaty)


def statue_mesh(n_u=935, n_v=465, k_ped=466, seed=7, height=1.0, centre_xz=(0.7, 0.6), base_y=-0.3):
    """Closed 'statue': a displaced (2,3) torus-knot tube with smooth vertex normals standing on
    a coin-shaped pedestal.  Triangle count = 2*n_u*n_v + 4*k_ped.
    Defaults give 871 414 triangles (Stanford-dragon count); (1042, 520, 1009) gives 1 087 716
    (Happy-Buddha count)."""
    rng = np.random.default_rng(seed)
    p_, q_ = 2, 3
    R, r, rho0 = 1.0, 0.42, 0.20
    t = np.arange(n_u, dtype=np.float64) * (2 * np.pi / n_u)
    phi = np.arange(n_v, dtype=np.float64) * (2 * np.pi / n_v)

    def curve(tt):
        w = R + r * np.cos(q_ * tt)
        return np.stack([w * np.cos(p_ * tt), r * np.sin(q_ * tt), w * np.sin(p_ * tt)], axis=-1)

    h = 1e-4
    C = curve(t)
    d1 = (curve(t + h) - curve(t - h)) / (2 * h)
    d2 = (curve(t + h) - 2 * C + curve(t - h)) / (h * h)
    T = d1 / np.linalg.norm(d1, axis=1, keepdims=True)
    N = d2 - np.einsum("ij,ij->i", d2, T)[:, None] * T
    N /= np.linalg.norm(N, axis=1, keepdims=True)
    B = np.cross(T, N)
    # multi-octave periodic displacement of the tube radius
    tt, pp = np.meshgrid(t, phi, indexing="ij")
    disp = np.zeros_like(tt)
    amp = 0.22
    for octave in range(6):
        f_t = int(rng.integers(3, 9)) * (2 ** octave)
        f_p = int(rng.integers(1, 4)) * (2 ** min(octave, 4))
        ph = rng.uniform(0, 2 * np.pi)
        disp += amp * np.sin(f_t * tt + f_p * pp + ph)
        amp *= 0.55
    rho = rho0 * (1.0 + disp)
    S = C[:, None, :] + rho[:, :, None] * (np.cos(pp)[:, :, None] * N[:, None, :] + np.sin(pp)[:, :, None] * B[:, None, :])
    # stand the ring up: torus axis horizontal
    S = S[:, :, [2, 0, 1]]
    Cc = C[:, [2, 0, 1]]
    du = np.roll(S, -1, axis=0) - np.roll(S, 1, axis=0)
    dv = np.roll(S, -1, axis=1) - np.roll(S, 1, axis=1)
    nrm = np.cross(du, dv)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=2, keepdims=True), 1e-30)
    if np.mean(np.einsum("ijk,ijk->ij", nrm, S - Cc[:, None, :])) < 0:
        nrm = -nrm
    # normalise into the requested box
    ped_h = 0.05 * height
    ymin, ymax = S[:, :, 1].min(), S[:, :, 1].max()
    scale = (height - ped_h) / (ymax - ymin)
    mid = 0.5 * (S.reshape(-1, 3).min(axis=0) + S.reshape(-1, 3).max(axis=0))
    S = (S - mid) * scale
    S[:, :, 0] += centre_xz[0]
    S[:, :, 2] += centre_xz[1]
    S[:, :, 1] += base_y + ped_h - S[:, :, 1].min()
    i0 = np.arange(n_u)[:, None]
    j0 = np.arange(n_v)[None, :]
    i1 = (i0 + 1) % n_u
    j1 = (j0 + 1) % n_v

    def vid(i, j):
        return (i * n_v + j).astype(np.int64)

    v00, v10, v11, v01 = vid(i0, j0), vid(i1, j0), vid(i1, j1), vid(i0, j1)
    shape = (n_u, n_v)
    v00, v10, v11, v01 = (np.broadcast_to(a, shape) for a in (v00, v10, v11, v01))
    tri = np.stack([np.stack([v00, v10, v11], -1), np.stack([v00, v11, v01], -1)], axis=2).reshape(-1, 3)
    V = S.reshape(-1, 3)
    VN = nrm.reshape(-1, 3)
    # winding must agree with the smooth normals (front face <=> inside == false)
    g = np.cross(V[tri[:, 1]] - V[tri[:, 0]], V[tri[:, 2]] - V[tri[:, 0]])
    if np.mean(np.einsum("ij,ij->i", g, VN[tri[:, 0]])) < 0:
        tri = tri[:, [0, 2, 1]]
    pos = V[tri].reshape(-1, 9)
    nor = VN[tri].reshape(-1, 9)
    # pedestal: closed coin, k-gon fans top and bottom + 2k side triangles
    if k_ped > 0:
        k = k_ped
        a = np.arange(k) * (2 * np.pi / k)
        rad = 0.45 * height
        ring = np.stack([centre_xz[0] + rad * np.cos(a), np.zeros(k), centre_xz[1] + rad * np.sin(a)], axis=1)
        top, bot = ring.copy(), ring.copy()
        top[:, 1] = base_y + ped_h
        bot[:, 1] = base_y + 1e-3
        ct = np.array([centre_xz[0], base_y + ped_h, centre_xz[1]])
        cb = np.array([centre_xz[0], base_y + 1e-3, centre_xz[1]])
        nxt = (np.arange(k) + 1) % k
        ped = []
        ped.append(np.stack([np.broadcast_to(ct, (k, 3)), top[nxt], top], axis=1))          # top fan, +y
        ped.append(np.stack([np.broadcast_to(cb, (k, 3)), bot, bot[nxt]], axis=1))          # bottom fan, -y
        ped.append(np.stack([top, top[nxt], bot[nxt]], axis=1))                              # side
        ped.append(np.stack([top, bot[nxt], bot], axis=1))
        ped = np.concatenate(ped).reshape(-1, 9)
        pos = np.concatenate([pos, ped])
        nor = np.concatenate([nor, _flat_normals(ped.astype(np.float32)).astype(np.float64)])
    return np.ascontiguousarray(pos, dtype=np.float32), np.ascontiguousarray(nor, dtype=np.float32)


def statue_scene(n_u=935, n_v=465, k_ped=466, seed=7, glass=GLASS_GREEN, floor_speculaty=0.0):
    """C3 (defaults, 871 414 + 3 triangles) / C4 (n_u=1042, n_v=520, k_ped=1009, blue glass, floor 0.5)."""
    pos, nrm = statue_mesh(n_u, n_v, k_ped, seed)
    return wrap_mesh(pos, nrm, np.zeros(len(pos), np.uint32), [glass], floor_speculaty)


def soup_scene(n=10_000_000, seed=5, floor_speculaty=0.0):
    """C5: n independent triangles, centres uniform in [-2,2]x[-0.3,1.7]x[-2,2], edge length
    log-uniform in [0.002, 0.02], random orientation, flat normals; 70 % diffuse (64-colour
    palette, albedo U[.2,.9]), 20 % glass n2 = 1.5, 10 % specular (speculaty 0.8)."""
    rng = np.random.default_rng(seed)
    ctr = rng.uniform([-2, -0.3, -2], [2, 1.7, 2], size=(n, 3))
    edge = np.exp(rng.uniform(np.log(0.002), np.log(0.02), size=n))
    a = rng.normal(size=(n, 3))
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = rng.normal(size=(n, 3))
    b -= np.einsum("ij,ij->i", a, b)[:, None] * a
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    p0 = ctr - (edge / 2)[:, None] * a - (edge * 0.2887)[:, None] * b
    p1 = ctr + (edge / 2)[:, None] * a - (edge * 0.2887)[:, None] * b
    p2 = ctr + (edge * 0.5774)[:, None] * b
    pos = np.concatenate([p0, p1, p2], axis=1).astype(np.float32)
    del ctr, a, b, p0, p1, p2
    nrm = _flat_normals(pos)
    rows = [diffuse(0.0, tuple(rng.uniform(0.2, 0.9, size=3))) for _ in range(64)]
    rows.append(dielectric(1.0, 1.5, (0.6, 0.9, 0.7)))
    rows.append(diffuse(0.8, (0.8, 0.8, 0.8)))
    kind = rng.uniform(size=n)
    mat = rng.integers(0, 64, size=n).astype(np.uint32)
    mat[kind > 0.7] = 64
    mat[kind > 0.9] = 65
    return wrap_mesh(pos, nrm, mat, rows, floor_speculaty)


def small_mesh_scene(n_u=24, n_v=12, k_ped=8, seed=3, floor_speculaty=0.5):
    """A few-hundred-triangle statue for fast parity tests."""
    return statue_scene(n_u, n_v, k_ped, seed, GLASS_BLUE, floor_speculaty)


def random_rays(n, seed=1, box=((-2.5, -0.3, -2.5), (2.5, 2.5, 2.5)), aim=(0.7, 0.2, 0.6), aim_sigma=0.6):
    """Ray batch for the traversal seam: origins uniform in `box`, directions towards a
    Gaussian blob around `aim` (so most rays meet geometry), tmax = +inf.  (n, 7) float32."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(box[0], box[1], size=(n, 3))
    tgt = np.asarray(aim) + rng.normal(scale=aim_sigma, size=(n, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.empty((n, 7), dtype=np.float32)
    rays[:, 0:3] = o
    rays[:, 3:6] = d
    rays[:, 6] = np.inf
    return rays
