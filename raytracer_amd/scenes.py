"""Synthetic scenes, cameras and tile lists for the parity tests and the benchmark.

All generators draw from one integer PRNG (splitmix64 -> 24-bit floats) and
use only IEEE add/multiply afterwards, so the same seed yields bit-identical
float32 vertices on every machine.
"""
import math

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix_u64(seed, n, offset=0):
    """n draws of the splitmix64 stream seeded with `seed`, starting at `offset`."""
    with np.errstate(over="ignore"):
        i = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed) + i * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def splitmix_unit(seed, n, offset=0):
    """n float64 values in [0,1) with 24 random bits each."""
    return (splitmix_u64(seed, n, offset) >> np.uint64(40)).astype(np.float64) * (2.0 ** -24)


KHAKI = (0.8, 0.659341, 0.43956)


def soup(n, seed=0x5EED0001, extent=1.0, size=0.02):
    """Triangle soup: centroids ~U([-extent,extent]^3), vertices = centroid +
    offsets ~U([-size,size]^3).  One material, one OBJ group.  Returns dict for
    raytracer_amd.api.Mesh / oracle.Scene."""
    u = splitmix_unit(seed, n * 12).reshape(n, 12)
    c = (u[:, 0:3] * 2 - 1) * extent
    off = (u[:, 3:12] * 2 - 1) * size
    verts = (np.tile(c, (1, 3)) + off).astype(np.float32)
    return {"verts": verts, "material": np.zeros(n, np.int32),
            "group_index": np.arange(n, dtype=np.int32),
            "kd": np.array([KHAKI], np.float32), "is_light": np.zeros(1, np.uint8)}


def _quad(p0, p1, p2, p3):
    """Two triangles (p0,p1,p2), (p0,p2,p3); front face = counter-clockwise."""
    return [list(p0) + list(p1) + list(p2), list(p0) + list(p2) + list(p3)]


def room_mesh(grid, seed=0x5EED0002, cell=0.05, total=None):
    """Closed room + ceiling light quad + displaced grid x grid height field,
    padded with soup triangles up to `total` triangles.

    Geometry is scaled so that triangle areas stay far above the reference's
    scale-dependent determinant cut-off (triangle.go:83, det < 1e-4).
    Materials: 0 Khaki (room), 1 Light, 2 field, 3 soup.
    """
    half = grid * cell / 2.0            # half extent of the field
    wall = half * 1.15                  # room half width
    height = half * 0.9                 # room height
    tris = []
    mats = []
    # room, faces pointing inward
    x0, x1, y0, y1, z0, z1 = -wall, wall, 0.0, height, -wall, wall
    faces = [
        _quad((x0, y0, z1), (x1, y0, z1), (x1, y0, z0), (x0, y0, z0)),   # floor, +y
        _quad((x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1)),   # ceiling, -y
        _quad((x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0)),   # back, +z
        _quad((x1, y0, z1), (x0, y0, z1), (x0, y1, z1), (x1, y1, z1)),   # front, -z
        _quad((x0, y0, z1), (x0, y0, z0), (x0, y1, z0), (x0, y1, z1)),   # left, +x
        _quad((x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0)),   # right, -x
    ]
    for f in faces:
        tris += f
        mats += [0, 0]
    # light quad just below the ceiling, facing down
    ls, ly = half * 0.35, height * 0.98
    tris += _quad((-ls, ly, -ls), (ls, ly, -ls), (ls, ly, ls), (-ls, ly, ls))
    mats += [1, 1]
    base = np.array(tris, np.float64)

    # height field: coarse random lattice, bilinearly interpolated, plus jitter
    n = grid
    coarse_n = 25
    lattice = splitmix_unit(seed, coarse_n * coarse_n).reshape(coarse_n, coarse_n)
    g = np.arange(n + 1, dtype=np.float64) * ((coarse_n - 1) / float(n))
    gi = np.minimum(g.astype(np.int64), coarse_n - 2)
    gf = g - gi
    a = lattice[gi][:, gi]
    b = lattice[gi][:, gi + 1]
    c = lattice[gi + 1][:, gi]
    d = lattice[gi + 1][:, gi + 1]
    fx = gf[None, :]
    fz = gf[:, None]
    hmap = (a * (1 - fx) + b * fx) * (1 - fz) + (c * (1 - fx) + d * fx) * fz
    jitter = splitmix_unit(seed ^ 0x1234567, (n + 1) * (n + 1)).reshape(n + 1, n + 1)
    ymap = height * 0.08 + hmap * (height * 0.35) + jitter * (cell * 0.4)
    xs = -half + np.arange(n + 1, dtype=np.float64) * cell
    X, Z = np.meshgrid(xs, xs)           # Z varies along axis 0
    P = np.stack([X, ymap, Z], axis=-1)  # (n+1, n+1, 3)
    p00 = P[:-1, :-1].reshape(-1, 3)
    p01 = P[:-1, 1:].reshape(-1, 3)      # +x
    p10 = P[1:, :-1].reshape(-1, 3)      # +z
    p11 = P[1:, 1:].reshape(-1, 3)
    # counter-clockwise seen from above (+y normal)
    t1 = np.concatenate([p00, p10, p11], axis=1)
    t2 = np.concatenate([p00, p11, p01], axis=1)
    field = np.empty((2 * n * n, 9), np.float64)
    field[0::2] = t1
    field[1::2] = t2

    parts = [base, field]
    mat_parts = [np.array(mats, np.int32), np.full(field.shape[0], 2, np.int32)]
    count = base.shape[0] + field.shape[0]
    if total is not None and total > count:
        pad = total - count
        u = splitmix_unit(seed ^ 0xABCDEF, pad * 12).reshape(pad, 12)
        cc = np.empty((pad, 3))
        cc[:, 0] = (u[:, 0] * 2 - 1) * half
        cc[:, 1] = height * 0.5 + u[:, 1] * height * 0.35
        cc[:, 2] = (u[:, 2] * 2 - 1) * half
        off = (u[:, 3:12] * 2 - 1) * (cell * 4)
        parts.append(np.tile(cc, (1, 3)) + off)
        mat_parts.append(np.full(pad, 3, np.int32))
    verts = np.concatenate(parts, axis=0).astype(np.float32)
    if total is not None:
        verts = verts[:total]
    material = np.concatenate(mat_parts)[:verts.shape[0]]
    kd = np.array([KHAKI, (1.0, 1.0, 1.0), (0.55, 0.7, 0.8), (0.445, 0.1, 0.1)], np.float32)
    is_light = np.array([0, 1, 0, 0], np.uint8)
    return {"verts": verts, "material": material,
            "group_index": np.arange(verts.shape[0], dtype=np.int32), "kd": kd,
            "is_light": is_light,
            "camera": {"x": 0.0, "y": height * 0.55, "z": wall * 0.92, "rx": -18.0, "ry": 0.0,
                       "rz": 0.0}}


# ---- camera / pass helpers (frontend utility/matrix.js, Renderer.js:365-437) ----

def camera_transform(x, y, z, rx_deg=0.0, ry_deg=0.0, rz_deg=0.0):
    """translate(x,y,z) * rotate(rx,ry,rz) in the frontend's row-of-16 layout."""
    rx, ry, rz = (math.radians(a) for a in (rx_deg, ry_deg, rz_deg))
    cx, sx, cy, sy, cz, sz = (math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry),
                              math.cos(rz), math.sin(rz))
    rot = [cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx, 0,
           sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx, 0,
           -sy, cy * sx, cy * cx, 0,
           0, 0, 0, 1]
    tr = [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, x, y, z, 1]

    def mat_point(m, p):
        px, py, pz, pw = p
        return [px * m[0] + py * m[4] + pz * m[8] + pw * m[12],
                px * m[1] + py * m[5] + pz * m[9] + pw * m[13],
                px * m[2] + py * m[6] + pz * m[10] + pw * m[14],
                px * m[3] + py * m[7] + pz * m[11] + pw * m[15]]
    out = []
    for r in range(4):
        out += mat_point(tr, rot[4 * r:4 * r + 4])
    return out


def make_pass_dict(width, height, camera, spp=1, bounces=0, light_samples=1, gamma=2.2,
                   gamma_on=True, intensity=100.0, seed=1, fov=45.0, ppd=1.0, projection=0,
                   x_offset=0, y_offset=0, tile_w=None, tile_h=None, task_id=0):
    """RenderPass JSON-shaped dict (reference models/context.go:40-52)."""
    return {
        "Camera": {"Transform": camera_transform(camera["x"], camera["y"], camera["z"],
                                                 camera.get("rx", 0.0), camera.get("ry", 0.0),
                                                 camera.get("rz", 0.0)),
                   "ProjectionPlaneDistance": ppd, "RaysPerPixel": spp, "Projection": projection,
                   "OrtographicSize": 3.0, "FieldOfView": fov},
        "TotalWidth": width, "TotalHeight": height, "TaskID": task_id, "RenderKey": 1,
        "Width": tile_w if tile_w is not None else width,
        "Height": tile_h if tile_h is not None else height,
        "XOffset": x_offset, "YOffset": y_offset, "RNGSeed": seed + task_id,
        "Settings": {"DrawSurfaceNormal": True, "GammaCorrection": gamma_on, "Gamma": gamma,
                     "BounceLimit": bounces, "LightSampleRays": light_samples,
                     "LightIntensity": intensity, "DebugLightSize": 1.0,
                     "ForceDebugLight": False, "DebugLightAtCamera": False,
                     "DebugLightTransform": camera_transform(0, 0, 0)},
    }


def tiles(width, height, tasks_per_dim):
    """The frontend's tile grid (Renderer.js:388-405): floor-sized tiles, the
    last row/column takes the remainder.  Returns [(task_id, x, y, w, h)]."""
    n = max(1, int(tasks_per_dim))
    tw, th = width // n, height // n
    out = []
    for j in range(n):
        for i in range(n):
            w = tw + (width - n * tw) if i == n - 1 else tw
            h = th + (height - n * th) if j == n - 1 else th
            out.append((i + j * n, i * tw, j * th, w, h))
    return out


def primary_rays(width, height, camera, fov=45.0, ppd=1.0):
    """Pixel-centre primary rays computed with numpy float32 arithmetic, for
    the batch ray-cast tests (jitter 0.5, 0.5).  Not bit-identical to the
    library's ray generator; it only feeds both sides the same ray set."""
    aspect = np.float32(width) / np.float32(height)
    vha = np.float32(math.pi) * np.float32(fov / 2.0) / np.float32(180.0)
    hha = vha * aspect
    half_w = np.float32(ppd * math.tan(float(hha)))
    half_h = np.float32(ppd * math.tan(float(vha)))
    xs = (np.arange(width, dtype=np.float32) + np.float32(0.5)) / np.float32(width)
    ys = (np.arange(height, dtype=np.float32) + np.float32(0.5)) / np.float32(height)
    lx = -half_w + xs * (2 * half_w)
    ly = half_h - ys * (2 * half_h)
    LX, LY = np.meshgrid(lx, ly)
    d = np.stack([LX, LY, np.full_like(LX, -ppd)], axis=-1).reshape(-1, 3).astype(np.float32)
    m = np.array(camera_transform(camera["x"], camera["y"], camera["z"], camera.get("rx", 0.0),
                                  camera.get("ry", 0.0), camera.get("rz", 0.0)),
                 np.float32).reshape(4, 4)     # rows = columns of the matrix
    rot = m[:3, :3]
    dw = (d @ rot).astype(np.float32)
    o = (dw + m[3, :3]).astype(np.float32)
    n = np.sqrt((dw * dw).sum(axis=1, keepdims=True)).astype(np.float32)
    dirs = (dw / n).astype(np.float32)
    return o, dirs
