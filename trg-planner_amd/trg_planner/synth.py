"""Seeded synthetic point-cloud maps (stand-ins for the reference's downloadable .pcd maps).

The reference's example maps (``prebuilt_maps/sim_{indoor,mountain}_0.1.pcd``, config/*.yaml:7)
are fetched by shellscripts/download_maps.sh and are not available offline, so every run of this
repository uses the generators below.  All outputs are float32 ``(N, 3)``
arrays, emitted in a seeded shuffled order so that an insertion-ordered kd-tree (the reference's
index) is not degenerate.
"""
from __future__ import annotations

import numpy as np


def _fade(t):
    return t * t * t * (t * (t * 6.0 - 15.0) + 10.0)


def _perlin2(x, y, grads):
    """Classic 2-D gradient noise on an integer lattice of unit gradients ``grads[P, P, 2]``."""
    P = grads.shape[0]
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    fx = x - x0
    fy = y - y0
    x0 %= P
    y0 %= P
    x1 = (x0 + 1) % P
    y1 = (y0 + 1) % P

    def dot(ix, iy, dx, dy):
        g = grads[ix, iy]
        return g[..., 0] * dx + g[..., 1] * dy

    n00 = dot(x0, y0, fx, fy)
    n10 = dot(x1, y0, fx - 1.0, fy)
    n01 = dot(x0, y1, fx, fy - 1.0)
    n11 = dot(x1, y1, fx - 1.0, fy - 1.0)
    u = _fade(fx)
    v = _fade(fy)
    return (n00 * (1 - u) + n10 * u) * (1 - v) + (n01 * (1 - u) + n11 * u) * v


def fbm_height(x, y, seed, amplitude=3.0, wavelength=40.0, octaves=4, lacunarity=2.0, gain=0.5):
    """fBm of Perlin gradient noise; x, y in metres (float64 arrays)."""
    rng = np.random.Generator(np.random.PCG64(seed ^ 0x5EED))
    P = 256
    ang = rng.uniform(0.0, 2.0 * np.pi, size=(P, P))
    grads = np.stack([np.cos(ang), np.sin(ang)], axis=-1)
    z = np.zeros_like(x, dtype=np.float64)
    amp = 1.0
    freq = 1.0 / wavelength
    for _ in range(octaves):
        z += amp * _perlin2(x * freq + 17.31, y * freq + 5.77, grads)
        amp *= gain
        freq *= lacunarity
    return amplitude * z


def mountain_cloud(nx, ny, seed=20250418, spacing=0.1, jitter=0.02, z_noise=0.01,
                   amplitude=3.0, wavelength=40.0, origin=(0.0, 0.0), shuffle=True):
    """``nx * ny`` terrain points on a jittered lattice (BASELINE configs 2-5).

    C2: nx=ny=1000 (1.0 M pts, 100 m x 100 m).  C3: nx=3200, ny=3125 (10 M pts).
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty((nx * ny, 3), dtype=np.float32)
    rows = max(1, (1 << 21) // max(nx, 1))
    k = 0
    for j0 in range(0, ny, rows):
        j1 = min(ny, j0 + rows)
        jj, ii = np.meshgrid(np.arange(j0, j1), np.arange(nx), indexing="ij")
        m = ii.size
        x = origin[0] + ii.ravel() * spacing + rng.uniform(-jitter, jitter, m)
        y = origin[1] + jj.ravel() * spacing + rng.uniform(-jitter, jitter, m)
        z = fbm_height(x, y, seed, amplitude, wavelength) + rng.uniform(-z_noise, z_noise, m)
        out[k:k + m, 0] = x
        out[k:k + m, 1] = y
        out[k:k + m, 2] = z
        k += m
    if shuffle:
        rng.shuffle(out, axis=0)
    return out


def indoor_cloud(seed=1, size=(40.0, 30.0), spacing=0.1, n_boxes=12, wall_h=2.5, shuffle=True):
    """Flat floor with axis-aligned box obstacles (walls sampled in x, y and z); C1 stand-in."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nx = int(round(size[0] / spacing))
    ny = int(round(size[1] / spacing))
    ii, jj = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    fx = ii.ravel() * spacing
    fy = jj.ravel() * spacing
    fz = rng.normal(0.0, 0.005, fx.size)
    parts = [np.stack([fx, fy, fz], axis=1)]
    boxes = []
    for _ in range(n_boxes):
        w, h = rng.uniform(1.0, 6.0), rng.uniform(0.3, 4.0)
        if rng.uniform() < 0.5:
            w, h = h, w
        w = min(w, size[0] - 5.0)
        h = min(h, size[1] - 5.0)
        x0 = rng.uniform(2.0, size[0] - 2.0 - w)
        y0 = rng.uniform(2.0, size[1] - 2.0 - h)
        boxes.append((x0, y0, x0 + w, y0 + h))
    # outer walls as thin boxes
    t = 0.2
    boxes += [(0, 0, size[0], t), (0, size[1] - t, size[0], size[1]),
              (0, 0, t, size[1]), (size[0] - t, 0, size[0], size[1])]
    zs = np.arange(0.0, wall_h + 1e-6, spacing)
    for (x0, y0, x1, y1) in boxes:
        xs = np.arange(x0, x1 + 1e-6, spacing)
        ys = np.arange(y0, y1 + 1e-6, spacing)
        gx, gy, gz = np.meshgrid(xs, ys, zs, indexing="ij")
        parts.append(np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=1))
    pts = np.concatenate(parts, axis=0)
    pts[:, :2] += rng.uniform(-0.005, 0.005, size=(pts.shape[0], 2))
    out = pts.astype(np.float32)
    if shuffle:
        rng.shuffle(out, axis=0)
    return out, boxes


def voxel_centroids(xyz, leaf):
    """Centroid per occupied voxel (PCL VoxelGrid semantics, PL.cpp:91-94), numpy reference.

    Output order: ascending voxel key (ix + iy*dx + iz*dx*dy with indices relative to the
    cloud minimum), which is the order pcl::VoxelGrid emits.
    """
    xyz = np.asarray(xyz, dtype=np.float32)
    inv = np.float32(1.0) / np.float32(leaf)
    mn = np.floor(xyz.min(axis=0) * inv).astype(np.int64)
    mx = np.floor(xyz.max(axis=0) * inv).astype(np.int64)
    dims = mx - mn + 1
    ijk = np.floor(xyz * inv).astype(np.int64) - mn
    key = ijk[:, 0] + ijk[:, 1] * dims[0] + ijk[:, 2] * dims[0] * dims[1]
    order = np.argsort(key, kind="stable")
    key_s = key[order]
    uniq, start, counts = np.unique(key_s, return_index=True, return_counts=True)
    sums = np.add.reduceat(xyz[order].astype(np.float64), start, axis=0)
    return (sums / counts[:, None]).astype(np.float32)


def _hash_u01(ix, iy, seed, salt):
    """Counter-based uniform [0,1) per lattice point (splitmix64 finaliser), vectorised."""
    with np.errstate(over="ignore"):
        h = (ix.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)
             ^ iy.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)
             ^ np.uint64(seed & 0xFFFFFFFFFFFFFFFF) * np.uint64(0x165667B19E3779F9)
             ^ np.uint64(salt) * np.uint64(0x27D4EB2F165667C5))
        h ^= h >> np.uint64(30)
        h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(27)
        h *= np.uint64(0x94D049BB133111EB)
        h ^= h >> np.uint64(31)
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def mountain_tile(ix0, ix1, iy0, iy1, seed=20250418, spacing=0.1, jitter=0.02, z_noise=0.01,
                  amplitude=3.0, wavelength=40.0, shuffle=True):
    """Points of the GLOBAL lattice indices [ix0, ix1) x [iy0, iy1) of one continuous terrain.

    Jitter and z-noise are functions of the global lattice index, so overlapping windows (a tile's
    halo and its neighbour's core) contain bit-identical points and the union of all tiles is one
    consistent cloud.  Used by the tiled multi-GPU build.
    """
    out = np.empty(((ix1 - ix0) * (iy1 - iy0), 3), dtype=np.float32)
    nx = ix1 - ix0
    rows = max(1, (1 << 21) // max(nx, 1))
    k = 0
    for j0 in range(iy0, iy1, rows):
        j1 = min(iy1, j0 + rows)
        jj, ii = np.meshgrid(np.arange(j0, j1, dtype=np.int64), np.arange(ix0, ix1, dtype=np.int64),
                             indexing="ij")
        ii = ii.ravel()
        jj = jj.ravel()
        x = ii * spacing + (2.0 * _hash_u01(ii, jj, seed, 1) - 1.0) * jitter
        y = jj * spacing + (2.0 * _hash_u01(ii, jj, seed, 2) - 1.0) * jitter
        z = fbm_height(x, y, seed, amplitude, wavelength) + \
            (2.0 * _hash_u01(ii, jj, seed, 3) - 1.0) * z_noise
        m = ii.size
        out[k:k + m, 0] = x
        out[k:k + m, 1] = y
        out[k:k + m, 2] = z
        k += m
    if shuffle:
        rng = np.random.Generator(np.random.PCG64([seed & 0xFFFFFFFF, ix0 & 0xFFFFFFFF,
                                                   iy0 & 0xFFFFFFFF]))
        rng.shuffle(out, axis=0)
    return out
