"""Test helpers for the z-slab partition: slab regen with the CPU oracle and per-column triangle counts."""
import numpy as np

F = 32
CELL = np.float32(5.0) / np.float32(32)


def oracle_slab_triangles(O, scene, levels, part, nparts, params=None):
    """Compact triangle list [T, 18] f32 (emission order) + voxel list of rank `part`'s z-slab, via the oracle."""
    vox, size = O.create_voxel_field()
    z0, z1 = F * part // nparts, F * (part + 1) // nparts
    zi = np.arange(F ** 3) % F
    vox = vox[(zi >= z0) & (zi < z1)].copy()
    for _ in range(levels):
        vox, size = O.refine(scene, vox, size, params)
    if len(vox) == 0:
        return np.zeros((0, 18), np.float32), vox, np.zeros(F * F, np.int64)
    raw = O.triangles_raw(scene, vox, size, params).reshape(-1, 5, 18)
    valid = np.isfinite(raw[:, :, 0])
    per_vox = valid.sum(1)
    col = (np.floor((vox[:, 0] + np.float32(2.5)) / CELL).astype(np.int64) * F
           + np.floor((vox[:, 1] + np.float32(2.5)) / CELL).astype(np.int64))
    counts = np.bincount(col, weights=per_vox, minlength=F * F).astype(np.int64)
    return raw[valid].reshape(-1, 18).copy(), vox, counts
