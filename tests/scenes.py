"""Scene builders shared by the CPU and GPU tests (inputs only)."""
import ctypes as C

import numpy as np

import oracle_lib
from conftest import load_package


def d3_density():
    """8^3 density = RNG(7).Uniform<float>() in x-fastest order."""
    lib = oracle_lib.load()
    f = (C.c_float * 512)()
    lib.oracle_rng_seq(7, 0, 0, 0, 512, None, f)
    return np.array(f, dtype=np.float32)


def grid_scene(density, n, sigma_a, sigma_s, g=0.0, bmin=(0, 0, 0), bmax=(1, 1, 1), W=16, H=16):
    """fog-box geometry with the homogeneous fog replaced by a GridMedium ("uniformgrid")."""
    P = load_package()
    s = oracle_lib.fog_box_scene(W, H)
    m = s.medium
    m.type = P.MEDIUM_GRID
    m.sigma_a[:] = (sigma_a,) * 3 if np.isscalar(sigma_a) else tuple(sigma_a)
    m.sigma_s[:] = (sigma_s,) * 3 if np.isscalar(sigma_s) else tuple(sigma_s)
    m.g = g
    m.nx, m.ny, m.nz = n
    m.bounds_min[:] = bmin
    m.bounds_max[:] = bmax
    assert density.dtype == np.float32 and density.size == n[0] * n[1] * n[2]
    m.density = density.ctypes.data_as(C.POINTER(C.c_float))
    s._density_keepalive = density
    return s


def cloud_density(n, seed=5):
    """seeded value-noise-like density in [0, 1.3], ~25 % empty voxels (stand-in for the Disney cloud)."""
    rng = np.random.default_rng(seed)
    coarse = rng.random((n // 4 + 2,) * 3)
    idx = (np.arange(n) + 0.5) / 4.0
    i0 = np.floor(idx).astype(int)
    f = idx - i0

    def interp(a, axis):
        a0 = np.take(a, i0, axis=axis)
        a1 = np.take(a, i0 + 1, axis=axis)
        shape = [1, 1, 1]
        shape[axis] = n
        w = f.reshape(shape)
        return a0 * (1 - w) + a1 * w

    v = interp(interp(interp(coarse, 0), 1), 2)
    v = np.clip(v * 2.2 - 0.75, 0.0, None)
    return np.ascontiguousarray(v.transpose(2, 1, 0).reshape(-1).astype(np.float32))  # x fastest
