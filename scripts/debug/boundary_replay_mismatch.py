"""Debug: the cloud_boundary scene (tests/test_host_adapter.py) -- which replayed paths differ between the HIP path and the oracle?"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as g
import oracle_lib
from scenes import add_quad, add_sphere, empty_scene
P = g.load_package(); P.load()
W, H = 64, 48
s = empty_scene(W, H, (0, 0.6, -4.2), (0, 0.15, 0), fov=38.0)
m = s.medium
m.type = P.MEDIUM_GRID
m.sigma_a[:] = (.08,) * 3; m.sigma_s[:] = (7.9,) * 3; m.g = 0.877
m.nx = m.ny = m.nz = 3
m.bounds_min[:] = (-0.8, -0.5, -0.8); m.bounds_max[:] = (0.8, 0.9, 0.8)
dens = np.array([0.2, 1, 0.7, 0.1, 0.9, 0.4, 1, 0.6, 0.3, 0.5, 1.2, 0.8, 0.9, 1.3, 0.6, 0.2, 0.7, 0.4, 0, 0.4, 0.1, 0.3, 0.8, 0.2, 0.1, 0.3, 0], dtype=np.float32)
m.density = dens.ctypes.data_as(C.POINTER(C.c_float))
s.camera_outside_medium = 1
add_sphere(s, (0, 0.2, 0), 1.34, material=P.MATERIAL_INTERFACE, iface=P.IFACE_INSIDE)
add_quad(s, (-6, -1.2, -6), (0, 0, 12), (12, 0, 0), kd=(.4, .35, .3))
P.add_infinite_light(s, P.LIGHT_UNIFORM_INFINITE, (.25, .35, .5))
P.add_infinite_light(s, P.LIGHT_DISTANT, (6, 5.5, 5), (0.4, 0.8, -0.3))
prm = P.app_f_params()
rng = np.random.default_rng(5)
n = 20000
xy = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
si = rng.integers(0, 256, n).astype(np.int32)
r = P.Renderer(s, prm, W, H)
c = oracle_lib.OracleRenderer(s, prm, W, H)
Lg, sg = r.trace_paths(xy, si)
Lc, sc = c.trace_paths(xy, si)
bad = np.flatnonzero((sg != sc) | np.any(Lg.view(np.uint32) != Lc.view(np.uint32), axis=1))
print("mismatching paths:", len(bad), "of", n)
for i in bad[:12]:
    print(i, "pixel", xy[i], "sample", si[i], "segs gpu/oracle", sg[i], sc[i], "L gpu", Lg[i], "oracle", Lc[i])
# vary maxdepth to find the first segment where they part
for md in range(0, 6):
    prm.maxdepth = md
    r2 = P.Renderer(s, prm, W, H); c2 = oracle_lib.OracleRenderer(s, prm, W, H)
    a, sa = r2.trace_paths(xy[bad[:12]], si[bad[:12]]); b, sb = c2.trace_paths(xy[bad[:12]], si[bad[:12]])
    print("maxdepth", md, "seg equal", (sa == sb).tolist(), "L equal", np.all(a.view(np.uint32) == b.view(np.uint32), axis=1).tolist())
    r2.close(); c2.close()
