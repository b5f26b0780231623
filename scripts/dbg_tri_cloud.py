"""Diagnostic: the cloud + terrain scene through the wavefront pipeline of the bounds-flagging debug build."""
import ctypes as C, os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["VSPG_LIB"] = os.path.abspath("build/libvspg_hip_dbg.so")
import numpy as np
import __graft_entry__ as g
P = g.load_package(); lib = P.load()
from scenes import cloud_density, grid_scene, heightfield_triangles
W, H = 64, 48
scene = grid_scene(cloud_density(24), (24, 24, 24), 0.08, 7.9, g=0.6, bmin=(-0.8, -0.5, -0.5), bmax=(0.8, 0.7, 0.9), W=W, H=H)
tris, kd = heightfield_triangles(int(os.environ.get("NT", "100")))
if os.environ.get("NOTRI") != "1":
    P.set_triangles(scene, tris, kd)
r = P.Renderer(scene, P.app_f_params(), W, H, seed=6)
print("kernel", r.kernel_name()); sys.stdout.flush()
xy0 = np.stack(np.meshgrid(np.arange(W), np.arange(H), indexing="xy"), -1).reshape(-1, 2).astype(np.int32)
Lbefore, segbefore = r.trace_paths(xy0, np.zeros(len(xy0), dtype=np.int32))
print("replay BEFORE the pipeline ran: pixel (52,2)", Lbefore[2 * 64 + 52], segbefore[2 * 64 + 52], "min", Lbefore.min()); sys.stdout.flush()
r.render_wave(0, 1)
out = (C.c_uint * 8)()
lib.vspg_dbg_read.argtypes = [C.POINTER(C.c_uint)]
print("dbg_read rc", lib.vspg_dbg_read(out), [hex(x) for x in out]); sys.stdout.flush()
print(r.counters())
film = r.film()
xy = np.stack(np.meshgrid(np.arange(W), np.arange(H), indexing="xy"), -1).reshape(-1, 2).astype(np.int32)
L0, seg = r.trace_paths(xy, np.zeros(len(xy), dtype=np.int32))
f = film[..., :3].reshape(-1, 3)
bad = np.nonzero(np.any(f.view(np.uint32) != L0.astype(np.float32).view(np.uint32), axis=1))[0]
print("replay after vs before: differing", int(np.any(Lbefore != L0, axis=1).sum()))
print("pixels differing from their replayed path:", len(bad), "of", len(xy))
for i in bad[:12]:
    print(xy[i], "film", f[i], "replay", L0[i], "segments", seg[i])
import collections
print("segments histogram of bad:", collections.Counter(seg[bad].tolist()), "all:", collections.Counter(seg.tolist()))
