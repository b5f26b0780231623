"""One-off confidence check (not part of the test suite: minutes of oracle time): replay many (pixel, sample) paths on the GPU and on
the CPU oracle and count bit-identical radiances -- 1080p scenes of the bench workloads plus triangle / sky / guided variants."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
import oracle_lib, scenes
P = load_package(); P.load()
W, H = 1920, 1080
rng = np.random.default_rng(2026)

def check(name, scene, prm, n, field=None):
    g = P.Renderer(scene, prm, W, H, seed=11)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=11)
    if field is not None:
        g.set_guiding_field(field, field); c.set_guiding_field(field, field)
    pix = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 1 << 20, n).astype(np.int32)
    t0 = time.time()
    Lg, sg = g.trace_paths(pix, si)
    Lc, sc = c.trace_paths(pix, si)
    same = np.all(Lg.view(np.uint32) == Lc.view(np.uint32), axis=1) & (sg == sc)
    print("%-34s %8d paths: bit-identical %d (%.6f), mean segments %.2f, %.1f s" % (name, n, same.sum(), same.mean(), sc.mean(), time.time() - t0), flush=True)
    g.close(); c.close()
    return bool(same.all())

ok = True
fog = P.fog_box_scene(W, H)
ok &= check("fog (App. F options)", fog, P.app_f_params(), 2000000)
chro = P.fog_box_scene(W, H)
chro.medium.sigma_a[:] = (0.3, 0.1, 0.02); chro.medium.sigma_s[:] = (0.2, 0.9, 1.6); chro.medium.g = 0.6
ok &= check("chromatic anisotropic fog", chro, P.app_f_params(), 500000)
ok &= check("cloud 256^3 (GridMedium)", P.cloud_box_scene(W, H, 256), P.app_f_params(), 300000)
ok &= check("cloud 256^3 (NanoVDB semantics)", P.nanovdb_box_scene(W, H, 256), P.app_f_params(), 200000)
field = scenes.light_field(P, n=4)
ok &= check("fog, reference-default guiding", fog, P.default_params(), 300000, field)
ok &= check("cloud, reference-default guiding", P.cloud_box_scene(W, H, 256), P.default_params(), 100000, field)
tri = P.fog_box_scene(W, H)
t, kd = scenes.heightfield_triangles(100)
P.set_triangles(tri, t, kd)
P.add_infinite_light(tri, P.LIGHT_UNIFORM_INFINITE, (0.35, 0.5, 0.9))
P.add_infinite_light(tri, P.LIGHT_DISTANT, (9.0, 8.0, 6.5), (0.3, 1.0, -0.4))
prm = P.app_f_params(); prm.lightsampler = 0
ok &= check("fog + 20k triangles + sky + sun", tri, prm, 60000)
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
