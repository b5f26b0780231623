"""One-off confidence check at the benchmark's own size (not part of the test suite: the oracle renders 1080p waves on the host cores):
ONE wave of each unguided bench workload on the GPU's wave kernels -- the schedulers, job cursors, regrouped chains of round 5 -- and
on the CPU oracle; with one sample per pixel the film IS the paths' radiances, so the two films must agree bit for bit, and so must
the counters.   python scripts/full_size_film_check.py [workload ...]  ->  profiles/r05_full_size_film_check.txt"""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
import oracle_lib, scenes
P = load_package(); P.load()
W, H = 1920, 1080
if len(sys.argv) > 1 and "x" in sys.argv[1] and sys.argv[1].replace("x", "").isdigit():  # full_size_film_check.py 3840x2160 [workload ...]
    W, H = (int(v) for v in sys.argv.pop(1).split("x"))
print("%d x %d" % (W, H), flush=True)
WL = {"fog": lambda: P.fog_box_scene(W, H), "cloud": lambda: P.cloud_box_scene(W, H, 256), "cloud-nvdb": lambda: P.nanovdb_box_scene(W, H, 256),
      "cloud-scene": lambda: P.cloud_scene(W, H, 256), "cloud-scene-nvdb": lambda: P.cloud_scene(W, H, 256, nvdb=True)}
ok = True
for wl in (sys.argv[1:] or list(WL) + [w + "-guided" for w in ("fog", "cloud", "cloud-scene")]):
    guided = wl.endswith("-guided")   # the reference's default options over a given (synthetic) guiding field: the guided vertex kernels
    scene, prm = WL[wl[:-7] if guided else wl](), (P.default_params() if guided else P.app_f_params())
    g = P.Renderer(scene, prm, W, H, seed=0)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=0)
    if guided:
        field = scenes.light_field(P, n=4)
        g.set_guiding_field(field, field); c.set_guiding_field(field, field)
    t0 = time.time()
    g.render_wave(0, 1); fg = g.film(); t1 = time.time()
    c.render_wave(0, 1, 0); fc = c.film(); t2 = time.time()
    same = np.all(fg.view(np.uint32) == fc.view(np.uint32), axis=-1)
    cg, co = g.counters(), c.counters()
    print("%-18s %s: %d of %d pixels bit-identical (RGB + weight), counters %s; GPU %.2f s (first wave, with set-up), oracle %.1f s"
          % (wl, g.kernel_name(), int(same.sum()), same.size, "equal" if cg == co else "DIFFER %s %s" % (cg, co), t1 - t0, t2 - t1), flush=True)
    ok &= bool(same.all()) and cg == co
    g.close(); c.close()
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
