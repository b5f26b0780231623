"""One-off: the same two 1080p waves rendered twelve times over (fresh renderer each time) against ONE oracle render -- the schedulers of
round 5 (ring queues, tile and job cursors, regrouped chains) decide who runs what when; a race would show as a render that differs.
(Round 5, final build: 0 mismatches in 48 renders; not part of the test suite.)"""
import os, sys
sys.path.insert(0, "tests")
import numpy as np
from conftest import load_package
import oracle_lib, scenes
P = load_package(); P.load()
W, H = 1920, 1080
for wl in ["cloud-scene", "cloud-scene-nvdb", "cloud-scene-guided", "fog"]:
    guided = wl.endswith("-guided")
    base = wl[:-7] if guided else wl
    scene = P.fog_box_scene(W, H) if base == "fog" else P.cloud_scene(W, H, 256, nvdb=base.endswith("nvdb"))
    prm = P.default_params() if guided else P.app_f_params()
    field = scenes.light_field(P, n=4) if guided else None
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=0)
    if guided: c.set_guiding_field(field, field)
    for w in range(2): c.render_wave(w, w + 1, 0)
    fc = c.film(); cc = c.counters(); c.close()
    bad = 0
    for rep in range(12):
        g = P.Renderer(scene, prm, W, H, seed=0)
        if guided: g.set_guiding_field(field, field)
        for w in range(2): g.render_wave(w, w + 1)
        fg = g.film()
        # two samples per pixel: the film holds their sum (float on the GPU, double then float in the oracle): weights exact, RGB to rounding
        ok = np.array_equal(fg[..., 3], fc[..., 3]) and np.allclose(fg[..., :3], fc[..., :3], rtol=2e-6, atol=1e-7) and g.counters() == cc
        bad += 0 if ok else 1
        g.close()
    print(wl, "12 renders of two 1080p waves against the oracle: mismatches", bad, flush=True)
