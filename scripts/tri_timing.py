"""Timing of triangle scenes (BVH traversal inside the path kernels): fog box / cloud with a heightfield terrain, 1080p."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from conftest import load_package
import scenes
P = load_package(); P.load()
W, H = 1920, 1080
for name, n in (("20 000", 100), ("100 352", 224)):
    for medium in (("fog",) if os.environ.get("VSPG_TRI_ONLY_FOG") else ("fog", "cloud")):
        if medium == "fog":
            scene = P.fog_box_scene(W, H)
        else:
            scene = P.cloud_box_scene(W, H, 256)
        tris, kd = scenes.heightfield_triangles(n)
        P.set_triangles(scene, tris, kd)
        r = P.Renderer(scene, P.app_f_params(), W, H, spp=24)
        for w in range(3): r.render_wave(w, w + 1)
        r.counters()
        t0 = time.perf_counter()
        for w in range(3, 19): r.render_wave(w, w + 1)
        r.counters()
        print("%s + %s-triangle terrain: %s %.3f ms per 1080p wave" % (medium, name, r.kernel_name(), (time.perf_counter() - t0) / 16 * 1e3))
        r.close()
