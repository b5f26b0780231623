"""Diagnostic: ms per 1080p wave of the fog box under parameter variations (what each part of the path costs)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import __graft_entry__ as g
P = g.load_package(); P.load()
W, H = 1920, 1080
def run(label, **kw):
    prm = P.app_f_params()
    for k, v in kw.items(): setattr(prm, k, v)
    r = P.Renderer(P.fog_box_scene(W, H), prm, W, H)
    for w in range(4): r.render_wave(w, w + 1); r.post_process_wave()
    torch.cuda.synchronize(); r.reset_counters() if hasattr(r, "reset_counters") else None
    t0 = time.perf_counter()
    n = 32
    for w in range(4, 4 + n): r.render_wave(w, w + 1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    c = r.counters()
    print("%-28s %.3f ms/wave  segments/path %.2f shadow/path %.2f" % (label, dt * 1e3, c["segments"] / max(1, c["paths"]), c["shadow_rays"] / max(1, c["paths"])))
    r.close()
run("default")
run("usenee=0", usenee=0)
run("vspguiding=0", vspguiding=0)
run("maxdepth=1", maxdepth=1)
run("maxdepth=1 usenee=0", maxdepth=1, usenee=0)
run("maxdepth=0", maxdepth=0)
run("maxdepth=10", maxdepth=10)
