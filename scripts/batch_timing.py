"""Diagnostic: ms per 1-spp 1080p wave when several waves share one launch (no PostProcessWave work between them)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import __graft_entry__ as g
P = g.load_package(); P.load()
W, H = 1920, 1080
r = P.Renderer(P.fog_box_scene(W, H), P.app_f_params(), W, H)
for w in range(4): r.render_wave(w, w + 1); r.post_process_wave()
w = 4
for batch in (1, 2, 4, 8, 16, 64):
    n = max(1, 64 // batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        r.render_wave(w, w + batch); w += batch
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (n * batch)
    print("batch %2d: %.3f ms per 1-spp wave  %.0f Mpaths/s" % (batch, dt * 1e3, W * H / dt / 1e6))
