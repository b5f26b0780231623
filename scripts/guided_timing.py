"""Diagnostic: ms per 1080p wave of the fog box with a trained guiding field, under guiding-option variations."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import __graft_entry__ as g
P = g.load_package(); P.load()
W, H = 1920, 1080
# train once with everything on, keep the field
prm = P.default_params(); prm.guide_num_training_waves = 12
r = P.Renderer(P.fog_box_scene(W, H), prm, W, H)
for w in range(12): r.render_wave(w, w + 1); r.post_process_wave()
fields = [r.get_guiding_field(f) for f in (0, 1)]
r.close()
F = [P.Field(list(n[:nn]), list(rg[:nr])) for (n, rg, nn, nr) in fields]
def run(label, **kw):
    prm = P.default_params()
    for k, v in kw.items(): setattr(prm, k, v)
    r = P.Renderer(P.fog_box_scene(W, H), prm, W, H)
    if prm.surfaceguiding or prm.volumeguiding or prm.vspsecondaryguiding: r.set_guiding_field(F[0], F[1])
    for w in range(12, 15): r.render_wave(w, w + 1); r.post_process_wave()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 16
    for w in range(15, 15 + n): r.render_wave(w, w + 1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    c = r.counters()
    print("%-44s %.3f ms/wave  segments/path %.2f" % (label, dt * 1e3, c["segments"] / max(1, c["paths"])))
    r.close()
run("all guiding (reference defaults)")
run("no guiding at all", surfaceguiding=0, volumeguiding=0, vspsecondaryguiding=0)
run("surface only (RIS)", volumeguiding=0, vspsecondaryguiding=0)
run("surface only (MIS)", volumeguiding=0, vspsecondaryguiding=0, surfaceguidingtype=P.GUIDE_MIS)
run("volume only (MIS)", surfaceguiding=0, vspsecondaryguiding=0)
run("volume only (RIS)", surfaceguiding=0, vspsecondaryguiding=0, volumeguidingtype=P.GUIDE_RIS)
run("secondary VSP only", surfaceguiding=0, volumeguiding=0)
run("all guiding, usenee=0", usenee=0)
