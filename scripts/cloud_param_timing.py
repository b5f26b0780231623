"""Diagnostic: ms per 1080p wave of the cloud workload under parameter variations, per kernel (VSPG_KERNEL)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import __graft_entry__ as g
P = g.load_package(); P.load()
W, H = 1920, 1080
scene = P.cloud_box_scene(W, H, 256)
def run(label, **kw):
    prm = P.app_f_params()
    for k, v in kw.items(): setattr(prm, k, v)
    r = P.Renderer(scene, prm, W, H)
    for w in range(2): r.render_wave(w, w + 1); r.post_process_wave()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 4
    for w in range(2, 2 + n): r.render_wave(w, w + 1)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    c = r.counters()
    print("%-5s %-24s %.2f ms/wave  seg/path %.2f dq/path %.1f shadow/path %.2f" % (os.environ.get("VSPG_KERNEL", "dflt"), label, dt * 1e3,
          c["segments"] / max(1, c["paths"]), c["density_queries"] / max(1, c["paths"]), c["shadow_rays"] / max(1, c["paths"])))
    r.close()
run("default")
run("usenee=0", usenee=0)
run("maxdepth=1", maxdepth=1)
run("maxdepth=1 usenee=0", maxdepth=1, usenee=0)
run("nds", vspsamplingmethod=P.VSP_NDS)
