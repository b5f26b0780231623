"""Timing ablation of k_render_wave by runtime parameters (diagnostic)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import __graft_entry__ as g
P = g.load_package(); P.load()
W, H = 1920, 1080
def run(tag, **kw):
    prm = P.app_f_params()
    for k, v in kw.items(): setattr(prm, k, v)
    r = P.Renderer(P.fog_box_scene(W, H), prm, W, H)
    for w in range(3): r.render_wave(w, w + 1); r.post_process_wave()
    torch.cuda.synchronize(); r.reset_counters()
    t0 = time.perf_counter()
    n = 10
    for w in range(3, 3 + n): r.render_wave(w, w + 1); r.post_process_wave()
    torch.cuda.synchronize()
    c = r.counters()  # syncs
    dt = (time.perf_counter() - t0) / n
    print("%-28s %.3f ms/wave  %.0f Mpaths/s  seg/path %.2f  shadow/path %.2f" % (tag, dt * 1e3, W * H / dt / 1e6, c['segments'] / c['paths'], c['shadow_rays'] / c['paths']))
    r.close()
run("default")
run("usenee=0", usenee=0)
run("vspguiding=0", vspguiding=0)
run("usenee=0 vspguiding=0", usenee=0, vspguiding=0)
run("maxdepth=1", maxdepth=1)
run("maxdepth=0", maxdepth=0)
run("maxdepth=0 usenee=0", maxdepth=0, usenee=0)
