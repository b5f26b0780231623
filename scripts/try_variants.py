import sys, time, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch
import __graft_entry__ as g
P = g.load_package()
variant = sys.argv[1]
P.LIB_PATH = os.path.join(os.path.dirname(P.LIB_PATH), variant)
P.load()
W, H = 1920, 1080
prm = P.app_f_params()
if os.environ.get('MAXDEPTH'): prm.maxdepth = int(os.environ['MAXDEPTH'])
r = P.Renderer(P.fog_box_scene(W, H), prm, W, H)
for w in range(3): r.render_wave(w, w + 1); r.post_process_wave()
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 20
for w in range(3, 3 + n): r.render_wave(w, w + 1); r.post_process_wave()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("%-24s %.3f ms/wave %.0f Mpaths/s" % (variant, dt * 1e3, W * H / dt / 1e6))
