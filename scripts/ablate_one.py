import sys
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import __graft_entry__ as g
P = g.load_package(); P.load()
W, H = 1920, 1080
prm = P.app_f_params()
for kv in sys.argv[1:]:
    k, v = kv.split('='); setattr(prm, k, int(v))
r = P.Renderer(P.fog_box_scene(W, H), prm, W, H)
for w in range(3): r.render_wave(w, w + 1); r.post_process_wave()
r.counters()
