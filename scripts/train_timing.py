"""Timing of in-loop training at 1080p (diagnostic): per-wave time while training and after."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import __graft_entry__ as g
P = g.load_package(); P.load()
W, H = 1920, 1080
prm = P.default_params(); prm.guide_num_training_waves = int(os.environ.get('VSPG_TT_TRAIN', '24'))
r = P.Renderer(P.fog_box_scene(W, H), prm, W, H)
for w in range(int(os.environ.get('VSPG_TT_TRAIN', '24')) + 6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r.render_wave(w, w + 1); torch.cuda.synchronize(); t1 = time.perf_counter()
    r.post_process_wave(); torch.cuda.synchronize(); t2 = time.perf_counter()
    st = r.training_stats()
    print("wave %2d render %.2f ms  post %.2f ms  train %d it %d regions %s" % (w, (t1 - t0) * 1e3, (t2 - t1) * 1e3, st["training"], st["iteration"], st["n_regions"]))
