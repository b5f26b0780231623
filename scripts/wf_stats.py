"""Diagnostic: where the lanes of the distance walk are (library built with -DVSPG_WF_STATS, selected with VSPG_LIB)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from __graft_entry__ import load_package
P = load_package(); lib = P.load()
wl = sys.argv[1] if len(sys.argv) > 1 else "cloud"
W, H = 1920, 1080
scene = (P.cloud_box_scene(W, H, 256) if wl == "cloud" else P.cloud_scene(W, H, 256) if wl == "cloud-scene" else P.nanovdb_box_scene(W, H, 256))
prm = P.app_f_params()
if wl == "cloud-scene":  # bench.py's options for the boundary scene
    prm.vspsamplingmethod = P.VSP_RESAMPLING
r = P.Renderer(scene, prm, W, H, spp=4)
out = (C.c_ulonglong * 16)()
r.render_wave(0, 1); lib.vspg_wf_stats_read(out)
r.render_wave(1, 2); lib.vspg_wf_stats_read(out)
s = list(out)[:8]
it, act, rounds, rl, cs, cl, dl, dr = s
print(wl, "iterations %d  lanes with a job %.1f  advance rounds/iteration %.2f  lanes per round %.1f  collision steps/iteration %.2f  lanes per collision step %.1f"
      % (it, act / it, rounds / it, rl / rounds, cs / it, cl / max(cs, 1)))
print("draining iterations (no job left to claim): %.1f %% of all, %.1f lanes with a job; the others: %.1f lanes" % (100.0 * dr / it, dl / max(dr, 1), (act - dl) / max(it - dr, 1)))
print("collisions per iteration-lane %.3f" % (cl / (64.0 * it)))
r.close()
