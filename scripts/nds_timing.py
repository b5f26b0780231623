"""Timing of the 256^3 cloud under vspsamplingmethod "nds" (SampleT_maj_OpticalDepthSpace on a grid medium): the per-lane kernel
against the workgroup kernel (VSPG_KERNEL=lane | wg), 1080p, ms per wave."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
import __graft_entry__ as g
P = g.load_package(); P.load()
W, H = 1920, 1080
for kind in (sys.argv[1:] or ["grid", "nvdb"]):
    scene = {"grid": lambda: P.cloud_box_scene(W, H, 256), "nvdb": lambda: P.nanovdb_box_scene(W, H, 256), "scene": lambda: P.cloud_scene(W, H, 256),
             "scene-nvdb": lambda: P.cloud_scene(W, H, 256, nvdb=True)}[kind]()
    for kernel in ("lane", "wf"):
        os.environ["VSPG_KERNEL"] = kernel
        try:
            prm = P.app_f_params(); prm.vspsamplingmethod = P.VSP_NDS
            r = P.Renderer(scene, prm, W, H)
            name = r.kernel_name()
            for w in range(2):
                r.render_wave(w, w + 1); r.post_process_wave()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for w in range(2, 8):
                r.render_wave(w, w + 1); r.post_process_wave()
            torch.cuda.synchronize(); t1 = time.perf_counter()
            print("%s nds %-5s %-60s %.2f ms per wave" % (kind, kernel, name, (t1 - t0) / 6 * 1e3), flush=True)
            r.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
