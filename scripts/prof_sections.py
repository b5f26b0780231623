"""Diagnostic: per-section wave time / lane utilisation of k_render_wave (needs `make -C csrc prof`)."""
import ctypes as C, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as g
P = g.load_package()
P.LIB_PATH = os.path.join(os.path.dirname(P.LIB_PATH), "libvspg_hip_prof.so")
lib = P.load()
W, H = 1920, 1080
r = P.Renderer(P.fog_box_scene(W, H), P.app_f_params(), W, H)
names = ["intersect", "hash+rng", "dist_guided", "dist_plain", "surf_pre", "nee", "nee_transmit", "vol_sample", "surf_sample",
         "finish", "start", "refill", "segment", "wg_refill", "wg_A", "wg_B", "wg_bar_R", "wg_bar_A", "wg_bar_B", "wg_vertex",
         "wg_total"]
buf = (C.c_ulonglong * (len(names) * 3))(); n = C.c_int()
lib.vspg_prof_read.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_int)]
for w in range(3):
    r.render_wave(w, w + 1); r.post_process_wave()
r.counters()
lib.vspg_prof_read(buf, C.byref(n))  # clear
for w in range(3, 7):
    r.render_wave(w, w + 1); r.post_process_wave()
print(r.counters())
lib.vspg_prof_read(buf, C.byref(n))
tot = buf[12 * 3] if os.environ.get('VSPG_KERNEL') == 'lane' else buf[20 * 3]
print("%-14s %12s %8s %10s %8s" % ("section", "wave-cycles", "share", "execs", "lanes"))
for i, nm in enumerate(names):
    t, l, e = buf[3 * i], buf[3 * i + 1], buf[3 * i + 2]
    if e: print("%-14s %12d %7.1f%% %10d %8.1f" % (nm, t, 100.0 * t / max(1, tot), e, l / e))
