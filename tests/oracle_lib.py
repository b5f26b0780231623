"""ctypes loader for oracle/liboracle.so (the CPU checker).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from conftest import ROOT, load_package

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    # VSPG_ORACLE_SO=liboracle_asan.so: the sanitizer build (scripts/run_sanitized_cpu_tests.sh)
    target = os.environ.get("VSPG_ORACLE_SO", "liboracle.so")
    so = os.path.join(ROOT, "oracle", target)
    src = os.path.join(ROOT, "oracle", "vspg_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), target])
    P = load_package()
    lib = C.CDLL(so)
    p, vp, f3 = C.POINTER, C.c_void_p, P.f3
    sig = {
        "oracle_murmur64a": (C.c_uint64, [C.c_char_p, C.c_size_t, C.c_uint64]),
        "oracle_mix_bits": (C.c_uint64, [C.c_uint64]),
        "oracle_hash_float": (C.c_uint64, [C.c_float]),
        "oracle_hash_pixel_seed": (C.c_uint64, [C.c_int32, C.c_int32, C.c_int32]),
        "oracle_hash_point3": (C.c_uint64, [C.c_float, C.c_float, C.c_float]),
        "oracle_rng_seq": (None, [C.c_uint64, C.c_uint64, C.c_int, C.c_int64, C.c_int, p(C.c_uint32), p(C.c_float)]),
        "oracle_fast_exp": (C.c_float, [C.c_float]),
        "oracle_sample_exponential": (C.c_float, [C.c_float, C.c_float]),
        "oracle_sample_discrete2": (C.c_int, [C.c_float, C.c_float, C.c_float]),
        "oracle_henyey_greenstein": (C.c_float, [C.c_float, C.c_float]),
        "oracle_sample_henyey_greenstein": (None, [f3, C.c_float, C.c_float, C.c_float, f3, p(C.c_float)]),
        "oracle_sample_uniform_sphere": (None, [C.c_float, C.c_float, f3]),
        "oracle_sample_cosine_hemisphere": (None, [C.c_float, C.c_float, f3]),
        "oracle_coordinate_system": (None, [f3, f3, f3]),
        "oracle_offset_ray_origin": (None, [f3, f3, f3, f3, f3]),
        "oracle_frame_xz": (None, [f3, f3, f3, f3, f3, f3]),
        "oracle_spawn_ray_to": (None, [f3, f3, f3, f3, f3, f3, f3, f3]),
        "oracle_apply_inverse_identity": (None, [f3, f3, C.c_float, f3, p(C.c_float)]),
        "oracle_apply_inverse_ray": (None, [C.c_float * 16, f3, f3, C.c_float, f3, f3, p(C.c_float)]),
        "oracle_apply_inverse_point": (None, [C.c_float * 16, f3, f3]),
        "oracle_bounds3": (C.c_int, [f3, f3, f3, f3, C.c_float, C.POINTER(C.c_float * 2), f3]),
        "oracle_independent_sampler": (None, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int, p(C.c_float)]),
        "oracle_light_sample_batch": (C.c_int, [vp, C.c_int, p(C.c_float), p(C.c_float), p(C.c_float), p(C.c_int32), p(C.c_float)]),
        "oracle_light_pmf_batch": (C.c_int, [vp, C.c_int, p(C.c_float), p(C.c_float), p(C.c_int32), p(C.c_float)]),
        "oracle_renderer_create": (C.c_int, [p(P.VspgScene), p(P.VspgIntegratorParams), p(P.VspgRenderConfig), p(vp)]),
        "oracle_renderer_destroy": (None, [vp]),
        "oracle_render_wave": (C.c_int, [vp, C.c_int, C.c_int, C.c_int]),
        "oracle_render_window": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
        "oracle_post_process_wave": (C.c_int, [vp]),
        "oracle_isg_update_due": (C.c_int, [vp, C.c_int]),
        "oracle_post_process_step": (C.c_int, [vp, C.c_int, p(C.c_float)]),
        "oracle_film_read": (None, [vp, p(C.c_float)]),
        "oracle_film_read_f64": (None, [vp, p(C.c_double)]),
        "oracle_film_clear": (None, [vp]),
        "oracle_vsp_buffer_read": (None, [vp, p(C.c_float), p(C.c_int)]),
        "oracle_vsp_buffer_write": (None, [vp, p(C.c_float), C.c_int]),
        "oracle_vsp_buffer_load": (None, [vp, p(C.c_float)]),
        "oracle_tr_buffer_read": (C.c_int, [vp, p(C.c_float)]),
        "oracle_tr_buffer_write": (C.c_int, [vp, p(C.c_float)]),
        "oracle_isg_stats_read": (None, [vp, p(C.c_float)]),
        "oracle_get_counters": (None, [vp, p(P.VspgCounters)]),
        "oracle_reset_counters": (None, [vp]),
        "oracle_trace_paths": (C.c_int, [vp, C.c_int, p(C.c_int32), p(C.c_int32), p(C.c_float), p(C.c_int32)]),
        "oracle_sample_tmaj_batch": (C.c_int, [vp, C.c_int, C.c_int, p(P.VspgTmajQuery), p(P.VspgTmajResult)]),
        "oracle_ray_batch": (C.c_int, [vp, C.c_int, p(P.VspgRayQuery), p(P.VspgRayResult)]),
        "oracle_renderer_set_guiding_field": (C.c_int, [vp, p(P.VspgField), p(P.VspgField)]),
        "oracle_guiding_query_batch": (C.c_int, [vp, C.c_int, C.c_float, C.c_int, p(C.c_float), p(C.c_float), p(C.c_float),
                                                 p(C.c_float), p(C.c_int32), p(C.c_float), p(C.c_float), p(C.c_float),
                                                 p(C.c_float), p(C.c_float)]),
        "oracle_renderer_training_stats": (C.c_int, [vp, p(P.VspgTrainStats)]),
        "oracle_train_samples_read": (C.c_int, [vp, p(P.VspgTrainSample), C.c_size_t, p(C.c_size_t)]),
        "oracle_renderer_get_guiding_field": (C.c_int, [vp, C.c_int, p(P.VspgKdNode), p(P.VspgFieldRegion), p(C.c_int32),
                                                        p(C.c_int32)]),
        "oracle_integrator_params_default": (None, [p(P.VspgIntegratorParams)]),
        "oracle_camera_look_at": (C.c_int, [p(P.VspgCamera), f3, f3, f3, C.c_float, C.c_int, C.c_int]),
        "oracle_scene_fog_box": (C.c_int, [p(P.VspgScene), C.c_int, C.c_int]),
        "oracle_interval_op": (None, [C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float * 2]),
        "oracle_blackbody": (None, [C.c_float, C.c_float, C.c_float * 6]),
        "oracle_blackbody_radiance": (C.c_float, [C.c_float, C.c_float]),
        "oracle_sphere_skip": (None, [p(P.VspgSphere), f3, f3, C.c_float, p(C.c_int), f3, f3, p(C.c_int)]),
        "oracle_sphere_intersect": (None, [p(P.VspgSphere), f3, f3, C.c_float, p(C.c_int), p(C.c_float), f3, f3, f3, f3]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def default_params():
    P = load_package()
    prm = P.VspgIntegratorParams()
    load().oracle_integrator_params_default(C.byref(prm))
    return prm


def app_f_params():
    prm = default_params()
    prm.surfaceguiding = 0
    prm.volumeguiding = 0
    prm.vspsecondaryguiding = 0
    return prm


def fog_box_scene(xres, yres):
    P = load_package()
    s = P.VspgScene()
    rc = load().oracle_scene_fog_box(C.byref(s), xres, yres)
    assert rc == 0
    return s


class OracleRenderer:
    def __init__(self, scene, params, xres, yres, spp=1, seed=0, shard_index=0, shard_count=1):
        P = load_package()
        self.lib = load()
        self.cfg = P.VspgRenderConfig(xres, yres, spp, seed, shard_index, shard_count, 0)
        self.h = C.c_void_p()
        rc = self.lib.oracle_renderer_create(C.byref(scene), C.byref(params), C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise RuntimeError("oracle_renderer_create failed: %d" % rc)
        self.xres, self.yres = xres, yres
        self.P = P

    def close(self):
        if getattr(self, "h", None):
            self.lib.oracle_renderer_destroy(self.h)
            self.h = None

    __del__ = close

    def render_wave(self, w0, w1, nthreads=0):
        assert self.lib.oracle_render_wave(self.h, w0, w1, nthreads) == 0

    def render_window(self, x0, y0, x1, y1, w0, w1, nthreads=0):
        assert self.lib.oracle_render_window(self.h, x0, y0, x1, y1, w0, w1, nthreads) == 0

    def post_process_wave(self):
        assert self.lib.oracle_post_process_wave(self.h) == 0

    def isg_update_due(self, n_waves=1):
        return bool(self.lib.oracle_isg_update_due(self.h, int(n_waves)))

    def post_process_step(self, n_waves, stats_sum=None):
        """stats_sum: torch / numpy float32 array of the all-reduced statistics, or None."""
        ptr = None
        if stats_sum is not None:
            a = stats_sum.numpy() if hasattr(stats_sum, "numpy") else np.asarray(stats_sum)
            assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"] and a.size == self.xres * self.yres * self.P.VSPG_ISG_STATS
            ptr = a.ctypes.data_as(C.POINTER(C.c_float))
        assert self.lib.oracle_post_process_step(self.h, int(n_waves), ptr) == 0

    def film(self):
        out = np.empty((self.yres, self.xres, 4), dtype=np.float32)
        self.lib.oracle_film_read(self.h, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def film_f64(self):
        out = np.empty((self.yres, self.xres, 4), dtype=np.float64)
        self.lib.oracle_film_read_f64(self.h, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def vsp_buffer(self):
        out = np.empty((self.yres, self.xres), dtype=np.float32)
        ready = C.c_int()
        self.lib.oracle_vsp_buffer_read(self.h, out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(ready))
        return out, bool(ready.value)

    def set_vsp_buffer(self, vsp, ready=True):
        v = np.ascontiguousarray(vsp, dtype=np.float32)
        self.lib.oracle_vsp_buffer_write(self.h, v.ctypes.data_as(C.POINTER(C.c_float)), int(ready))

    def load_vsp_buffer(self, vsp):
        v = np.ascontiguousarray(vsp, dtype=np.float32)
        assert v.shape == (self.yres, self.xres)
        self.lib.oracle_vsp_buffer_load(self.h, v.ctypes.data_as(C.POINTER(C.c_float)))

    def tr_buffer(self):
        out = np.empty((self.yres, self.xres, 3), dtype=np.float32)
        rc = self.lib.oracle_tr_buffer_read(self.h, out.ctypes.data_as(C.POINTER(C.c_float)))
        assert rc == 0, "the oracle renderer keeps no transmittance buffer"
        return out

    def set_tr_buffer(self, rgb):
        v = np.ascontiguousarray(rgb, dtype=np.float32)
        assert v.shape == (self.yres, self.xres, 3)
        assert self.lib.oracle_tr_buffer_write(self.h, v.ctypes.data_as(C.POINTER(C.c_float))) == 0

    def isg_stats(self):
        out = np.empty((self.yres, self.xres, self.P.VSPG_ISG_STATS), dtype=np.float32)
        self.lib.oracle_isg_stats_read(self.h, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def counters(self):
        c = self.P.VspgCounters()
        self.lib.oracle_get_counters(self.h, C.byref(c))
        return c.as_dict()

    def trace_paths(self, pixel_xy, sample_index):
        pix = np.ascontiguousarray(pixel_xy, dtype=np.int32).reshape(-1, 2)
        si = np.ascontiguousarray(sample_index, dtype=np.int32).reshape(-1)
        n = si.shape[0]
        L = np.empty((n, 3), dtype=np.float32)
        seg = np.empty(n, dtype=np.int32)
        rc = self.lib.oracle_trace_paths(self.h, n, pix.ctypes.data_as(C.POINTER(C.c_int32)),
                                         si.ctypes.data_as(C.POINTER(C.c_int32)),
                                         L.ctypes.data_as(C.POINTER(C.c_float)),
                                         seg.ctypes.data_as(C.POINTER(C.c_int32)))
        assert rc == 0
        return L, seg

    def ray_batch(self, queries):
        q = np.ascontiguousarray(queries, dtype=self.P.RAY_QUERY_DTYPE)
        out = np.zeros(len(q), dtype=self.P.RAY_RESULT_DTYPE)
        assert self.lib.oracle_ray_batch(self.h, len(q), q.ctypes.data_as(C.POINTER(self.P.VspgRayQuery)), out.ctypes.data_as(C.POINTER(self.P.VspgRayResult))) == 0
        return out

    def sample_tmaj_batch(self, variant, queries):
        n = len(queries)
        q = (self.P.VspgTmajQuery * n)(*queries)
        out = (self.P.VspgTmajResult * n)()
        assert self.lib.oracle_sample_tmaj_batch(self.h, variant, n, q, out) == 0
        return list(out)

    def training_stats(self):
        st = self.P.VspgTrainStats()
        assert self.lib.oracle_renderer_training_stats(self.h, C.byref(st)) == 0
        return st.as_dict()

    def train_samples(self):
        n = C.c_size_t(0)
        assert self.lib.oracle_train_samples_read(self.h, None, 0, C.byref(n)) == 0
        buf = (self.P.VspgTrainSample * max(1, n.value))()
        assert self.lib.oracle_train_samples_read(self.h, buf, n.value, C.byref(n)) == 0
        return np.frombuffer(buf, dtype=self.P.TRAIN_SAMPLE_DTYPE, count=n.value).copy()

    def get_guiding_field(self, volume):
        nn, nr = C.c_int32(0), C.c_int32(0)
        assert self.lib.oracle_renderer_get_guiding_field(self.h, int(volume), None, None, C.byref(nn), C.byref(nr)) == 0
        nodes = (self.P.VspgKdNode * max(1, nn.value))()
        regs = (self.P.VspgFieldRegion * max(1, nr.value))()
        assert self.lib.oracle_renderer_get_guiding_field(self.h, int(volume), nodes, regs, C.byref(nn), C.byref(nr)) == 0
        return nodes, regs, nn.value, nr.value

    def set_guiding_field(self, surface, volume):
        rc = self.lib.oracle_renderer_set_guiding_field(self.h, C.byref(surface.pod) if surface else None,
                                                        C.byref(volume.pod) if volume else None)
        assert rc == 0, rc

    def guiding_query_batch(self, is_volume, g, p, n_or_wo, wi, u):
        return guiding_query(self.lib.oracle_guiding_query_batch, self.h, is_volume, g, p, n_or_wo, wi, u, None)

    def light_sample_batch(self, p, ns, u):
        """lightSampler.Sample(ctx, u) per context: (light index into {emissive rectangles, infinite lights}, -1 = none; pmf)"""
        fp = C.POINTER(C.c_float)
        p, ns, u = (np.ascontiguousarray(x, dtype=np.float32) for x in (p, ns, u))
        li = np.zeros(len(u), dtype=np.int32)
        pmf = np.zeros(len(u), dtype=np.float32)
        self.lib.oracle_light_sample_batch(self.h, len(u), p.ctypes.data_as(fp), ns.ctypes.data_as(fp), u.ctypes.data_as(fp),
                                           li.ctypes.data_as(C.POINTER(C.c_int32)), pmf.ctypes.data_as(fp))
        return li, pmf

    def light_pmf_batch(self, p, ns, light):
        fp = C.POINTER(C.c_float)
        p, ns = (np.ascontiguousarray(x, dtype=np.float32) for x in (p, ns))
        light = np.ascontiguousarray(light, dtype=np.int32)
        pmf = np.zeros(len(light), dtype=np.float32)
        self.lib.oracle_light_pmf_batch(self.h, len(light), p.ctypes.data_as(fp), ns.ctypes.data_as(fp),
                                        light.ctypes.data_as(C.POINTER(C.c_int32)), pmf.ctypes.data_as(fp))
        return pmf


def guiding_query(fn, handle, is_volume, g, p, n_or_wo, wi, u, stream):
    fp = C.POINTER(C.c_float)
    p, a, wi, u = (np.ascontiguousarray(x, dtype=np.float32) for x in (p, n_or_wo, wi, u))
    n = p.shape[0]
    ok = np.zeros(n, dtype=np.int32)
    pdf, inc, vsp, pdfs = (np.zeros(n, dtype=np.float32) for _ in range(4))
    ws = np.zeros((n, 3), dtype=np.float32)
    args = [handle, int(is_volume), float(g), n, p.ctypes.data_as(fp), a.ctypes.data_as(fp), wi.ctypes.data_as(fp),
            u.ctypes.data_as(fp), ok.ctypes.data_as(C.POINTER(C.c_int32)), pdf.ctypes.data_as(fp), inc.ctypes.data_as(fp),
            vsp.ctypes.data_as(fp), ws.ctypes.data_as(fp), pdfs.ctypes.data_as(fp)]
    if stream is not None:
        args.append(C.c_void_p(stream))
    rc = fn(*args)
    assert rc == 0, rc
    return dict(ok=ok, pdf=pdf, incoming_pdf=inc, vsp=vsp, ws=ws, pdf_s=pdfs)
