"""vspg-pbrt-v4_amd -- Python plumbing over the C-ABI of the MI355X-native VSPG hot path.

The product is `csrc/libvspg_hip.so` (hand-written HIP kernels for gfx950 + the C-ABI declared
in include/vspg.h) and the C++ host adapter in `host/`.  This module is only a ctypes binding
used by tests/ and bench.py; it contains no algorithm and NO CPU fallback: if the shared
library is missing, `load()` raises.

The directory name carries a hyphen (it is the repo's package directory, not a Python
identifier); import it with `importlib` under the name `vspg_pbrt_v4_amd`
(see tests/conftest.py / bench.py: `load_package()`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VSPG_LIB") or os.path.join(_HERE, "csrc", "libvspg_hip.so")  # VSPG_LIB: experiment builds (scripts/)

VSPG_MAX_QUADS = 16
VSPG_ISG_STATS = 8

MEDIUM_NONE, MEDIUM_HOMOGENEOUS, MEDIUM_GRID, MEDIUM_NANOVDB = 0, 1, 2, 3
GUIDE_MIS, GUIDE_RIS = 0, 1
VSP_CONTRIBUTION, VSP_VARIANCE = 0, 1
VSP_RESAMPLING, VSP_NDS = 0, 1
LIGHTSAMPLER_UNIFORM, LIGHTSAMPLER_POWER, LIGHTSAMPLER_BVH = 0, 1, 2
TMAJ_PLAIN, TMAJ_OPTICAL_DEPTH, TMAJ_RESAMPLING = 0, 1, 2

VSPG_OK, VSPG_EINVAL, VSPG_ENODEVICE, VSPG_EHIP, VSPG_ESCOPE = 0, -1, -2, -3, -4
ARITH_EXACT, ARITH_FAST_WEIGHTS, ARITH_FAST = 0, 1, 2

f3 = C.c_float * 3


class VspgQuad(C.Structure):
    _fields_ = [("p00", f3), ("e1", f3), ("e2", f3), ("Kd", f3), ("Le", f3),
                ("two_sided", C.c_int32), ("reverse_orientation", C.c_int32),
                ("material", C.c_int32), ("medium_interface", C.c_int32)]


VSPG_MAX_SPHERES = 8
MATERIAL_DIFFUSE, MATERIAL_INTERFACE = 0, 1
IFACE_INSIDE, IFACE_OUTSIDE = 1, 2
TRI_INTERFACE, TRI_IFACE_SHIFT, TRI_FLIP_NORMAL = 1, 1, 8


class VspgSphere(C.Structure):
    _fields_ = [("render_from_object", C.c_float * 16), ("object_from_render", C.c_float * 16), ("radius", C.c_float),
                ("Kd", f3), ("reverse_orientation", C.c_int32), ("material", C.c_int32), ("medium_interface", C.c_int32)]


class VspgCamera(C.Structure):
    _fields_ = [("origin", f3), ("right", f3), ("up", f3), ("fwd", f3),
                ("sx", C.c_float), ("ox", C.c_float), ("sy", C.c_float), ("oy", C.c_float)]


class VspgMedium(C.Structure):
    _fields_ = [("type", C.c_int32), ("sigma_a", f3), ("sigma_s", f3), ("g", C.c_float),
                ("Le", f3), ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("bounds_min", f3), ("bounds_max", f3), ("density", C.POINTER(C.c_float)),
                ("index_min", C.c_int32 * 3), ("voxel_size", f3), ("grid_origin", f3),
                ("density_offset", C.c_float), ("majorant_scale", C.c_float),
                ("le_scale", C.POINTER(C.c_float)), ("le_nx", C.c_int32), ("le_ny", C.c_int32), ("le_nz", C.c_int32),
                ("has_transform", C.c_int32), ("render_from_medium", C.c_float * 16), ("medium_from_render", C.c_float * 16),
                ("temperature", C.POINTER(C.c_float)), ("nvdb_le_scale", C.c_float), ("temperature_offset", C.c_float),
                ("temperature_scale", C.c_float)]


VSPG_MAX_INFINITE_LIGHTS = 4
LIGHT_UNIFORM_INFINITE, LIGHT_DISTANT = 0, 1


class VspgInfiniteLight(C.Structure):
    _fields_ = [("type", C.c_int32), ("L", f3), ("w_light", f3)]


class VspgScene(C.Structure):
    _fields_ = [("n_quads", C.c_int32), ("quads", VspgQuad * VSPG_MAX_QUADS),
                ("camera", VspgCamera), ("medium", VspgMedium),
                ("n_triangles", C.c_int32), ("tri_p", C.POINTER(C.c_float)), ("tri_kd", C.POINTER(C.c_float)),
                ("n_infinite_lights", C.c_int32), ("infinite_lights", VspgInfiniteLight * VSPG_MAX_INFINITE_LIGHTS),
                ("tri_flags", C.POINTER(C.c_int32)), ("n_spheres", C.c_int32), ("spheres", VspgSphere * VSPG_MAX_SPHERES),
                ("camera_outside_medium", C.c_int32)]


class VspgIntegratorParams(C.Structure):
    _fields_ = [("maxdepth", C.c_int32), ("minrrdepth", C.c_int32), ("usenee", C.c_int32),
                ("surfaceguiding", C.c_int32), ("volumeguiding", C.c_int32),
                ("surfaceguidingtype", C.c_int32), ("volumeguidingtype", C.c_int32),
                ("vspguiding", C.c_int32), ("vspprimaryguiding", C.c_int32),
                ("vspsecondaryguiding", C.c_int32), ("vspmisratio", C.c_float),
                ("vspcriterion", C.c_int32), ("vspsamplingmethod", C.c_int32),
                ("collisionProbabilityBias", C.c_int32), ("rrguiding", C.c_int32),
                ("lightsampler", C.c_int32), ("regularize", C.c_int32),
                ("guide_num_training_waves", C.c_int32), ("storeTrBuffer", C.c_int32),
                ("surfacerrguiding", C.c_int32), ("volumerrguiding", C.c_int32)]


class VspgRenderConfig(C.Structure):
    _fields_ = [("xres", C.c_int32), ("yres", C.c_int32), ("spp", C.c_int32), ("seed", C.c_int32),
                ("shard_index", C.c_int32), ("shard_count", C.c_int32), ("device", C.c_int32)]


class VspgCounters(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("segments", C.c_uint64), ("volume_scatters", C.c_uint64),
                ("surface_hits", C.c_uint64), ("density_queries", C.c_uint64),
                ("shadow_rays", C.c_uint64), ("shadow_density_queries", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class VspgTrainSample(C.Structure):
    _fields_ = [("p", C.c_float * 3), ("dir", C.c_float * 3), ("weight", C.c_float), ("pdf", C.c_float),
                ("distance", C.c_float), ("flags", C.c_uint32)]


class VspgTrainStats(C.Structure):
    _fields_ = [("training", C.c_int32), ("iteration", C.c_int32), ("n_samples", C.c_uint64), ("n_zero", C.c_uint64),
                ("n_nodes", C.c_int32 * 2), ("n_regions", C.c_int32 * 2), ("n_dropped", C.c_uint64)]

    def as_dict(self):
        return {"training": int(self.training), "iteration": int(self.iteration), "n_samples": int(self.n_samples),
                "n_zero": int(self.n_zero), "n_nodes": list(self.n_nodes), "n_regions": list(self.n_regions),
                "n_dropped": int(self.n_dropped)}


TRAIN_SAMPLE_DTYPE = [("p", "<f4", 3), ("dir", "<f4", 3), ("weight", "<f4"), ("pdf", "<f4"), ("distance", "<f4"),
                      ("flags", "<u4")]

VSPG_FIELD_LOBES = 8
_fl = C.c_float * VSPG_FIELD_LOBES


class VspgKdNode(C.Structure):
    _fields_ = [("split", C.c_float), ("packed", C.c_uint32)]


class VspgFieldRegion(C.Structure):
    _fields_ = [("pivot", f3), ("n_lobes", C.c_int32), ("weight", _fl), ("kappa", _fl), ("mu", _fl * 3),
                ("distance", _fl), ("vsp", _fl)]


class VspgField(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_regions", C.c_int32), ("nodes", C.POINTER(VspgKdNode)),
                ("regions", C.POINTER(VspgFieldRegion))]


class VspgTmajQuery(C.Structure):
    _fields_ = [("o", f3), ("d", f3), ("tMax", C.c_float), ("u", C.c_float),
                ("rng_a", C.c_float), ("rng_b", C.c_float), ("vsp", C.c_float),
                ("channel", C.c_int32), ("stop_after", C.c_int32)]


class VspgRayQuery(C.Structure):
    _fields_ = [("o", f3), ("d", f3), ("tMax", C.c_float), ("mode", C.c_int32), ("w", f3), ("tMax2", C.c_float)]


class VspgRayResult(C.Structure):
    _fields_ = [("hit", C.c_int32), ("prim", C.c_int32), ("t", C.c_float), ("p", f3), ("n", f3), ("o2", f3), ("d2", f3),
                ("hit2", C.c_int32), ("any2", C.c_int32), ("t2", C.c_float)]


RAY_QUERY_DTYPE = [("o", "<f4", 3), ("d", "<f4", 3), ("tMax", "<f4"), ("mode", "<i4"), ("w", "<f4", 3), ("tMax2", "<f4")]
RAY_RESULT_DTYPE = [("hit", "<i4"), ("prim", "<i4"), ("t", "<f4"), ("p", "<f4", 3), ("n", "<f4", 3), ("o2", "<f4", 3), ("d2", "<f4", 3),
                    ("hit2", "<i4"), ("any2", "<i4"), ("t2", "<f4")]


class VspgTmajResult(C.Structure):
    _fields_ = [("T_maj", f3), ("r_u_factor", f3), ("last_t", C.c_float), ("last_p", f3),
                ("n_callbacks", C.c_int32), ("sum_sigt_over_maj", C.c_float),
                ("vrc", C.c_float), ("majorant_scale", C.c_float)]


# every symbol include/vspg.h declares: (name, restype, argtypes)
_P = C.POINTER
_vp = C.c_void_p
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)   # VspgExchangeFn
SYMBOLS = [
    ("vspg_abi_version", C.c_int, []),
    ("vspg_last_error", C.c_char_p, []),
    ("vspg_integrator_params_default", None, [_P(VspgIntegratorParams)]),
    ("vspg_camera_look_at", C.c_int, [_P(VspgCamera), f3, f3, f3, C.c_float, C.c_int, C.c_int]),
    ("vspg_scene_fog_box", C.c_int, [_P(VspgScene), C.c_int, C.c_int]),
    ("vspg_transform_inverse", C.c_int, [C.c_float * 16, C.c_float * 16]),
    ("vspg_renderer_create", C.c_int, [_P(VspgScene), _P(VspgIntegratorParams), _P(VspgRenderConfig), _P(_vp)]),
    ("vspg_renderer_destroy", C.c_int, [_vp]),
    ("vspg_render_wave", C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    ("vspg_post_process_wave", C.c_int, [_vp, _vp]),
    ("vspg_isg_update_due", C.c_int, [_vp, C.c_int]),
    ("vspg_post_process_step", C.c_int, [_vp, C.c_int, _vp, _vp]),
    ("vspg_renderer_set_exchange", C.c_int, [_vp, _vp, _vp]),
    ("vspg_renderer_kernel_name", C.c_char_p, [_vp]),
    ("vspg_ray_batch", C.c_int, [_vp, C.c_int, _P(VspgRayQuery), _P(VspgRayResult), _vp]),
    ("vspg_renderer_set_arithmetic", C.c_int, [_vp, C.c_int]),
    ("vspg_renderer_get_arithmetic", C.c_int, [_vp]),
    ("vspg_flush", C.c_int, [_vp, _vp]),
    ("vspg_film_device_ptr", C.c_int, [_vp, _P(_vp), _P(C.c_size_t)]),
    ("vspg_film_read", C.c_int, [_vp, _P(C.c_float), _vp]),
    ("vspg_film_clear", C.c_int, [_vp, _vp]),
    ("vspg_vsp_buffer_device_ptr", C.c_int, [_vp, _P(_vp), _P(C.c_size_t)]),
    ("vspg_vsp_buffer_read", C.c_int, [_vp, _P(C.c_float), _P(C.c_int), _vp]),
    ("vspg_vsp_buffer_load", C.c_int, [_vp, _P(C.c_float), _vp]),
    ("vspg_isg_stats_device_ptr", C.c_int, [_vp, _P(_vp), _P(C.c_size_t)]),
    ("vspg_get_counters", C.c_int, [_vp, _P(VspgCounters), _vp]),
    ("vspg_reset_counters", C.c_int, [_vp, _vp]),
    ("vspg_trace_paths", C.c_int, [_vp, C.c_int, _P(C.c_int32), _P(C.c_int32), _P(C.c_float), _P(C.c_int32), _vp]),
    ("vspg_sample_tmaj_batch", C.c_int, [_vp, C.c_int, C.c_int, _P(VspgTmajQuery), _P(VspgTmajResult), _vp]),
    ("vspg_primitives_batch", C.c_int, [_vp, C.c_int, _P(C.c_float), _P(C.c_float), _P(C.c_uint64), _P(C.c_uint32), _P(C.c_float), _vp]),
    ("vspg_renderer_training_stats", C.c_int, [_vp, _P(VspgTrainStats), _vp]),
    ("vspg_train_samples_read", C.c_int, [_vp, _P(VspgTrainSample), C.c_size_t, _P(C.c_size_t), _vp]),
    ("vspg_renderer_get_guiding_field", C.c_int, [_vp, C.c_int, _P(VspgKdNode), _P(VspgFieldRegion), _P(C.c_int32),
                                                  _P(C.c_int32), _vp]),
    ("vspg_libm_batch", C.c_int, [_vp, C.c_int, _P(C.c_float), _P(C.c_float), _P(C.c_float), _P(C.c_float), _vp]),
    ("vspg_libm_log1m_batch", C.c_int, [_vp, C.c_int, _P(C.c_float), _P(C.c_float), _vp]),
    ("vspg_libm_powf_batch", C.c_int, [_vp, C.c_int, _P(C.c_float), _P(C.c_float), _P(C.c_float), _vp]),
    ("vspg_blackbody_batch", C.c_int, [_vp, C.c_int, _P(C.c_float), _P(C.c_float), _P(C.c_float), _vp]),
    ("vspg_renderer_get_tr_buffer", C.c_int, [_vp, _P(C.c_float), _P(C.c_int32), _vp]),
    ("vspg_renderer_set_tr_buffer", C.c_int, [_vp, _P(C.c_float), _vp]),
    ("vspg_renderer_set_guiding_field", C.c_int, [_vp, _P(VspgField), _P(VspgField), _vp]),
    ("vspg_guiding_query_batch", C.c_int, [_vp, C.c_int, C.c_float, C.c_int, _P(C.c_float), _P(C.c_float), _P(C.c_float),
                                          _P(C.c_float), _P(C.c_int32), _P(C.c_float), _P(C.c_float), _P(C.c_float),
                                          _P(C.c_float), _P(C.c_float), _vp]),
]

_lib = None


class VspgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("vspg error %d: %s" % (code, msg))
        self.code = code


def load():
    """Load csrc/libvspg_hip.so (built by __graft_entry__.build()).  Raises if it is missing:
    there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`"
                           % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _check(lib, rc):
    if rc != 0:
        raise VspgError(rc, (lib.vspg_last_error() or b"").decode())


def default_params():
    lib = load()
    p = VspgIntegratorParams()
    lib.vspg_integrator_params_default(C.byref(p))
    return p


def fog_box_scene(xres, yres):
    lib = load()
    s = VspgScene()
    _check(lib, lib.vspg_scene_fog_box(C.byref(s), xres, yres))
    return s


def procedural_cloud_density(n, seed=5, shape="noise"):
    """Seeded value noise, density in [0, ~1.3], ~25-30 % empty voxels, x fastest (the layout
    cmd/nanovdb2pbrt.cpp dumps for GridMedium): the heterogeneous stand-in of SURVEY.md 8d until a
    Disney-cloud asset is supplied.  shape="blob": the same noise inside a soft-edged ball of radius 0.36 n around the
    grid's centre and nothing outside it (~80 % empty voxels, empty majorant cells all around: what a real cloud in
    its bounding box looks like to the majorant grid)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    coarse = rng.random((n // 4 + 2,) * 3).astype(np.float32)
    idx = (np.arange(n, dtype=np.float32) + 0.5) / 4.0
    i0 = np.floor(idx).astype(int)
    f = (idx - i0).astype(np.float32)

    def interp(a, axis):
        shape = [1, 1, 1]
        shape[axis] = n
        w = f.reshape(shape)
        return np.take(a, i0, axis=axis) * (1 - w) + np.take(a, i0 + 1, axis=axis) * w

    v = interp(interp(interp(coarse, 0), 1), 2)
    v = np.clip(v * 2.2 - 0.75, 0.0, None)
    if shape == "blob":
        c = (np.arange(n, dtype=np.float32) + 0.5) / n - 0.5
        r = np.sqrt(c[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2)
        v = (v + 0.35) * np.clip((0.36 - r) / 0.06, 0.0, 1.0).astype(np.float32)
    return np.ascontiguousarray(v.transpose(2, 1, 0).reshape(-1).astype(np.float32))


def cloud_box_scene(xres, yres, n=256, sigma_t=8.0, albedo=0.99, g=0.877, seed=5, shape="noise"):
    """Fog-box geometry and light with the homogeneous fog replaced by a procedural n^3 GridMedium
    (sigma_t scale 8, albedo 0.99, g 0.877 -- SURVEY.md 8d's cloud-like choice)."""
    s = fog_box_scene(xres, yres)
    dens = procedural_cloud_density(n, seed, shape)
    m = s.medium
    m.type = MEDIUM_GRID
    m.sigma_a[:] = (sigma_t * (1 - albedo),) * 3
    m.sigma_s[:] = (sigma_t * albedo,) * 3
    m.g = g
    m.nx = m.ny = m.nz = n
    m.bounds_min[:] = (-0.9, -0.9, -0.6)
    m.bounds_max[:] = (0.9, 0.8, 0.9)
    m.density = dens.ctypes.data_as(C.POINTER(C.c_float))
    s._density_keepalive = dens
    return s


def nanovdb_box_scene(xres, yres, n=256, sigma_t=8.0, albedo=0.99, g=0.877, seed=5, density_offset=0.0, majorant_scale=1.0, shape="noise"):
    """Same cloud, handed over as a NanoVDBMedium (dense copy): index bbox [0, n-1]^3, cubic voxels, the world
    bounding box covers the voxel extents (what cmd/nanovdb2pbrt.cpp reads from the grid); 64^3 majorants."""
    s = cloud_box_scene(xres, yres, n, sigma_t, albedo, g, seed, shape)
    m = s.medium
    m.type = MEDIUM_NANOVDB
    ext = [m.bounds_max[k] - m.bounds_min[k] for k in range(3)]
    for k in range(3):
        m.index_min[k] = 0
        m.voxel_size[k] = ext[k] / n
        m.grid_origin[k] = m.bounds_min[k]
    m.density_offset = density_offset
    m.majorant_scale = majorant_scale
    return s


def add_quad(scene, p00, e1, e2, kd=(0.5, 0.5, 0.5), le=(0, 0, 0), material=0, iface=0, reverse=0, two_sided=0):
    """One more rectangle p00 + u e1 + v e2 (normal = normalize(e1 x e2), negated by `reverse`); material / iface: MATERIAL_*, IFACE_*."""
    k = scene.n_quads
    assert k < VSPG_MAX_QUADS
    q = scene.quads[k]
    q.p00[:] = p00
    q.e1[:] = e1
    q.e2[:] = e2
    q.Kd[:] = kd
    q.Le[:] = le
    q.two_sided, q.reverse_orientation, q.material, q.medium_interface = two_sided, reverse, material, iface
    scene.n_quads = k + 1
    return scene


def add_sphere(scene, center, radius, material=0, iface=0, kd=(0.5, 0.5, 0.5), scale=(1, 1, 1), reverse=0):
    """Shape "sphere" under Translate(center) * Scale(scale)."""
    import numpy as np
    k = scene.n_spheres
    assert k < VSPG_MAX_SPHERES
    sp = scene.spheres[k]
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[1, 1], m[2, 2] = scale
    m[0, 3], m[1, 3], m[2, 3] = center
    sp.render_from_object[:] = [float(x) for x in m.reshape(16)]
    # the inverse as the header and the scene-file reader form it (vspg_transform_inverse: pbrt's float Inverse(), transform.cpp):
    # the same scaled sphere gets the same matrix bits -- and so the same interval-arithmetic hits -- whichever way it came in
    inv = (C.c_float * 16)()
    lib = load()
    _check(lib, lib.vspg_transform_inverse(sp.render_from_object, inv))
    sp.object_from_render[:] = list(inv)
    sp.radius = radius
    sp.Kd[:] = kd
    sp.reverse_orientation, sp.material, sp.medium_interface = reverse, material, iface
    scene.n_spheres = k + 1
    return scene


def cloud_scene(xres, yres, n=256, sigma_t=8.0, albedo=0.99, g=0.877, seed=5, shape="noise", nvdb=False):
    """The shape of the reference's cloud scenes (BASELINE configs 3-5): the camera in VACUUM, the procedural n^3 cloud inside an
    interface-material bounding sphere (MediumInterface "cloud" "" + Material "interface"), a diffuse ground under it, a
    distant light and a uniform sky -- no closed box, no emissive geometry."""
    lib = load()
    s = VspgScene()
    _check(lib, lib.vspg_camera_look_at(C.byref(s.camera), f3(0.0, 0.6, -4.2), f3(0.0, 0.15, 0.0), f3(0, 1, 0), 38.0, xres, yres))
    dens = procedural_cloud_density(n, seed, shape)
    m = s.medium
    m.type = MEDIUM_NANOVDB if nvdb else MEDIUM_GRID
    m.sigma_a[:] = (sigma_t * (1 - albedo),) * 3
    m.sigma_s[:] = (sigma_t * albedo,) * 3
    m.g = g
    m.nx = m.ny = m.nz = n
    m.bounds_min[:] = (-0.8, -0.5, -0.8)
    m.bounds_max[:] = (0.8, 0.9, 0.8)
    m.density = dens.ctypes.data_as(C.POINTER(C.c_float))
    s._density_keepalive = dens
    if nvdb:
        for k in range(3):
            m.index_min[k] = 0
            m.voxel_size[k] = (m.bounds_max[k] - m.bounds_min[k]) / n
            m.grid_origin[k] = m.bounds_min[k]
        m.density_offset = 0.0
        m.majorant_scale = 1.0
    s.camera_outside_medium = 1
    add_sphere(s, (0.0, 0.2, 0.0), 1.34, material=MATERIAL_INTERFACE, iface=IFACE_INSIDE)
    add_quad(s, (-6, -1.2, -6), (0, 0, 12), (12, 0, 0), kd=(0.4, 0.35, 0.3))   # ground, n = +y
    add_infinite_light(s, LIGHT_DISTANT, (6.0, 5.5, 5.0), (0.4, 0.8, -0.3))
    add_infinite_light(s, LIGHT_UNIFORM_INFINITE, (0.25, 0.35, 0.5))
    return s


def set_medium_transform(scene, m):
    """renderFromMedium = the row-major 4x4 `m` (affine); the inverse is filled by vspg_transform_inverse."""
    import numpy as np
    lib = load()
    a = (C.c_float * 16)(*[float(x) for x in np.asarray(m, dtype=np.float32).reshape(16)])
    inv = (C.c_float * 16)()
    _check(lib, lib.vspg_transform_inverse(a, inv))
    scene.medium.has_transform = 1
    scene.medium.render_from_medium[:] = list(a)
    scene.medium.medium_from_render[:] = list(inv)
    return scene


def add_infinite_light(scene, kind, L, w_light=(0.0, 1.0, 0.0)):
    """kind: LIGHT_UNIFORM_INFINITE (sky, radiance L) or LIGHT_DISTANT (sun, radiance L from direction w_light, normalised here)."""
    import numpy as np
    k = scene.n_infinite_lights
    assert k < VSPG_MAX_INFINITE_LIGHTS
    w = np.asarray(w_light, dtype=np.float64)
    w = (w / np.linalg.norm(w)).astype(np.float32)
    il = scene.infinite_lights[k]
    il.type = kind
    il.L[:] = [float(x) for x in L]
    il.w_light[:] = [float(x) for x in w]
    scene.n_infinite_lights = k + 1
    return scene


def set_triangles(scene, tri_p, tri_kd=None):
    """Attach a triangle soup (n x 3 x 3 vertex array, optional n x 3 diffuse reflectances) to the scene."""
    import numpy as np
    tp = np.ascontiguousarray(tri_p, dtype=np.float32).reshape(-1, 9)
    scene.n_triangles = tp.shape[0]
    scene.tri_p = tp.ctypes.data_as(C.POINTER(C.c_float))
    scene._tri_keepalive = [tp]
    if tri_kd is not None:
        tk = np.ascontiguousarray(tri_kd, dtype=np.float32).reshape(-1, 3)
        assert tk.shape[0] == tp.shape[0]
        scene.tri_kd = tk.ctypes.data_as(C.POINTER(C.c_float))
        scene._tri_keepalive.append(tk)
    return scene


def app_f_params():
    """SURVEY.md App. F integrator line: primary-ray VSP guiding only."""
    p = default_params()
    p.surfaceguiding = 0
    p.volumeguiding = 0
    p.vspsecondaryguiding = 0
    return p


class Renderer:
    """Thin RAII wrapper over the vspg_renderer_* entry points."""

    def __init__(self, scene, params, xres, yres, spp=1, seed=0, shard_index=0, shard_count=1, device=0):
        self.lib = load()
        self.cfg = VspgRenderConfig(xres, yres, spp, seed, shard_index, shard_count, device)
        self.scene, self.params = scene, params
        self.h = _vp()
        _check(self.lib, self.lib.vspg_renderer_create(C.byref(scene), C.byref(params), C.byref(self.cfg),
                                                       C.byref(self.h)))
        self.xres, self.yres = xres, yres

    def close(self):
        if getattr(self, "h", None):
            self.lib.vspg_renderer_destroy(self.h)
            self.h = None

    __del__ = close

    def render_wave(self, w0, w1, stream=None):
        _check(self.lib, self.lib.vspg_render_wave(self.h, w0, w1, _vp(stream or 0)))

    def isg_update_due(self, n_waves=1):
        return bool(self.lib.vspg_isg_update_due(self.h, int(n_waves)))

    def post_process_step(self, n_waves, isg_stats_sum_ptr=None, stream=None):
        """PostProcessWave after a step of n_waves sample indices; isg_stats_sum_ptr = device pointer (int) to the
        all-reduced statistics or None."""
        _check(self.lib, self.lib.vspg_post_process_step(self.h, int(n_waves), _vp(isg_stats_sum_ptr or 0), _vp(stream or 0)))

    def set_exchange(self, fn):
        """Sharded guiding-field training: fn(dev_ptr: int, n_floats: int, stream: int) sums the floats in place over the
        ranks (VspgExchangeFn); None removes the hook.  Exceptions in fn surface as an error of post_process_step."""
        if fn is None:
            self._exchange_cb = None
            _check(self.lib, self.lib.vspg_renderer_set_exchange(self.h, None, None))
            return

        def _cb(ptr, n, stream, _user):
            try:
                fn(int(ptr or 0), int(n), int(stream or 0))
                return 0
            except Exception:  # the C side turns a non-zero return into an error code
                import traceback
                traceback.print_exc()
                return VSPG_EHIP
        self._exchange_cb = EXCHANGE_FN(_cb)   # keep the thunk alive as long as the renderer may call it
        _check(self.lib, self.lib.vspg_renderer_set_exchange(self.h, C.cast(self._exchange_cb, _vp), None))

    def kernel_name(self):
        return (self.lib.vspg_renderer_kernel_name(self.h) or b"").decode()

    def set_arithmetic(self, mode):
        """ARITH_EXACT (default) / ARITH_FAST_WEIGHTS / ARITH_FAST (include/vspg.h, csrc/vspg_arith.h)."""
        _check(self.lib, self.lib.vspg_renderer_set_arithmetic(self.h, int(mode)))

    def arithmetic(self):
        return int(self.lib.vspg_renderer_get_arithmetic(self.h))

    def post_process_wave(self, stream=None):
        _check(self.lib, self.lib.vspg_post_process_wave(self.h, _vp(stream or 0)))

    def flush(self, stream=None):
        """Adds the samples the last one-sample wave parked to the film and the image-space statistics (asynchronous on
        `stream`): what a host that keeps film_ptr() / isg_stats_ptr() across waves calls before it reads through them."""
        _check(self.lib, self.lib.vspg_flush(self.h, _vp(stream or 0)))

    def film_ptr(self):
        p, n = _vp(), C.c_size_t()
        _check(self.lib, self.lib.vspg_film_device_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def isg_stats_ptr(self):
        p, n = _vp(), C.c_size_t()
        _check(self.lib, self.lib.vspg_isg_stats_device_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def vsp_ptr(self):
        p, n = _vp(), C.c_size_t()
        _check(self.lib, self.lib.vspg_vsp_buffer_device_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def film(self, stream=None):
        import numpy as np
        out = np.empty((self.yres, self.xres, 4), dtype=np.float32)
        _check(self.lib, self.lib.vspg_film_read(self.h, out.ctypes.data_as(_P(C.c_float)), _vp(stream or 0)))
        return out

    def film_clear(self, stream=None):
        _check(self.lib, self.lib.vspg_film_clear(self.h, _vp(stream or 0)))

    def vsp_buffer(self, stream=None):
        import numpy as np
        out = np.empty((self.yres, self.xres), dtype=np.float32)
        ready = C.c_int()
        _check(self.lib, self.lib.vspg_vsp_buffer_read(self.h, out.ctypes.data_as(_P(C.c_float)), C.byref(ready),
                                                      _vp(stream or 0)))
        return out, bool(ready.value)

    def load_vsp_buffer(self, vsp, stream=None):
        import numpy as np
        v = np.ascontiguousarray(vsp, dtype=np.float32)
        assert v.shape == (self.yres, self.xres)
        _check(self.lib, self.lib.vspg_vsp_buffer_load(self.h, v.ctypes.data_as(_P(C.c_float)), _vp(stream or 0)))

    def tr_buffer(self, stream=None):
        """TrBuffer read-back: (rgb (H, W, 3) float32, spp (H, W) int32)."""
        import numpy as np
        rgb = np.empty((self.yres, self.xres, 3), dtype=np.float32)
        spp = np.empty((self.yres, self.xres), dtype=np.int32)
        _check(self.lib, self.lib.vspg_renderer_get_tr_buffer(self.h, rgb.ctypes.data_as(_P(C.c_float)),
                                                             spp.ctypes.data_as(_P(C.c_int32)), _vp(stream or 0)))
        return rgb, spp

    def set_tr_buffer(self, rgb, stream=None):
        import numpy as np
        v = np.ascontiguousarray(rgb, dtype=np.float32)
        assert v.shape == (self.yres, self.xres, 3)
        _check(self.lib, self.lib.vspg_renderer_set_tr_buffer(self.h, v.ctypes.data_as(_P(C.c_float)), _vp(stream or 0)))

    def libm_powf_batch(self, x, y):
        import numpy as np
        x = np.ascontiguousarray(x, dtype=np.float32)
        y = np.ascontiguousarray(y, dtype=np.float32)
        out = np.empty_like(x)
        fp = _P(C.c_float)
        _check(self.lib, self.lib.vspg_libm_powf_batch(self.h, x.shape[0], x.ctypes.data_as(fp), y.ctypes.data_as(fp),
                                                      out.ctypes.data_as(fp), _vp(0)))
        return out

    def blackbody_batch(self, u, T):
        """[n, 6]: SampleVisible(u)'s three wavelengths and BlackbodySpectrum(T).Sample at them, as the kernels evaluate them."""
        import numpy as np
        u = np.ascontiguousarray(u, dtype=np.float32)
        T = np.ascontiguousarray(T, dtype=np.float32)
        out = np.empty((u.shape[0], 6), dtype=np.float32)
        fp = _P(C.c_float)
        _check(self.lib, self.lib.vspg_blackbody_batch(self.h, u.shape[0], u.ctypes.data_as(fp), T.ctypes.data_as(fp),
                                                      out.ctypes.data_as(fp), _vp(0)))
        return out

    def counters(self, stream=None):
        c = VspgCounters()
        _check(self.lib, self.lib.vspg_get_counters(self.h, C.byref(c), _vp(stream or 0)))
        return c.as_dict()

    def reset_counters(self, stream=None):
        _check(self.lib, self.lib.vspg_reset_counters(self.h, _vp(stream or 0)))

    def trace_paths(self, pixel_xy, sample_index):
        import numpy as np
        pix = np.ascontiguousarray(pixel_xy, dtype=np.int32).reshape(-1, 2)
        si = np.ascontiguousarray(sample_index, dtype=np.int32).reshape(-1)
        n = si.shape[0]
        L = np.empty((n, 3), dtype=np.float32)
        seg = np.empty(n, dtype=np.int32)
        _check(self.lib, self.lib.vspg_trace_paths(self.h, n, pix.ctypes.data_as(_P(C.c_int32)),
                                                   si.ctypes.data_as(_P(C.c_int32)),
                                                   L.ctypes.data_as(_P(C.c_float)),
                                                   seg.ctypes.data_as(_P(C.c_int32)), _vp(0)))
        return L, seg

    def ray_batch(self, queries):
        """queries: numpy structured array of RAY_QUERY_DTYPE; returns one of RAY_RESULT_DTYPE (vspg_ray_batch)."""
        import numpy as np
        q = np.ascontiguousarray(queries, dtype=RAY_QUERY_DTYPE)
        out = np.zeros(len(q), dtype=RAY_RESULT_DTYPE)
        assert q.itemsize == C.sizeof(VspgRayQuery) and out.itemsize == C.sizeof(VspgRayResult)
        _check(self.lib, self.lib.vspg_ray_batch(self.h, len(q), q.ctypes.data_as(_P(VspgRayQuery)), out.ctypes.data_as(_P(VspgRayResult)), None))
        return out

    def sample_tmaj_batch(self, variant, queries):
        n = len(queries)
        q = (VspgTmajQuery * n)(*queries)
        out = (VspgTmajResult * n)()
        _check(self.lib, self.lib.vspg_sample_tmaj_batch(self.h, variant, n, q, out, _vp(0)))
        return list(out)

    def primitives_batch(self, f, g):
        import numpy as np
        f = np.ascontiguousarray(f, dtype=np.float32)
        g = np.ascontiguousarray(g, dtype=np.float32)
        n = f.shape[0]
        h = np.empty(n, dtype=np.uint64)
        r = np.empty(n, dtype=np.uint32)
        e = np.empty(n, dtype=np.float32)
        _check(self.lib, self.lib.vspg_primitives_batch(self.h, n, f.ctypes.data_as(_P(C.c_float)),
                                                        g.ctypes.data_as(_P(C.c_float)),
                                                        h.ctypes.data_as(_P(C.c_uint64)),
                                                        r.ctypes.data_as(_P(C.c_uint32)),
                                                        e.ctypes.data_as(_P(C.c_float)), _vp(0)))
        return h, r, e

    def libm_batch(self, x):
        import numpy as np
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[0]
        lo, so, co = (np.empty(n, dtype=np.float32) for _ in range(3))
        fp = _P(C.c_float)
        _check(self.lib, self.lib.vspg_libm_batch(self.h, n, x.ctypes.data_as(fp), lo.ctypes.data_as(fp),
                                                  so.ctypes.data_as(fp), co.ctypes.data_as(fp), _vp(0)))
        return lo, so, co

    # ---- guiding-cache training (a18)
    def training_stats(self):
        st = VspgTrainStats()
        _check(self.lib, self.lib.vspg_renderer_training_stats(self.h, C.byref(st), _vp(0)))
        return st.as_dict()

    def train_samples(self):
        import numpy as np
        n = C.c_size_t(0)
        _check(self.lib, self.lib.vspg_train_samples_read(self.h, None, 0, C.byref(n), _vp(0)))
        buf = (VspgTrainSample * max(1, n.value))()
        _check(self.lib, self.lib.vspg_train_samples_read(self.h, buf, n.value, C.byref(n), _vp(0)))
        return np.frombuffer(buf, dtype=TRAIN_SAMPLE_DTYPE, count=n.value).copy()

    def get_guiding_field(self, volume):
        nn, nr = C.c_int32(0), C.c_int32(0)
        _check(self.lib, self.lib.vspg_renderer_get_guiding_field(self.h, int(volume), None, None, C.byref(nn), C.byref(nr), _vp(0)))
        nodes = (VspgKdNode * max(1, nn.value))()
        regs = (VspgFieldRegion * max(1, nr.value))()
        _check(self.lib, self.lib.vspg_renderer_get_guiding_field(self.h, int(volume), nodes, regs, C.byref(nn), C.byref(nr), _vp(0)))
        return nodes, regs, nn.value, nr.value

    def libm_log1m_batch(self, x):
        import numpy as np
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        fp = _P(C.c_float)
        _check(self.lib, self.lib.vspg_libm_log1m_batch(self.h, x.shape[0], x.ctypes.data_as(fp), out.ctypes.data_as(fp), _vp(0)))
        return out

    def set_guiding_field(self, surface, volume, stream=None):
        _check(self.lib, self.lib.vspg_renderer_set_guiding_field(self.h, C.byref(surface.pod) if surface else None,
                                                                  C.byref(volume.pod) if volume else None, _vp(stream or 0)))


def _guiding_query(self, is_volume, g, p, n_or_wo, wi, u):
    import numpy as np
    fp = _P(C.c_float)
    p, a, wi, u = (np.ascontiguousarray(x, dtype=np.float32) for x in (p, n_or_wo, wi, u))
    n = p.shape[0]
    ok = np.zeros(n, dtype=np.int32)
    pdf, inc, vsp, pdfs = (np.zeros(n, dtype=np.float32) for _ in range(4))
    ws = np.zeros((n, 3), dtype=np.float32)
    _check(self.lib, self.lib.vspg_guiding_query_batch(self.h, int(is_volume), float(g), n, p.ctypes.data_as(fp),
                                                       a.ctypes.data_as(fp), wi.ctypes.data_as(fp), u.ctypes.data_as(fp),
                                                       ok.ctypes.data_as(_P(C.c_int32)), pdf.ctypes.data_as(fp),
                                                       inc.ctypes.data_as(fp), vsp.ctypes.data_as(fp), ws.ctypes.data_as(fp),
                                                       pdfs.ctypes.data_as(fp), _vp(0)))
    return dict(ok=ok, pdf=pdf, incoming_pdf=inc, vsp=vsp, ws=ws, pdf_s=pdfs)


Renderer.guiding_query_batch = _guiding_query


class Field:
    """Host-side container of a guiding field (kd-tree nodes + regions) keeping the arrays alive."""

    def __init__(self, nodes, regions):
        self.nodes = (VspgKdNode * len(nodes))(*nodes)
        self.regions = (VspgFieldRegion * len(regions))(*regions)
        self.pod = VspgField(len(nodes), len(regions), self.nodes, self.regions)
