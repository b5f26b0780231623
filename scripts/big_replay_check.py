"""One-off confidence check (not part of the test suite: minutes of oracle time): replay many (pixel, sample) paths on the GPU and on
the CPU oracle and count bit-identical radiances -- 1080p scenes of the bench workloads plus triangle / sky / guided variants."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
import oracle_lib, scenes
P = load_package(); P.load()
W, H = 1920, 1080
rng = np.random.default_rng(2026)

def check(name, scene, prm, n, field=None):
    g = P.Renderer(scene, prm, W, H, seed=11)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=11)
    if field is not None:
        g.set_guiding_field(field, field); c.set_guiding_field(field, field)
    pix = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 1 << 20, n).astype(np.int32)
    t0 = time.time()
    Lg, sg = g.trace_paths(pix, si)
    Lc, sc = c.trace_paths(pix, si)
    same = np.all(Lg.view(np.uint32) == Lc.view(np.uint32), axis=1) & (sg == sc)
    print("%-34s %8d paths: bit-identical %d (%.6f), mean segments %.2f, %.1f s" % (name, n, same.sum(), same.mean(), sc.mean(), time.time() - t0), flush=True)
    g.close(); c.close()
    return bool(same.all())

ok = True
fog = P.fog_box_scene(W, H)
ok &= check("fog (App. F options)", fog, P.app_f_params(), 2000000)
chro = P.fog_box_scene(W, H)
chro.medium.sigma_a[:] = (0.3, 0.1, 0.02); chro.medium.sigma_s[:] = (0.2, 0.9, 1.6); chro.medium.g = 0.6
ok &= check("chromatic anisotropic fog", chro, P.app_f_params(), 500000)
ok &= check("cloud 256^3 (GridMedium)", P.cloud_box_scene(W, H, 256), P.app_f_params(), 300000)
ok &= check("cloud 256^3 (NanoVDB semantics)", P.nanovdb_box_scene(W, H, 256), P.app_f_params(), 200000)
field = scenes.light_field(P, n=4)
ok &= check("fog, reference-default guiding", fog, P.default_params(), 300000, field)
ok &= check("cloud, reference-default guiding", P.cloud_box_scene(W, H, 256), P.default_params(), 100000, field)
tri = P.fog_box_scene(W, H)
t, kd = scenes.heightfield_triangles(100)
P.set_triangles(tri, t, kd)
P.add_infinite_light(tri, P.LIGHT_UNIFORM_INFINITE, (0.35, 0.5, 0.9))
P.add_infinite_light(tri, P.LIGHT_DISTANT, (9.0, 8.0, 6.5), (0.3, 1.0, -0.4))
prm = P.app_f_params(); prm.lightsampler = 0
ok &= check("fog + 20k triangles + sky + sun", tri, prm, 60000)
# round 3: five lights of very different power (dim floor, bright one-sided panel, two-sided panel, sky, sun) under the BVH and the
# power light samplers, fog with the reference's default guiding on top
def multi_light():
    sc = P.fog_box_scene(W, H)
    for k in range(3):
        sc.medium.sigma_a[k] = 0.02; sc.medium.sigma_s[k] = 0.25
    floor = type(sc.quads[0]).from_buffer_copy(sc.quads[0])
    for i in range(P.VSPG_MAX_QUADS):
        sc.quads[i] = type(floor)()
    sc.quads[0] = floor
    sc.quads[0].Le[:] = (0.25, 0.2, 0.15)

    def panel(k, p00, e1, e2, Le, two_sided):
        q = type(floor)()
        q.p00[:], q.e1[:], q.e2[:] = p00, e1, e2
        q.Kd[:] = (0.5, 0.5, 0.5); q.Le[:] = Le; q.two_sided = two_sided
        sc.quads[k] = q
    panel(1, (-0.15, 0.7, 0.1), (0.3, 0, 0), (0, 0, 0.3), (30.0, 24.0, 12.0), 0)
    panel(2, (-0.75, -0.2, 0.4), (0, 0.5, 0), (0, 0, 0.4), (2.0, 4.0, 9.0), 1)
    sc.n_quads = 3
    t2, kd2 = scenes.heightfield_triangles(40, y=-0.75, amp=0.2)
    P.set_triangles(sc, t2, kd2)
    P.add_infinite_light(sc, P.LIGHT_UNIFORM_INFINITE, (0.35, 0.5, 0.9))
    P.add_infinite_light(sc, P.LIGHT_DISTANT, (9.0, 8.0, 6.5), (0.3, 1.0, -0.4))
    return sc
ml = multi_light()
for name, ls in (("bvh", P.LIGHTSAMPLER_BVH), ("power", P.LIGHTSAMPLER_POWER)):
    prm = P.app_f_params(); prm.lightsampler = ls
    ok &= check("five lights, %s light sampler" % name, ml, prm, 60000)
prm = P.default_params(); prm.lightsampler = P.LIGHTSAMPLER_BVH
ok &= check("five lights, bvh sampler, guided", ml, prm, 40000, scenes.light_field(P, n=4, light=(0.3, 5.0, -0.4)))
# round 4: medium boundaries (the reference's cloud-scene shape: camera in vacuum, interface sphere, ground, sun + sky), an interface box of
# quads around a fog, spheres as diffuse objects, and temperature grids (blackbody emission) under "nds"
import ctypes as C
cs = P.cloud_scene(W, H, 256)
ok &= check("cloud scene (boundaries), App. F", cs, P.app_f_params(), 300000)
ok &= check("cloud scene, reference defaults", cs, P.default_params(), 100000, scenes.light_field(P, n=4, bmin=(-3, -3, -3), bmax=(3, 3, 3), light=(0.0, 2.9, 0.0)))
ok &= check("cloud scene, NanoVDB semantics", P.cloud_scene(W, H, 256, nvdb=True), P.app_f_params(), 200000)
nds = P.app_f_params(); nds.vspsamplingmethod = P.VSP_NDS
ok &= check("cloud scene under nds", cs, nds, 200000)
fire = P.cloud_scene(W, H, 256, nvdb=True)
temp = (150.0 + 2600.0 * np.clip(P.procedural_cloud_density(256, seed=11), 0, 1.4)).astype(np.float32)
fire.medium.temperature = temp.ctypes.data_as(C.POINTER(C.c_float))
fire.medium.temperature_offset, fire.medium.temperature_scale, fire.medium.nvdb_le_scale = 120.0, 1.3, 40.0
ok &= check("burning cloud scene (blackbody), nds", fire, nds, 200000)
gfire = P.cloud_box_scene(W, H, 256)
gfire.medium.temperature = temp.ctypes.data_as(C.POINTER(C.c_float))
gfire.medium.temperature_offset, gfire.medium.temperature_scale = 120.0, 1.3
ok &= check("burning GridMedium box (blackbody), nds", gfire, nds, 200000)
sph = P.fog_box_scene(W, H)
P.add_sphere(sph, (0.3, -0.5, 0.2), 0.35)
P.add_sphere(sph, (-0.4, 0.1, -0.1), 0.25, material=P.MATERIAL_INTERFACE, iface=P.IFACE_OUTSIDE)   # a hollow bubble in the fog
ok &= check("fog + a diffuse sphere + a hollow bubble", sph, P.app_f_params(), 300000)
print("ALL BIT-IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
