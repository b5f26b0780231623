"""CPU tests of the oracle's round-2 callers of the path: triangle geometry, infinite lights, placed media.
shapes.cpp / lights.cpp cannot be compiled here (the full pbrt build is needed), so these restatements are pinned by
analytic answers and by cross-checks between independent restatements -- no GPU involved."""
import ctypes as C

import numpy as np

import oracle_lib
from scenes import box_wall_triangles, cloud_density, grid_scene, load_package


def _paths(W, H, n, seed):
    rng = np.random.default_rng(seed)
    pix = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 4096, n).astype(np.int32)
    return pix, si


def _only_quad(P, scene, index):
    q = type(scene.quads[index]).from_buffer_copy(scene.quads[index])
    for i in range(P.VSPG_MAX_QUADS):
        scene.quads[i] = type(q)()
    scene.quads[0] = q
    scene.n_quads = 1
    return scene


def test_triangle_box_agrees_with_rectangle_box():
    """IntersectTriangle / InteractionFromIntersection (shapes.cpp:168-262, shapes.h:930-1010) against the rectangle code
    (BilinearPatch's planar case): the box's walls as 12 triangles give the same picture.  Hit points differ in the last
    bits (b0 p0 + b1 p1 + b2 p2 vs p00 + u e1 + v e2), which re-seeds the shadow-ray RNG: the agreement is statistical for
    radiance, near-exact for path structure."""
    P = load_package()
    W, H = 48, 36
    prm = P.app_f_params()
    tri_scene = _only_quad(P, P.fog_box_scene(W, H), 6)   # the emissive rectangle stays; the walls come as triangles
    tris, kd = box_wall_triangles()
    P.set_triangles(tri_scene, tris, kd)
    t = oracle_lib.OracleRenderer(tri_scene, prm, W, H, seed=1)
    r = oracle_lib.OracleRenderer(P.fog_box_scene(W, H), prm, W, H, seed=1)
    pix, si = _paths(W, H, 6000, 31)
    Lt, st = t.trace_paths(pix, si)
    Lr, sr = r.trace_paths(pix, si)
    t.close(); r.close()
    assert np.isfinite(Lt).all()
    assert np.mean(st == sr) > 0.9
    assert abs(Lt.mean() / Lr.mean() - 1) < 0.05


def test_distant_light_on_a_floor_is_lambert():
    """DistantLight NEE (lights.h:320-327, the delta-light weighting of :1248-1249) on a diffuse floor, no medium, one bounce:
    every camera ray that hits the floor returns exactly Kd / pi * L * cos(theta) -- the light sample is deterministic."""
    P = load_package()
    W, H = 32, 24
    scene = _only_quad(P, P.fog_box_scene(W, H), 0)
    scene.medium.type = P.MEDIUM_NONE
    q = scene.quads[0]
    n = np.cross(np.array(list(q.e1)), np.array(list(q.e2)))
    n = n / np.linalg.norm(n)
    kd = np.array(list(q.Kd))
    assert not any(q.Le)
    L = np.array([3.0, 2.0, 1.0])
    w = np.array([0.3, 1.0, -0.4]); w = w / np.linalg.norm(w)
    P.add_infinite_light(scene, P.LIGHT_DISTANT, L, w)
    prm = P.app_f_params()
    prm.maxdepth = 1
    prm.lightsampler = 0
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=3)
    pix, si = _paths(W, H, 3000, 5)
    Lc, sc = c.trace_paths(pix, si)
    c.close()
    expect = kd / np.pi * L * abs(float(n @ w))
    floor = sc == 2                              # camera ray hit the floor, the bounce ray escaped
    assert floor.mean() > 0.03 and np.all((sc == 1) | floor)
    assert np.allclose(Lc[floor], expect[None, :], rtol=2e-5, atol=0)
    # camera rays that miss the floor escape at depth 0, where this fork's escaped-ray loop lists DeltaDirection lights too
    # (integrators.h:79) and DistantLight::Le returns scale * Lemit for ANY ray (lights.h:291-293): restated as is
    assert np.all(Lc[~floor] == L[None, :].astype(np.float32))


def test_uniform_sky_furnace_through_a_cloud():
    """UniformInfiniteLight reached by escaping rays (:353-374) through a purely scattering cloud (albedo 1) inside an
    interface-material sphere, nothing else in the scene: radiance 1 from every direction is a fixed point of the transport
    equation, so the estimator's mean is 1 -- this exercises the resampling routine, the placed-medium transform, the boundary
    crossings (:399-404; a ray that escapes is not sampled, :318, so an open cloud would never be traversed), Russian roulette
    and the escaped-ray weights together.  (Paths cut at maxdepth lose their energy: maxdepth is far above the mean path length.)"""
    P = load_package()
    W, H = 32, 24
    dens = cloud_density(16)
    scene = grid_scene(dens, (16, 16, 16), 0.0, 6.0, g=0.5, bmin=(-0.7, -0.6, -0.5), bmax=(0.7, 0.6, 0.7), W=W, H=H)
    for i in range(P.VSPG_MAX_QUADS):
        scene.quads[i] = type(scene.quads[0])()
    scene.n_quads = 0
    from scenes import add_sphere
    add_sphere(scene, (0.0, 0.0, 0.1), 1.45, material=P.MATERIAL_INTERFACE, iface=P.IFACE_INSIDE)
    scene.camera_outside_medium = 0      # the eye (0, 0, -0.95) is inside the sphere: one crossing on the way out
    M = np.eye(4, dtype=np.float32)
    a = 0.4
    M[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]) @ np.diag([0.9, 1.1, 0.8])
    M[:3, 3] = (0.05, -0.1, 0.1)
    P.set_medium_transform(scene, M)
    P.add_infinite_light(scene, P.LIGHT_UNIFORM_INFINITE, (1.0, 1.0, 1.0))
    prm = P.app_f_params()
    prm.maxdepth = 200
    prm.lightsampler = 0
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=7)
    pix, si = _paths(W, H, 12000, 9)
    Lc, sc = c.trace_paths(pix, si)
    c.close()
    assert np.isfinite(Lc).all()
    assert sc.max() < 200 and sc.mean() > 2.3     # the cloud is really traversed (a crossing + the scatterings), no path reaches the cut
    m = Lc.mean(axis=0)
    se = Lc.std(axis=0) / np.sqrt(len(Lc))
    print("furnace mean", m, "+-", se)
    assert np.all(np.abs(m - 1.0) < np.maximum(4 * se, 0.01))
