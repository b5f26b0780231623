"""Scene builders shared by the CPU and GPU tests (inputs only)."""
import ctypes as C

import numpy as np

import oracle_lib
from conftest import load_package


def d3_density():
    """8^3 density = RNG(7).Uniform<float>() in x-fastest order."""
    lib = oracle_lib.load()
    f = (C.c_float * 512)()
    lib.oracle_rng_seq(7, 0, 0, 0, 512, None, f)
    return np.array(f, dtype=np.float32)


def grid_scene(density, n, sigma_a, sigma_s, g=0.0, bmin=(0, 0, 0), bmax=(1, 1, 1), W=16, H=16):
    """fog-box geometry with the homogeneous fog replaced by a GridMedium ("uniformgrid")."""
    P = load_package()
    s = oracle_lib.fog_box_scene(W, H)
    m = s.medium
    m.type = P.MEDIUM_GRID
    m.sigma_a[:] = (sigma_a,) * 3 if np.isscalar(sigma_a) else tuple(sigma_a)
    m.sigma_s[:] = (sigma_s,) * 3 if np.isscalar(sigma_s) else tuple(sigma_s)
    m.g = g
    m.nx, m.ny, m.nz = n
    m.bounds_min[:] = bmin
    m.bounds_max[:] = bmax
    assert density.dtype == np.float32 and density.size == n[0] * n[1] * n[2]
    m.density = density.ctypes.data_as(C.POINTER(C.c_float))
    s._density_keepalive = density
    return s


def cloud_density(n, seed=5):
    """seeded value-noise-like density in [0, 1.3], ~25 % empty voxels (stand-in for the Disney cloud)."""
    rng = np.random.default_rng(seed)
    coarse = rng.random((n // 4 + 2,) * 3)
    idx = (np.arange(n) + 0.5) / 4.0
    i0 = np.floor(idx).astype(int)
    f = idx - i0

    def interp(a, axis):
        a0 = np.take(a, i0, axis=axis)
        a1 = np.take(a, i0 + 1, axis=axis)
        shape = [1, 1, 1]
        shape[axis] = n
        w = f.reshape(shape)
        return a0 * (1 - w) + a1 * w

    v = interp(interp(interp(coarse, 0), 1), 2)
    v = np.clip(v * 2.2 - 0.75, 0.0, None)
    return np.ascontiguousarray(v.transpose(2, 1, 0).reshape(-1).astype(np.float32))  # x fastest


def light_field(P, n=4, bmin=(-1, -1, -1), bmax=(1, 1, 1), light=(0.0, 0.999, 0.0), vsp_lo=0.25, vsp_hi=0.8, seed=0):
    """A hand-made guiding field for the fog box: balanced kd-tree with n^3 leaves; every region has
    a sharp parallax-aware lobe towards the ceiling light, a broad lobe up and a near-uniform lobe.
    Stands in for a trained OpenPGL field in the query-side tests."""
    rng = np.random.default_rng(seed)
    nodes, regions = [], []

    def build(lo, hi, depth):
        idx = len(nodes)
        nodes.append(None)
        if depth == 3 * int(np.log2(n)):
            c = [(a + b) / 2 for a, b in zip(lo, hi)]
            R = P.VspgFieldRegion()
            R.pivot[:] = c
            R.n_lobes = 3
            d = np.array(light) - np.array(c)
            dist = float(np.linalg.norm(d))
            d = d / dist
            lobes = [(0.55, 25.0 + 10 * rng.random(), d, dist, vsp_lo), (0.3, 1.5, np.array([0.0, 1.0, 0.0]), np.inf, 0.5),
                     (0.15, 0.05, np.array([1.0, 0.0, 0.0]), np.inf, vsp_hi)]
            for k, (w, kap, mu, dd, vsp) in enumerate(lobes):
                R.weight[k] = w
                R.kappa[k] = kap
                for a in range(3):
                    R.mu[a][k] = float(mu[a])
                R.distance[k] = dd
                R.vsp[k] = vsp
            nodes[idx] = P.VspgKdNode(0.0, 3 | (len(regions) << 2))
            regions.append(R)
            return idx
        axis = depth % 3
        mid = (lo[axis] + hi[axis]) / 2
        # children must be adjacent: reserve both slots, then fill
        left = len(nodes)
        nodes.append(None)
        nodes.append(None)

        def fill(slot, lo2, hi2):
            sub_first = len(nodes)
            sub = build(lo2, hi2, depth + 1)  # appended at the end
            nodes[slot] = nodes[sub]
            # the subtree root was appended at `sub`; move it into the reserved slot
            nodes.pop(sub)
            # fix child indices that pointed past the removed entry
            for i in range(sub_first, len(nodes)):
                nd = nodes[i]
                if nd is not None and (nd.packed & 3) != 3 and (nd.packed >> 2) > sub:
                    nodes[i] = P.VspgKdNode(nd.split, (nd.packed & 3) | (((nd.packed >> 2) - 1) << 2))
            nd = nodes[slot]
            if (nd.packed & 3) != 3 and (nd.packed >> 2) > sub:
                nodes[slot] = P.VspgKdNode(nd.split, (nd.packed & 3) | (((nd.packed >> 2) - 1) << 2))

        hi_l = list(hi); hi_l[axis] = mid
        lo_r = list(lo); lo_r[axis] = mid
        fill(left, list(lo), hi_l)
        fill(left + 1, lo_r, list(hi))
        nodes[idx] = P.VspgKdNode(float(mid), axis | (left << 2))
        return idx

    build(list(bmin), list(bmax), 0)
    return P.Field(nodes, regions)


def nvdb_scene(density, n, sigma_a, sigma_s, g=0.0, index_min=(0, 0, 0), voxel=(0.1, 0.1, 0.1), origin=(-0.8, -0.8, -0.5),
               density_offset=0.0, majorant_scale=1.0, W=16, H=16):
    """fog-box geometry with a NanoVDBMedium over a dense copy of the grid (index space = origin + index * voxel;
    the world bounding box covers index_min .. index_min + n, i.e. the voxel extents)."""
    P = load_package()
    bmin = tuple(origin[k] + index_min[k] * voxel[k] for k in range(3))
    bmax = tuple(origin[k] + (index_min[k] + n[k]) * voxel[k] for k in range(3))
    s = grid_scene(density, n, sigma_a, sigma_s, g=g, bmin=bmin, bmax=bmax, W=W, H=H)
    m = s.medium
    m.type = P.MEDIUM_NANOVDB
    m.index_min[:] = index_min
    m.voxel_size[:] = voxel
    m.grid_origin[:] = origin
    m.density_offset = density_offset
    m.majorant_scale = majorant_scale
    return s


def heightfield_triangles(n, x0=-1.0, x1=1.0, z0=-1.0, z1=1.0, y=-0.7, amp=0.25, seed=2):
    """2 n^2 triangles of a bumpy terrain over [x0,x1] x [z0,z1] around height y (smooth low-frequency bumps + fine ripples)."""
    rng = np.random.default_rng(seed)
    xs = np.linspace(x0, x1, n + 1, dtype=np.float32)
    zs = np.linspace(z0, z1, n + 1, dtype=np.float32)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    ph = rng.uniform(0, 6.28, 6)
    Y = (y + amp * (0.5 * np.sin(3.1 * X + ph[0]) * np.cos(2.3 * Z + ph[1]) + 0.3 * np.sin(7.7 * X + 5.1 * Z + ph[2]) +
                    0.08 * np.sin(31 * X + ph[3]) * np.sin(29 * Z + ph[4]))).astype(np.float32)
    P = np.stack([X, Y, Z], -1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    t1 = np.stack([a, c, b], -2)   # faces +y
    t2 = np.stack([a, d, c], -2)
    tris = np.concatenate([t1.reshape(-1, 3, 3), t2.reshape(-1, 3, 3)], 0).astype(np.float32)
    kd = np.tile(np.array([[0.6, 0.45, 0.3]], dtype=np.float32), (tris.shape[0], 1))
    kd[::3] = (0.3, 0.5, 0.6)
    return np.ascontiguousarray(tris), np.ascontiguousarray(kd)


def box_wall_triangles():
    """The five non-emissive walls of the App.-F box ... plus the wall behind the camera, as 12 triangles (Kd .73)."""
    def quad(p, e1, e2):
        p, e1, e2 = (np.array(v, dtype=np.float32) for v in (p, e1, e2))
        a, b, c, d = p, p + e1, p + e1 + e2, p + e2
        return [np.stack([a, b, c]), np.stack([a, c, d])]
    tris = []
    tris += quad((-1, -1, -1), (0, 0, 2), (2, 0, 0))   # floor
    tris += quad((-1, 1, -1), (2, 0, 0), (0, 0, 2))    # ceiling
    tris += quad((-1, -1, 1), (0, 2, 0), (2, 0, 0))    # back
    tris += quad((-1, -1, -1), (2, 0, 0), (0, 2, 0))   # front
    tris += quad((-1, -1, -1), (0, 2, 0), (0, 0, 2))   # left
    tris += quad((1, -1, -1), (0, 0, 2), (0, 2, 0))    # right
    tris = np.stack(tris).astype(np.float32)
    return tris, np.full((tris.shape[0], 3), 0.73, dtype=np.float32)


# ---------------------------------------------------------------------------------------------
# medium boundaries (round 4): camera outside the medium, interface surfaces, spheres
# ---------------------------------------------------------------------------------------------
def empty_scene(W, H, eye, look, up=(0, 1, 0), fov=40.0):
    """No geometry, no medium, no light: camera only."""
    P = load_package()
    s = P.VspgScene()
    lib = oracle_lib.load()
    rc = lib.oracle_camera_look_at(C.byref(s.camera), P.f3(*eye), P.f3(*look), P.f3(*up), C.c_float(fov), W, H)
    assert rc == 0
    return s


def add_quad(scene, p00, e1, e2, kd=(0.5, 0.5, 0.5), le=(0, 0, 0), material=0, iface=0, reverse=0, two_sided=0):
    k = scene.n_quads
    q = scene.quads[k]
    q.p00[:] = p00
    q.e1[:] = e1
    q.e2[:] = e2
    q.Kd[:] = kd
    q.Le[:] = le
    q.two_sided, q.reverse_orientation, q.material, q.medium_interface = two_sided, reverse, material, iface
    scene.n_quads = k + 1
    return scene


def interface_box(scene, bmin, bmax, iface=None):
    """Six interface-material rectangles around [bmin, bmax], normals pointing OUT, the scene's medium inside:
    AttributeBegin MediumInterface "m" "" Material "interface" Shape ... AttributeEnd."""
    P = load_package()
    iface = P.IFACE_INSIDE if iface is None else iface
    x0, y0, z0 = bmin
    x1, y1, z1 = bmax
    dx, dy, dz = x1 - x0, y1 - y0, z1 - z0
    faces = [((x0, y0, z0), (0, 0, dz), (0, dy, 0)),   # x = x0, n = -x  (e1 x e2 = z x y = -x)
             ((x1, y0, z0), (0, dy, 0), (0, 0, dz)),   # x = x1, n = +x
             ((x0, y0, z0), (dx, 0, 0), (0, 0, dz)),   # y = y0, n = -y  (x x z = -y)
             ((x0, y1, z0), (0, 0, dz), (dx, 0, 0)),   # y = y1, n = +y
             ((x0, y0, z0), (0, dy, 0), (dx, 0, 0)),   # z = z0, n = -z  (y x x = -z)
             ((x0, y0, z1), (dx, 0, 0), (0, dy, 0))]   # z = z1, n = +z
    for p00, e1, e2 in faces:
        add_quad(scene, p00, e1, e2, kd=(0, 0, 0), material=P.MATERIAL_INTERFACE, iface=iface)
    return scene


def add_sphere(scene, center, radius, material=0, iface=0, kd=(0.5, 0.5, 0.5), scale=(1, 1, 1), reverse=0):
    """Shape "sphere" under Translate(center) * Scale(scale)."""
    P = load_package()
    k = scene.n_spheres
    sp = scene.spheres[k]
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[1, 1], m[2, 2] = scale
    m[0, 3], m[1, 3], m[2, 3] = center
    inv = np.linalg.inv(m.astype(np.float64)).astype(np.float32)
    sp.render_from_object[:] = [float(x) for x in m.reshape(16)]
    sp.object_from_render[:] = [float(x) for x in inv.reshape(16)]
    sp.radius = radius
    sp.Kd[:] = kd
    sp.reverse_orientation, sp.material, sp.medium_interface = reverse, material, iface
    scene.n_spheres = k + 1
    return scene


def cloud_scene(W, H, density, n, sigma_t=8.0, albedo=0.99, g=0.877, nvdb=False, sphere=True, ground=True, sun=True, sky=True):
    """The shape of the reference's cloud scenes (BASELINE configs 3-5): camera in vacuum, the medium inside an
    interface-material bounding shape (MediumInterface "cloud" "" + Material "interface"), a diffuse ground below, a
    distant light and a uniform sky."""
    P = load_package()
    s = empty_scene(W, H, eye=(0.0, 0.6, -4.2), look=(0, 0.15, 0), fov=38.0)
    m = s.medium
    m.type = P.MEDIUM_NANOVDB if nvdb else P.MEDIUM_GRID
    m.sigma_a[:] = (sigma_t * (1 - albedo),) * 3
    m.sigma_s[:] = (sigma_t * albedo,) * 3
    m.g = g
    m.nx = m.ny = m.nz = n
    m.bounds_min[:] = (-0.8, -0.5, -0.8)
    m.bounds_max[:] = (0.8, 0.9, 0.8)
    m.density = density.ctypes.data_as(C.POINTER(C.c_float))
    s._density_keepalive = density
    if nvdb:
        for k in range(3):
            m.index_min[k] = 0
            m.voxel_size[k] = (m.bounds_max[k] - m.bounds_min[k]) / n
            m.grid_origin[k] = m.bounds_min[k]
        m.density_offset = 0.0
        m.majorant_scale = 1.0
    s.camera_outside_medium = 1
    if sphere:   # radius sqrt(0.8^2 + 0.7^2 + 0.8^2) around the bounds' centre: the bounds fit inside
        add_sphere(s, (0.0, 0.2, 0.0), 1.34, material=P.MATERIAL_INTERFACE, iface=P.IFACE_INSIDE)
    else:
        interface_box(s, (-0.8, -0.5, -0.8), (0.8, 0.9, 0.8))
    if ground:
        add_quad(s, (-6, -1.2, -6), (0, 0, 12), (12, 0, 0), kd=(0.4, 0.35, 0.3))   # n = +y
    if sun:
        P.add_infinite_light(s, P.LIGHT_DISTANT, (6.0, 5.5, 5.0), (0.4, 0.8, -0.3))
    if sky:
        P.add_infinite_light(s, P.LIGHT_UNIFORM_INFINITE, (0.25, 0.35, 0.5))
    return s
