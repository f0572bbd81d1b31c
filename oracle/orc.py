"""orc.py -- TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/liboracle.so (rt_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")

# numpy mirrors of the POD records in include/rt_mi355x.h
NODE = np.dtype([("tm", "<f4", 9), ("itm", "<f4", 9), ("pos", "<f4", 3), ("parent", "<i4"),
                 ("obj_type", "<i4"), ("mesh", "<i4"), ("material", "<i4")])
BVHNODE = np.dtype([("box", "<f4", 6), ("data", "<u4")])
BLINN = np.dtype([("diffuse", "<f4", 3), ("specular", "<f4", 3), ("reflection", "<f4", 3),
                  ("refraction", "<f4", 3), ("emission", "<f4", 3), ("absorption", "<f4", 3),
                  ("glossiness", "<f4"), ("ior", "<f4"), ("reflection_glossiness", "<f4"),
                  ("refraction_glossiness", "<f4")])
LIGHT = np.dtype([("type", "<i4"), ("intensity", "<f4", 3), ("position", "<f4", 3),
                  ("direction", "<f4", 3), ("size", "<f4")])
PHOTON = np.dtype([("position", "<f4", 3), ("power", "<f4"), ("color", "u1", 3),
                   ("plane_and_dirz", "u1"), ("dir_x", "<i2"), ("dir_y", "<i2")])
HIT = np.dtype([("z", "<f4"), ("p", "<f4", 3), ("N", "<f4", 3), ("node", "<i4"), ("front", "<i4"),
                ("uvw", "<f4", 3)])
TEXTURE = np.dtype([("type", "<i4"), ("width", "<i4"), ("height", "<i4"), ("texel_offset", "<u4"),
                    ("color1", "<f4", 3), ("color2", "<f4", 3)])
TEXMAP = np.dtype([("texture", "<i4"), ("tm", "<f4", 9), ("itm", "<f4", 9), ("pos", "<f4", 3)])
assert (NODE.itemsize, BVHNODE.itemsize, BLINN.itemsize, LIGHT.itemsize, PHOTON.itemsize,
        HIT.itemsize, TEXTURE.itemsize, TEXMAP.itemsize) == (100, 28, 88, 44, 24, 48, 40, 88)


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("dir", C.c_float * 3), ("up", C.c_float * 3),
                ("fov", C.c_float), ("focaldist", C.c_float), ("dof", C.c_float),
                ("width", C.c_int32), ("height", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("min_sample", C.c_int32), ("max_sample", C.c_int32), ("threshold", C.c_float),
                ("bounce", C.c_int32), ("hemisphere_sample", C.c_int32), ("knn_k", C.c_int32),
                ("knn_radius", C.c_float), ("shade_model", C.c_int32),
                ("shadow_samples", C.c_int32), ("seed", C.c_uint32), ("gamma", C.c_double),
                ("caustic_k", C.c_int32), ("caustic_radius", C.c_float), ("photon_count", C.c_int32), ("photon_bounce", C.c_int32)]


def default_params(**kw):
    """FIN defaults (FIN/main.cpp:19-32)."""
    p = Params(min_sample=4, max_sample=8, threshold=1e-3, bounce=4, hemisphere_sample=30,
               knn_k=400, knn_radius=1.0, shade_model=0, shadow_samples=4, seed=20171203, gamma=2.2)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


class _Mesh(C.Structure):
    _fields_ = [("v", C.c_void_p), ("nv", C.c_int32), ("f", C.c_void_p), ("nf", C.c_int32),
                ("vn", C.c_void_p), ("nvn", C.c_int32), ("fn", C.c_void_p),
                ("nodes", C.c_void_p), ("nnodes", C.c_int32), ("elements", C.c_void_p),
                ("vt", C.c_void_p), ("nvt", C.c_int32), ("ft", C.c_void_p)]


class _Scene(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("n_nodes", C.c_int32), ("meshes", C.c_void_p),
                ("n_meshes", C.c_int32), ("materials", C.c_void_p), ("n_materials", C.c_int32),
                ("lights", C.c_void_p), ("n_lights", C.c_int32), ("photons", C.c_void_p),
                ("n_photons", C.c_uint32), ("env", C.c_float * 3), ("bg", C.c_float * 3),
                ("textures", C.c_void_p), ("n_textures", C.c_int32), ("texels", C.c_void_p),
                ("material_maps", C.c_void_p), ("env_map", C.c_void_p), ("bg_map", C.c_void_p),
                ("caustic", C.c_void_p), ("n_caustic", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays_primary", "rays_shadow", "rays_reflect",
                                          "rays_refract", "box_tests", "tri_tests", "node_visits",
                                          "photon_queries", "photons_visited")]


def build():
    subprocess.run(["make", "-s", "-C", HERE, "liboracle.so"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_halton.restype = C.c_float
        _lib.orc_halton.argtypes = [C.c_int, C.c_int]
        _lib.orc_shadow.restype = C.c_float
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


class Mesh:
    def __init__(self, v, f, vn, fn, nodes, elements, vt=None, ft=None):
        self.v = _c(v, np.float32).reshape(-1, 3)
        self.f = _c(f, np.uint32).reshape(-1, 3)
        self.vn = _c(vn, np.float32).reshape(-1, 3)
        self.fn = _c(fn, np.uint32).reshape(-1, 3)
        self.nodes = _c(nodes, BVHNODE)
        self.elements = _c(elements, np.uint32)
        self.vt = _c(vt, np.float32).reshape(-1, 3) if vt is not None and len(vt) else None
        self.ft = _c(ft, np.uint32).reshape(-1, 3) if self.vt is not None else None

    def c(self):
        return _Mesh(_p(self.v), len(self.v), _p(self.f), len(self.f), _p(self.vn), len(self.vn),
                     _p(self.fn), _p(self.nodes), len(self.nodes), _p(self.elements),
                     _p(self.vt), len(self.vt) if self.vt is not None else 0, _p(self.ft))


class Scene:
    """Holds the numpy arrays alive and exposes an orc_scene."""

    def __init__(self, nodes, meshes=(), materials=None, lights=None, photons=None,
                 env=(0, 0, 0), bg=(0, 0, 0), textures=None, texels=None, material_maps=None,
                 env_map=None, bg_map=None, caustic=None):
        self.nodes = _c(nodes, NODE)
        self.meshes = list(meshes)
        self.materials = _c(materials if materials is not None else np.zeros(0, BLINN), BLINN)
        self.lights = _c(lights if lights is not None else np.zeros(0, LIGHT), LIGHT)
        self.photons = _c(photons, PHOTON) if photons is not None else np.zeros(0, PHOTON)
        self._cm = (_Mesh * max(1, len(self.meshes)))(*[m.c() for m in self.meshes])
        self.c = _Scene(_p(self.nodes), len(self.nodes), C.cast(self._cm, C.c_void_p),
                        len(self.meshes), _p(self.materials), len(self.materials),
                        _p(self.lights), len(self.lights), _p(self.photons),
                        max(0, len(self.photons) - 1), (C.c_float * 3)(*env), (C.c_float * 3)(*bg))
        self.textures = _c(textures, TEXTURE) if textures is not None else np.zeros(0, TEXTURE)
        self.texels = _c(texels, np.uint8) if texels is not None else np.zeros(0, np.uint8)
        self.material_maps = _c(material_maps, TEXMAP) if material_maps is not None else None
        self.env_map = _c(env_map, TEXMAP).reshape(1) if env_map is not None else None
        self.bg_map = _c(bg_map, TEXMAP).reshape(1) if bg_map is not None else None
        self.c.textures, self.c.n_textures, self.c.texels = _p(self.textures), len(self.textures), _p(self.texels)
        self.c.material_maps = _p(self.material_maps) if self.material_maps is not None else None
        self.c.env_map = _p(self.env_map) if self.env_map is not None else None
        self.c.bg_map = _p(self.bg_map) if self.bg_map is not None else None
        self.caustic = _c(caustic, PHOTON) if caustic is not None else np.zeros(0, PHOTON)
        self.c.caustic, self.c.n_caustic = _p(self.caustic), max(0, len(self.caustic) - 1)


def halton(i, base):
    return lib().orc_halton(int(i), int(base))


def color24(rgb):
    rgb = _c(rgb, np.float32).reshape(-1, 3)
    out = np.zeros((len(rgb), 3), np.uint8)
    for i in range(len(rgb)):
        lib().orc_color24(_p(rgb[i]), _p(out[i]))
    return out


def _prim(fn, model, rays, z0):
    rays = _c(rays, np.float32).reshape(-1, 6)
    n = len(rays)
    hits = np.zeros(n, HIT)
    hit = np.zeros(n, np.int32)
    hits["z"] = z0
    hits["uvw"] = 0.5           # HitInfo::Init(), scene.h:163
    hits["front"] = 1
    hits["node"] = -1
    for i in range(n):
        hit[i] = fn(int(model), _p(rays[i]), C.c_void_p(hits[i:i + 1].ctypes.data))
    return hit, hits


def sphere_intersect(model, rays, z0):
    return _prim(lib().orc_sphere_intersect, model, rays, z0)


def plane_intersect(model, rays, z0):
    return _prim(lib().orc_plane_intersect, model, rays, z0)


def box_intersect(box, ray, tmax):
    box, ray = _c(box, np.float32), _c(ray, np.float32)
    return lib().orc_box_intersect(_p(box), _p(ray), C.c_float(tmax))


def mesh_intersect(model, mesh, rays, z0):
    cm = mesh.c()
    return _prim(lambda m, r, h: lib().orc_mesh_intersect(m, C.byref(cm), r, h), model, rays, z0)


def to_node_coords(node, rays):
    node = _c(node, NODE).reshape(1)
    rays = _c(rays, np.float32).reshape(-1, 6)
    out = np.zeros_like(rays)
    for i in range(len(rays)):
        lib().orc_to_node_coords(_p(node), _p(rays[i]), _p(out[i]))
    return out


def from_node_coords(node, pN):
    node = _c(node, NODE).reshape(1)
    pN = _c(pN, np.float32).reshape(-1, 6)
    out = np.zeros_like(pN)
    h = np.zeros(1, HIT)
    for i in range(len(pN)):
        h["p"][0] = pN[i, :3]
        h["N"][0] = pN[i, 3:]
        lib().orc_from_node_coords(_p(node), _p(h))
        out[i, :3], out[i, 3:] = h["p"][0], h["N"][0]
    return out


def trace(scene, model, rays):
    rays = _c(rays, np.float32).reshape(-1, 6)
    hits = np.zeros(len(rays), HIT)
    hit = np.zeros(len(rays), np.int32)
    for i in range(len(rays)):
        hit[i] = lib().orc_trace(C.byref(scene.c), int(model), _p(rays[i]),
                                 C.c_void_p(hits[i:i + 1].ctypes.data))
    return hit, hits


def shade_rays(scene, params, rays):
    """Trace + Shade(ray, hit, lights, BOUNCE) per ray -> (hit flags, linear rgb, z)."""
    rays = _c(rays, np.float32).reshape(-1, 6)
    n = len(rays)
    rgb = np.zeros((n, 3), np.float32)
    hit = np.zeros(n, np.uint8)
    z = np.full(n, 1.0e30, np.float32)
    h = np.zeros(1, HIT)
    for i in range(n):
        lib().orc_set_rng(C.c_uint32(params.seed), C.c_uint32(i), C.c_uint32(1))      # sample id = ray index
        if lib().orc_trace(C.byref(scene.c), params.shade_model, _p(rays[i]), _p(h)):
            hit[i] = 1
            z[i] = h["z"][0]
            lib().orc_shade(C.byref(scene.c), C.byref(params), _p(rays[i]), _p(h),
                            params.bounce, _p(rgb[i]))
    return hit, rgb, z


def shade_cases(scene, params, rays, bounces):
    """TraceNode + Shade(ray, hit, lights, bounce_i, 0) per case -> (hit flags, hit records, linear rgb): the calls
    oracle/ref_main_harness.cpp `shade` makes on the reference (params.bounce stays the snapshot's BOUNCE)."""
    rays = _c(rays, np.float32).reshape(-1, 6)
    n = len(rays)
    rgb = np.zeros((n, 3), np.float32)
    hit = np.zeros(n, np.int32)
    hits = np.zeros(n, HIT)
    for i in range(n):
        lib().orc_set_rng(C.c_uint32(params.seed), C.c_uint32(i), C.c_uint32(1))
        h = hits[i:i + 1]
        hit[i] = lib().orc_trace(C.byref(scene.c), params.shade_model, _p(rays[i]), C.c_void_p(h.ctypes.data))
        if hit[i]:
            lib().orc_shade(C.byref(scene.c), C.byref(params), _p(rays[i]), C.c_void_p(h.ctypes.data), int(bounces[i]), _p(rgb[i]))
    return hit, hits, rgb


def glibc_rand(seed, n):
    """What glibc's srand(seed) followed by n calls of rand() returns (the TYPE_3 additive feedback generator of random_r:
    r[i] = r[i-3] + r[i-31] over a state seeded by the Lehmer generator 16807, first 310 outputs discarded, top 31 bits).
    The reference draws from it; fixtures of rand()-dependent cases keep the seed and the number of values consumed, and
    oracle/gen_golden.py checks this restatement against the captured stream of the reference run."""
    r = np.zeros(int(n) + 344, np.int64)
    seed = int(seed) & 0xFFFFFFFF
    r[0] = seed if seed else 1
    for i in range(1, 31):
        prev = int(r[i - 1])
        if prev >= 2 ** 31:
            prev -= 2 ** 32
        hi, lo = divmod(prev, 127773)
        w = 16807 * lo - 2836 * hi
        r[i] = w + 2147483647 if w < 0 else w
    for i in range(31, 34):
        r[i] = r[i - 31]
    rl = r.tolist()
    for i in range(34, len(rl)):
        rl[i] = (rl[i - 31] + rl[i - 3]) & 0xFFFFFFFF
    return (np.array(rl[344:], np.int64) >> 1).astype(np.int32)


def shade_cases_scripted(scene, params, rays, bounces, seeds, consumed):
    """shade_cases with the reference's rand() stream: case i runs with the first consumed[i] values of srand(seeds[i]) as its
    script; returns (hit flags, hit records, rgb, values the oracle consumed per case)"""
    rays = _c(rays, np.float32).reshape(-1, 6)
    n = len(rays)
    rgb = np.zeros((n, 3), np.float32)
    hit = np.zeros(n, np.int32)
    hits = np.zeros(n, HIT)
    used = np.zeros(n, np.int32)
    for i in range(n):
        h = hits[i:i + 1]
        hit[i] = lib().orc_trace(C.byref(scene.c), params.shade_model, _p(rays[i]), C.c_void_p(h.ctypes.data))
        if not hit[i]:
            continue
        raw = glibc_rand(seeds[i], int(consumed[i]) + 4)
        lib().orc_script_begin(_p(raw), len(raw), None, 0, None, 0)
        lib().orc_shade(C.byref(scene.c), C.byref(params), _p(rays[i]), C.c_void_p(h.ctypes.data), int(bounces[i]), _p(rgb[i]))
        u = C.c_int()
        lib().orc_script_end(C.byref(u))
        used[i] = u.value
    return hit, hits, rgb, used


def shadow(scene, model, rays, t_max):
    """GenLight::Shadow(ray, t_max) per ray"""
    rays = _c(rays, np.float32).reshape(-1, 6)
    lib().orc_shadow.restype = C.c_float
    return np.array([lib().orc_shadow(C.byref(scene.c), int(model), _p(rays[i]), C.c_float(float(t_max[i]))) for i in range(len(rays))], np.float32)


def texture_sample(texture, texels, uvw):
    texture = _c(texture, TEXTURE).reshape(1)
    texels = _c(texels, np.uint8)
    uvw = _c(uvw, np.float32).reshape(-1, 3)
    out = np.zeros_like(uvw)
    for i in range(len(uvw)):
        lib().orc_texture_sample(_p(texture), _p(texels), _p(uvw[i]), _p(out[i]))
    return out


def texmap_transform(texmap, uvw):
    texmap = _c(texmap, TEXMAP).reshape(1)
    uvw = _c(uvw, np.float32).reshape(-1, 3)
    out = np.zeros_like(uvw)
    for i in range(len(uvw)):
        lib().orc_texmap_transform(_p(texmap), _p(uvw[i]), _p(out[i]))
    return out


def environment_coord(dirs):
    dirs = _c(dirs, np.float32).reshape(-1, 3)
    out = np.zeros_like(dirs)
    for i in range(len(dirs)):
        lib().orc_environment_coord(_p(dirs[i]), _p(out[i]))
    return out


def photon_pack(pos, dirn, power):
    pos, dirn, power = (_c(a, np.float32).reshape(-1, 3) for a in (pos, dirn, power))
    out = np.zeros(len(pos), PHOTON)
    for i in range(len(pos)):
        lib().orc_photon_pack(_p(pos[i]), _p(dirn[i]), _p(power[i]), C.c_void_p(out[i:i + 1].ctypes.data))
    return out


def photon_decode(photons):
    photons = _c(photons, PHOTON)
    d = np.zeros((len(photons), 3), np.float32)
    pw = np.zeros((len(photons), 3), np.float32)
    for i in range(len(photons)):
        ptr = C.c_void_p(photons[i:i + 1].ctypes.data)
        lib().orc_photon_direction(ptr, _p(d[i]))
        lib().orc_photon_power(ptr, _p(pw[i]))
    return d, pw


def photon_balance(photons_1based):
    """photons_1based: array of n+1 records (index 0 unused).  Returns the balanced array."""
    a = _c(photons_1based, PHOTON).copy()
    out = np.zeros_like(a)
    lib().orc_photon_balance(_p(a), C.c_uint32(len(a) - 1), _p(out))
    return out


def estimate_irradiance(balanced, k, radius, pos, normal):
    balanced = _c(balanced, PHOTON)
    pos, normal = (_c(a, np.float32).reshape(-1, 3) for a in (pos, normal))
    irr = np.zeros_like(pos)
    d = np.zeros_like(pos)
    for i in range(len(pos)):
        lib().orc_estimate_irradiance(_p(balanced), C.c_uint32(len(balanced) - 1), int(k),
                                      C.c_float(radius), _p(pos[i]), _p(normal[i]), _p(irr[i]), _p(d[i]))
    return irr, d


def photon_pass(scene, max_photons, max_bounce=8, seed=20171203):
    out = np.zeros(int(max_photons) + 9, PHOTON)
    att = C.c_uint64()
    lib().orc_photon_pass.restype = C.c_uint32
    n = lib().orc_photon_pass(C.byref(scene.c), C.c_uint32(int(seed)), C.c_uint32(int(max_photons)), int(max_bounce),
                              _p(out), C.byref(att))
    return out[: n + 1].copy(), att.value


def bvh_build(v, f, max_per_leaf=4):
    v = _c(v, np.float32).reshape(-1, 3)
    f = _c(f, np.uint32).reshape(-1, 3)
    nodes = np.zeros(2 * len(f) + 2, BVHNODE)
    el = np.zeros(len(f), np.uint32)
    n = lib().orc_bvh_build(_p(v), _p(f), len(f), int(max_per_leaf), _p(nodes), _p(el))
    return nodes[:n].copy(), el


def primary_ray(cam, x, y, j):
    r = np.zeros(6, np.float32)
    lib().orc_primary_ray(C.byref(cam), int(x), int(y), int(j), _p(r))
    return r


def render(scene, cam, params, x0=0, y0=0, x1=None, y1=None):
    w, h = cam.width, cam.height
    x1 = w if x1 is None else x1
    y1 = h if y1 is None else y1
    rgb = np.zeros((h, w, 3), np.uint8)
    z = np.zeros((h, w), np.float32)
    cnt = np.zeros((h, w), np.uint8)
    lib().orc_render(C.byref(scene.c), C.byref(cam), C.byref(params), int(x0), int(y0), int(x1),
                     int(y1), _p(rgb), _p(z), _p(cnt))
    return rgb, z, cnt


def counters_reset():
    lib().orc_counters_reset()


def counters():
    c = Counters()
    lib().orc_counters_get(C.byref(c))
    return {n: getattr(c, n) for n, _ in Counters._fields_}


# ---- scripted randomness / Shadow (test hooks of rt_oracle.c) and the small image helpers -------------
def illuminate_scripted(model, intensity, position, size, p, raw_rand, script, shadow_samples=4):
    """PointLight::Illuminate with the raw rand() values and the Shadow return values of a reference run.
    Returns (result[3], shadow-call log (n,7) = ray.p, ray.dir, t_max, rand values used)."""
    light = np.zeros(1, LIGHT)
    light["type"], light["intensity"], light["position"], light["size"] = 2, intensity, position, size
    scene = Scene(np.zeros(0, NODE), lights=light)
    params = default_params(shade_model=int(model), shadow_samples=int(shadow_samples))
    raw = _c(raw_rand, np.int32)
    sc = _c(script, np.float32)
    log = np.zeros((32, 7), np.float32)
    out = np.zeros(3, np.float32)
    pp, nn = _c(p, np.float32), np.array([0, 0, 1], np.float32)
    lib().orc_script_begin(_p(raw), len(raw), _p(sc), len(sc), _p(log), len(log))
    lib().orc_illuminate(C.byref(scene.c), C.byref(params), _p(scene.lights), _p(pp), _p(nn), _p(out))
    used = C.c_int()
    n = lib().orc_script_end(C.byref(used))
    return out, log[:n], used.value


def random_photon_bounce_scripted(material, ray, hit_p, hit_N, hit_z, front, c, raw_rand):
    m = _c(material, BLINN).reshape(1)
    h = np.zeros(1, HIT)
    h["p"], h["N"], h["z"], h["front"] = hit_p, hit_N, hit_z, front
    r = _c(ray, np.float32).copy()
    cc = _c(c, np.float32).copy()
    raw = _c(raw_rand, np.int32)
    lib().orc_script_begin(_p(raw), len(raw), None, 0, None, 0)
    ret = lib().orc_random_photon_bounce(_p(m), _p(h), _p(r), _p(cc))
    used = C.c_int()
    lib().orc_script_end(C.byref(used))
    return ret, r, cc, used.value


def attenuation(absorption, l):
    out = np.zeros(3, np.float32)
    lib().orc_attenuation(_p(_c(absorption, np.float32)), C.c_float(l), _p(out))
    return out


def coordinate_system(N):
    nt, nb = np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().orc_coordinate_system(_p(_c(N, np.float32)), _p(nt), _p(nb))
    return nt, nb


def zbuffer_image(z):
    z = _c(z, np.float32)
    out = np.zeros(z.shape, np.uint8)
    lib().orc_zbuffer_image(_p(z), z.shape[1], z.shape[0], _p(out))
    return out


def sample_count_image(cnt):
    cnt = _c(cnt, np.uint8)
    out = np.zeros(cnt.shape, np.uint8)
    smax = lib().orc_sample_count_image(_p(cnt), cnt.shape[1], cnt.shape[0], _p(out))
    return out, smax


# ---- the product's exported arrays as oracle inputs (same bytes on both sides) -----------------------
def scene_from_export(export, photons=None, env=None, bg=None, caustic=None):
    """orc.Scene over the very arrays raytracing_folder_amd.capi.Scene.export() returns (environment and
    background colours come from the export unless given)."""
    env = tuple(export.get("env", (0, 0, 0))) if env is None else env
    bg = tuple(export.get("bg", (0, 0, 0))) if bg is None else bg
    meshes = [Mesh(m["v"], m["f"], m["vn"], m["fn"], m["nodes"], m["elements"], m.get("vt"), m.get("ft")) for m in export["meshes"]]
    return Scene(export["nodes"], meshes, export["materials"], export["lights"], photons, env, bg,
                 textures=export.get("textures"), texels=export.get("texels"),
                 material_maps=export.get("material_maps"), env_map=export.get("env_map"),
                 bg_map=export.get("bg_map"), caustic=caustic)


def camera_from(cam):
    oc = Camera()
    for f, _ in Camera._fields_:
        setattr(oc, f, getattr(cam, f))
    return oc


def params_from(p):
    op = Params()
    for f, _ in Params._fields_:
        setattr(op, f, getattr(p, f))
    return op


def set_trace_discarded(on):
    """FIN's discarded hemisphere loop at primary hits: traced (and thrown away) like the reference, or skipped"""
    lib().orc_set_trace_discarded(1 if on else 0)


def discarded_rays():
    lib().orc_discarded_rays.restype = C.c_uint64
    return lib().orc_discarded_rays()


def caustic_pass(scene, max_diffuse_hits, max_bounce=5, seed=20171203):
    out = np.zeros(int(max_diffuse_hits) + 9, PHOTON)
    att = C.c_uint64()
    lib().orc_caustic_pass.restype = C.c_uint32
    n = lib().orc_caustic_pass(C.byref(scene.c), C.c_uint32(int(seed)), C.c_uint32(int(max_diffuse_hits)), int(max_bounce),
                               _p(out), C.byref(att))
    return out[: n + 1].copy(), att.value


def light_illuminate_scripted(light, p, shadow_value, model=0):
    """AmbientLight / DirectLight / PointLight: (Illuminate result, Direction(p), shadow-call log) with Shadow scripted"""
    light = _c(light, LIGHT).reshape(1)
    scene = Scene(np.zeros(0, NODE), lights=light)
    params = default_params(shade_model=int(model))
    sc = np.array([shadow_value], np.float32)
    log = np.zeros((8, 7), np.float32)
    out, d = np.zeros(3, np.float32), np.zeros(3, np.float32)
    pp, nn = _c(p, np.float32), np.array([0, 0, 1], np.float32)
    lib().orc_script_begin(None, 0, _p(sc), 1, _p(log), len(log))
    lib().orc_illuminate(C.byref(scene.c), C.byref(params), _p(scene.lights), _p(pp), _p(nn), _p(out))
    n = lib().orc_script_end(None)
    lib().orc_light_direction(_p(scene.lights), _p(pp), _p(d))
    return out, d, log[:n]
