"""ctypes binding of librt_mi355x.so (the C ABI declared in include/rt_mi355x.h).

The render entry points run ONLY on the HIP kernels; when the library or a gfx950 device is
missing they raise -- there is no CPU fallback in this package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "librt_mi355x.so")
CSRC = os.path.join(HERE, "csrc")

# numpy mirrors of the POD records of include/rt_mi355x.h
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
TEXTURE = np.dtype([("type", "<i4"), ("width", "<i4"), ("height", "<i4"), ("texel_offset", "<u4"),
                    ("color1", "<f4", 3), ("color2", "<f4", 3)])
TEXMAP = np.dtype([("texture", "<i4"), ("tm", "<f4", 9), ("itm", "<f4", 9), ("pos", "<f4", 3)])
assert (NODE.itemsize, BVHNODE.itemsize, BLINN.itemsize, LIGHT.itemsize, PHOTON.itemsize,
        TEXTURE.itemsize, TEXMAP.itemsize) == (100, 28, 88, 44, 24, 40, 88)
TEX_FILE, TEX_CHECKER, MAP_NONE, MAP_EMPTY = 1, 2, -1, -2

OBJ_NONE, OBJ_SPHERE, OBJ_PLANE, OBJ_MESH = 0, 1, 2, 3
LIGHT_AMBIENT, LIGHT_DIRECT, LIGHT_POINT = 0, 1, 2
SHADE_FIN, SHADE_P13, SHADE_P12, SHADE_P6, SHADE_P3 = 0, 1, 2, 3, 4


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


class SetupMs(C.Structure):
    """rt_setup_ms: wall time of the stages of rt_scene_generate_photons, milliseconds"""
    _fields_ = [(n, C.c_double) for n in ("photon_pass", "balance", "structure_build", "upload", "total")]

    def as_dict(self):
        return {n: round(getattr(self, n), 3) for n, _ in self._fields_}


class TileRange(C.Structure):
    _fields_ = [("tile_w", C.c_int32), ("tile_h", C.c_int32), ("first", C.c_int32), ("stride", C.c_int32)]


class Stats(C.Structure):
    _fields_ = ([(n, C.c_uint64) for n in ("rays_primary", "rays_shadow", "rays_reflect", "rays_refract",
                                           "instance_visits", "bvh_nodes_visited", "tris_tested",
                                           "photon_queries", "photons_visited", "pixels", "samples")] +
                [(n, C.c_double) for n in ("ms_trace", "ms_gather", "ms_resolve", "ms_total")] +
                [(n, C.c_uint64) for n in ("launches_trace", "launches_gather", "launches_resolve",
                                           "gather_rounds", "gather_slow", "gather_leaf_reads")] +
                [(n, C.c_double) for n in ("ms_primary", "ms_bounce")] +
                [(n, C.c_uint64) for n in ("launches_primary", "launches_bounce", "streams", "peak_rays", "peak_queries", "attempts")])

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# every symbol include/rt_mi355x.h declares
SYMBOLS = [
    "rt_abi_version", "rt_last_error", "rt_device_count", "rt_params_default",
    "rt_scene_create", "rt_scene_destroy", "rt_scene_set_nodes", "rt_scene_set_mesh",
    "rt_scene_set_mesh_texcoords", "rt_scene_get_mesh_texcoords",
    "rt_scene_set_materials", "rt_scene_set_lights", "rt_scene_set_environment", "rt_scene_get_environment",
    "rt_scene_set_photons", "rt_scene_set_caustic_photons", "rt_scene_set_textures", "rt_scene_set_material_maps", "rt_scene_set_environment_maps",
    "rt_scene_get_textures", "rt_scene_get_maps", "rt_image_read_rgb", "rt_image_write_png", "rt_image_zbuffer", "rt_image_sample_count", "rt_scene_load_xml", "rt_scene_get_camera", "rt_scene_counts",
    "rt_scene_get_nodes", "rt_scene_get_materials", "rt_scene_get_lights", "rt_scene_mesh_counts",
    "rt_scene_get_mesh", "rt_bvh_build", "rt_photon_balance", "rt_photons_write_dat", "rt_photons_read_dat", "rt_photon_pass", "rt_caustic_pass", "rt_render_begin",
    "rt_render_tiles_device", "rt_render_tiles_packed_device", "rt_tiles_packed_size", "rt_tiles_unpack_device", "rt_render_check", "rt_render_counters", "rt_render_progress", "rt_render_stop", "rt_render_wait",
    "rt_job_stats", "rt_job_setup_ms", "rt_job_destroy", "rt_trace_rays", "rt_estimate_irradiance", "rt_shade_rays",
    "rt_scene_generate_photons", "rt_scene_set_photon_dump", "rt_scene_get_photons", "rt_photon_unreachable", "rt_photon_unreachable_device",
]


class RtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"rt_mi355x status {status}: {message}")
        self.status = status


def build(verbose=False):
    """Compile librt_mi355x.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", CSRC] + ([] if verbose else ["-s"]), check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("RT_MI355X_LIB", LIB_PATH)      # another build of the same ABI (tuning runs)
        if not os.path.exists(path):
            raise RtError(-3, f"{path} is missing: run __graft_entry__.build() (no CPU fallback exists)")
        _lib = C.CDLL(path)
        _lib.rt_last_error.restype = C.c_char_p
        for name in SYMBOLS:
            getattr(_lib, name)          # AttributeError here = header/library mismatch
        _lib.rt_render_progress.restype = C.c_int
    return _lib


def _check(st):
    if st != 0:
        raise RtError(st, lib().rt_last_error().decode(errors="replace"))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _stream_handle(stream):
    if stream is None:
        return None
    if int(stream) == 0:
        raise ValueError("stream handle 0 (torch's legacy default stream) cannot be named here: NULL means the library's own "
                         "stream; use an explicit torch.cuda.Stream")
    return C.c_void_p(int(stream))


def default_params(**kw):
    p = Params()
    lib().rt_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def device_count():
    return lib().rt_device_count()


def bvh_build(v, f, max_per_leaf=4):
    v = _c(v, np.float32).reshape(-1, 3)
    f = _c(f, np.uint32).reshape(-1, 3)
    nodes = np.zeros(2 * len(f) + 2, BVHNODE)
    el = np.zeros(len(f), np.uint32)
    n = C.c_int32()
    _check(lib().rt_bvh_build(_p(v), len(v), _p(f), len(f), int(max_per_leaf), _p(nodes), C.byref(n), _p(el)))
    return nodes[:n.value].copy(), el


def image_read_rgb(path):
    w, h = C.c_int32(), C.c_int32()
    _check(lib().rt_image_read_rgb(os.fsencode(path), C.byref(w), C.byref(h), None, C.c_uint64(0)))
    rgb = np.zeros((h.value, w.value, 3), np.uint8)
    _check(lib().rt_image_read_rgb(os.fsencode(path), C.byref(w), C.byref(h), _p(rgb), C.c_uint64(rgb.size)))
    return rgb


def image_write_png(path, data):
    data = np.ascontiguousarray(data, np.uint8)
    comps = 1 if data.ndim == 2 else data.shape[2]
    _check(lib().rt_image_write_png(os.fsencode(path), _p(data), data.shape[1], data.shape[0], comps))


def zbuffer_image(z):
    """RenderImage::ComputeZBufferImage (scene.h:591-613) of a float z buffer (h, w)."""
    z = np.ascontiguousarray(z, np.float32)
    out = np.zeros(z.shape, np.uint8)
    _check(lib().rt_image_zbuffer(_p(z), z.shape[1], z.shape[0], _p(out)))
    return out


def sample_count_image(cnt):
    """RenderImage::ComputeSampleCountImage (scene.h:615-637): (image, smax)."""
    cnt = np.ascontiguousarray(cnt, np.uint8)
    out = np.zeros(cnt.shape, np.uint8)
    smax = C.c_int32()
    _check(lib().rt_image_sample_count(_p(cnt), cnt.shape[1], cnt.shape[0], _p(out), C.byref(smax)))
    return out, smax.value


def tiles_packed_size(width, height, tiles):
    nbytes, n = C.c_uint64(), C.c_int32()
    _check(lib().rt_tiles_packed_size(int(width), int(height), C.byref(tiles), C.byref(nbytes), C.byref(n)))
    return nbytes.value, n.value


def tiles_unpack_device(device, stream, gathered_ptr, world, tiles_per_rank, width, height, tile_w, tile_h, rgb_ptr, z_ptr, cnt_ptr):
    """gathered packed tiles of `world` ranks -> RenderImage planes, one HIP kernel on `stream` (an explicit stream's
    handle; None = the null stream of the device)"""
    handle = _stream_handle(stream)
    _check(lib().rt_tiles_unpack_device(int(device), handle, C.c_void_p(gathered_ptr), int(world), int(tiles_per_rank), int(width),
                                        int(height), int(tile_w), int(tile_h), C.c_void_p(rgb_ptr), C.c_void_p(z_ptr), C.c_void_p(cnt_ptr)))


def identity_map(texture=MAP_NONE):
    m = np.zeros(1, TEXMAP)
    m["texture"] = texture
    m["tm"][0, [0, 4, 8]] = 1
    m["itm"][0, [0, 4, 8]] = 1
    return m


def photons_write_dat(path, photons_1based):
    """the reference's photonmap.dat: records [1..n] of a 1-based photon array, 24 bytes each"""
    a = _c(photons_1based, PHOTON)
    _check(lib().rt_photons_write_dat(str(path).encode(), _p(a), C.c_uint32(max(0, len(a) - 1))))


def photons_read_dat(path):
    """1-based photon array (entry 0 unused) from a photonmap.dat"""
    n = C.c_uint32()
    _check(lib().rt_photons_read_dat(str(path).encode(), None, 0, C.byref(n)))
    out = np.zeros(n.value + 1, PHOTON)
    _check(lib().rt_photons_read_dat(str(path).encode(), _p(out), len(out), C.byref(n)))
    return out


def photon_balance(photons_1based):
    a = _c(photons_1based, PHOTON).copy()
    out = np.zeros_like(a)
    _check(lib().rt_photon_balance(_p(a), C.c_uint32(len(a) - 1), _p(out)))
    return out


def photon_unreachable(photons_1based):
    """1-based indices into an UNBALANCED photon array of the photons LocatePhotons cannot reach after balancing"""
    a = _c(photons_1based, PHOTON)
    idx = np.zeros(8, np.uint32)
    n = C.c_uint32()
    _check(lib().rt_photon_unreachable(_p(a), C.c_uint32(len(a) - 1), _p(idx), 8, C.byref(n)))
    return idx[: n.value].copy()


def photon_unreachable_device(photons_1based, device=0):
    """the same on the GPU (rt_photon_unreachable_device): (sorted 1-based raw indices, exact); exact == False: a median key is not
    unique, only the host function answers"""
    ph = np.ascontiguousarray(photons_1based, PHOTON)
    n = len(ph) - 1
    idx = np.zeros(16, np.uint32)
    cnt, exact = C.c_uint32(), C.c_int32()
    _check(lib().rt_photon_unreachable_device(int(device), _p(ph), C.c_uint32(n), _p(idx), 16, C.byref(cnt), C.byref(exact)))
    return idx[:cnt.value].copy(), bool(exact.value)


class Scene:
    """Owns an rt_scene handle."""

    def __init__(self):
        self._h = C.c_void_p()
        _check(lib().rt_scene_create(C.byref(self._h)))
        self._keep = []

    def close(self):
        if self._h:
            lib().rt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- setters -------------------------------------------------------------------------------
    def set_nodes(self, nodes):
        nodes = _c(nodes, NODE)
        _check(lib().rt_scene_set_nodes(self._h, _p(nodes), len(nodes)))

    def set_mesh(self, index, v, f, vn, fn, nodes, elements, vt=None, ft=None):
        v, vn = _c(v, np.float32).reshape(-1, 3), _c(vn, np.float32).reshape(-1, 3)
        f, fn = _c(f, np.uint32).reshape(-1, 3), _c(fn, np.uint32).reshape(-1, 3)
        nodes, elements = _c(nodes, BVHNODE), _c(elements, np.uint32)
        _check(lib().rt_scene_set_mesh(self._h, int(index), _p(v), len(v), _p(f), len(f), _p(vn), len(vn),
                                       _p(fn), _p(nodes), len(nodes), _p(elements)))
        if vt is not None and len(vt):
            self.set_mesh_texcoords(index, vt, ft)

    def set_mesh_texcoords(self, index, vt, ft):
        """cyTriMesh VT/FT of a mesh already set (read by the PROJ13-family triangle)."""
        vt, ft = _c(vt, np.float32).reshape(-1, 3), _c(ft, np.uint32).reshape(-1, 3)
        _check(lib().rt_scene_set_mesh_texcoords(self._h, int(index), _p(vt), len(vt), _p(ft)))

    def set_materials(self, m):
        m = _c(m, BLINN)
        _check(lib().rt_scene_set_materials(self._h, _p(m), len(m)))

    def set_lights(self, l):
        l = _c(l, LIGHT)
        _check(lib().rt_scene_set_lights(self._h, _p(l), len(l)))

    def set_environment(self, env=(0, 0, 0), bg=(0, 0, 0)):
        e, b = (C.c_float * 3)(*env), (C.c_float * 3)(*bg)
        _check(lib().rt_scene_set_environment(self._h, e, b))

    def set_textures(self, textures, texels):
        textures, texels = _c(textures, TEXTURE), _c(texels, np.uint8)
        _check(lib().rt_scene_set_textures(self._h, _p(textures), len(textures), _p(texels), C.c_uint64(texels.size)))

    def set_material_maps(self, maps):
        maps = _c(maps, TEXMAP)
        assert len(maps) % 2 == 0
        _check(lib().rt_scene_set_material_maps(self._h, _p(maps), len(maps) // 2))

    def set_environment_maps(self, environment=None, background=None):
        e = _c(environment, TEXMAP).reshape(1) if environment is not None else None
        b = _c(background, TEXMAP).reshape(1) if background is not None else None
        _check(lib().rt_scene_set_environment_maps(self._h, _p(e) if e is not None else None, _p(b) if b is not None else None))

    def set_photons(self, balanced_1based):
        if balanced_1based is None or len(balanced_1based) < 2:
            _check(lib().rt_scene_set_photons(self._h, None, C.c_uint32(0)))
            return
        a = _c(balanced_1based, PHOTON)
        _check(lib().rt_scene_set_photons(self._h, _p(a), C.c_uint32(len(a) - 1)))

    def set_caustic_photons(self, balanced_1based):
        """the second map (P13's causticmap): same format as set_photons; used when params.caustic_k > 0"""
        if balanced_1based is None or len(balanced_1based) < 2:
            _check(lib().rt_scene_set_caustic_photons(self._h, None, C.c_uint32(0)))
            return
        a = _c(balanced_1based, PHOTON)
        _check(lib().rt_scene_set_caustic_photons(self._h, _p(a), C.c_uint32(len(a) - 1)))

    def generate_photons(self, max_photons=1000000, photon_bounce=8, seed=20171203, device=0, dat_path=None):
        """generatePhotonMap as a whole on the GPU (photon pass -> [dump] -> queryable structure); returns the stage times"""
        ms = SetupMs()
        _check(lib().rt_scene_generate_photons(self._h, int(device), C.c_uint32(int(max_photons)), int(photon_bounce), C.c_uint32(int(seed)),
                                               os.fsencode(dat_path) if dat_path else None, C.byref(ms)))
        return ms

    def set_photon_dump(self, dat_path):
        _check(lib().rt_scene_set_photon_dump(self._h, os.fsencode(dat_path) if dat_path else None))

    def get_photons(self):
        """the scene's photon map in the reference's balanced form (1-based; empty array when there is none)"""
        n = C.c_uint32()
        _check(lib().rt_scene_get_photons(self._h, None, 0, C.byref(n)))
        if n.value == 0:
            return np.zeros(0, PHOTON)
        out = np.zeros(n.value + 1, PHOTON)
        _check(lib().rt_scene_get_photons(self._h, _p(out), len(out), None))
        return out

    def load_xml(self, path):
        _check(lib().rt_scene_load_xml(self._h, os.fsencode(path)))

    # -- getters -------------------------------------------------------------------------------
    def camera(self):
        cam = Camera()
        _check(lib().rt_scene_get_camera(self._h, C.byref(cam)))
        return cam

    def counts(self):
        n = [C.c_int32() for _ in range(4)]
        nph = C.c_uint32()
        _check(lib().rt_scene_counts(self._h, *[C.byref(x) for x in n], C.byref(nph)))
        return dict(nodes=n[0].value, meshes=n[1].value, materials=n[2].value, lights=n[3].value, photons=nph.value)

    def export(self):
        """All host-side arrays (for handing the same bytes to another consumer)."""
        c = self.counts()
        nodes = np.zeros(c["nodes"], NODE)
        mats = np.zeros(c["materials"], BLINN)
        lights = np.zeros(c["lights"], LIGHT)
        _check(lib().rt_scene_get_nodes(self._h, _p(nodes), len(nodes)))
        if len(mats):
            _check(lib().rt_scene_get_materials(self._h, _p(mats), len(mats)))
        if len(lights):
            _check(lib().rt_scene_get_lights(self._h, _p(lights), len(lights)))
        meshes = []
        for m in range(c["meshes"]):
            k = [C.c_int32() for _ in range(4)]
            _check(lib().rt_scene_mesh_counts(self._h, m, *[C.byref(x) for x in k]))
            nv, nf, nvn, nn = (x.value for x in k)
            d = dict(v=np.zeros((nv, 3), np.float32), f=np.zeros((nf, 3), np.uint32),
                     vn=np.zeros((nvn, 3), np.float32), fn=np.zeros((nf, 3), np.uint32),
                     nodes=np.zeros(nn, BVHNODE), elements=np.zeros(nf, np.uint32))
            _check(lib().rt_scene_get_mesh(self._h, m, _p(d["v"]), _p(d["f"]), _p(d["vn"]), _p(d["fn"]),
                                           _p(d["nodes"]), _p(d["elements"])))
            nvt = C.c_int32()
            _check(lib().rt_scene_get_mesh_texcoords(self._h, m, C.byref(nvt), None, None))
            d["vt"], d["ft"] = np.zeros((nvt.value, 3), np.float32), np.zeros((nf if nvt.value else 0, 3), np.uint32)
            if nvt.value:
                _check(lib().rt_scene_get_mesh_texcoords(self._h, m, None, _p(d["vt"]), _p(d["ft"])))
            meshes.append(d)
        ntex, nbytes = C.c_int32(), C.c_uint64()
        _check(lib().rt_scene_get_textures(self._h, None, 0, None, C.c_uint64(0), C.byref(ntex), C.byref(nbytes)))
        textures, texels = np.zeros(ntex.value, TEXTURE), np.zeros(nbytes.value, np.uint8)
        _check(lib().rt_scene_get_textures(self._h, _p(textures), len(textures), _p(texels), C.c_uint64(texels.size), None, None))
        maps, env_map, bg_map = np.zeros(2 * len(mats), TEXMAP), np.zeros(1, TEXMAP), np.zeros(1, TEXMAP)
        _check(lib().rt_scene_get_maps(self._h, _p(maps), len(maps), _p(env_map), _p(bg_map)))
        has_maps = bool(len(maps)) and bool((maps["tm"] != 0).any() or (maps["texture"] != 0).any())
        env, bg = np.zeros(3, np.float32), np.zeros(3, np.float32)
        _check(lib().rt_scene_get_environment(self._h, _p(env), _p(bg)))
        return dict(nodes=nodes, materials=mats, lights=lights, meshes=meshes, textures=textures, texels=texels,
                    material_maps=maps if has_maps else None, env_map=env_map, bg_map=bg_map, env=env, bg=bg)

    # -- GPU work -------------------------------------------------------------------------------
    def trace_rays(self, rays, shade_model=SHADE_FIN, device=0):
        rays = _c(rays, np.float32).reshape(-1, 6)
        n = len(rays)
        out = dict(hit=np.zeros(n, np.uint8), z=np.zeros(n, np.float32), p=np.zeros((n, 3), np.float32),
                   N=np.zeros((n, 3), np.float32), node=np.zeros(n, np.int32), front=np.zeros(n, np.uint8))
        _check(lib().rt_trace_rays(self._h, int(shade_model), int(device), _p(rays), C.c_int64(n), _p(out["hit"]),
                                   _p(out["z"]), _p(out["p"]), _p(out["N"]), _p(out["node"]), _p(out["front"])))
        return out

    def estimate_irradiance(self, k, radius, pos, normal, device=0):
        pos, normal = _c(pos, np.float32).reshape(-1, 3), _c(normal, np.float32).reshape(-1, 3)
        irr, d = np.zeros_like(pos), np.zeros_like(pos)
        _check(lib().rt_estimate_irradiance(self._h, int(device), int(k), C.c_float(radius), _p(pos), _p(normal),
                                            C.c_int64(len(pos)), _p(irr), _p(d)))
        return irr, d

    def shade_rays(self, params, rays, device=0):
        rays = _c(rays, np.float32).reshape(-1, 6)
        n = len(rays)
        hit, rgb, z = np.zeros(n, np.uint8), np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
        _check(lib().rt_shade_rays(self._h, C.byref(params), int(device), _p(rays), C.c_int64(n), _p(hit), _p(rgb), _p(z)))
        return hit, rgb, z

    def photon_pass(self, max_photons, photon_bounce=8, seed=20171203, device=0):
        """generatePhotonMap on the GPU: returns (unbalanced 1-based photon array, attempts)."""
        out = np.zeros(int(max_photons) + 9, PHOTON)
        n, att = C.c_uint32(), C.c_uint64()
        _check(lib().rt_photon_pass(self._h, int(device), C.c_uint32(int(max_photons)), int(photon_bounce),
                                    C.c_uint32(int(seed)), _p(out), C.c_uint32(len(out)), C.byref(n), C.byref(att)))
        return out[: n.value + 1].copy(), att.value

    def caustic_pass(self, max_diffuse_hits, photon_bounce=5, seed=20171203, device=0):
        """P13's caustic loop on the GPU: returns (unbalanced 1-based photon array, attempts)."""
        out = np.zeros(int(max_diffuse_hits) + 9, PHOTON)
        n, att = C.c_uint32(), C.c_uint64()
        _check(lib().rt_caustic_pass(self._h, int(device), C.c_uint32(int(max_diffuse_hits)), int(photon_bounce),
                                     C.c_uint32(int(seed)), _p(out), C.c_uint32(len(out)), C.byref(n), C.byref(att)))
        return out[: n.value + 1].copy(), att.value

    def render(self, cam, params, tiles=None, device=0, photon_pass=False):
        """Blocking render through the asynchronous job API (rt_render_begin + rt_render_wait).  photon_pass=False (the
        tests' default) renders the scene's photon map as it is: params.photon_count is taken as 0 for this call;
        photon_pass=True leaves it alone, so that rt_render_begin first runs generatePhotonMap like BeginRender does."""
        if not photon_pass and params.photon_count != 0:
            q = Params()
            C.memmove(C.byref(q), C.byref(params), C.sizeof(Params))
            q.photon_count = 0
            params = q
        w, h = cam.width, cam.height
        rgb, z, cnt = np.zeros((h, w, 3), np.uint8), np.zeros((h, w), np.float32), np.zeros((h, w), np.uint8)
        tiles = tiles or TileRange(32, 8, 0, 1)
        job = C.c_void_p()
        _check(lib().rt_render_begin(self._h, C.byref(cam), C.byref(params), C.byref(tiles), int(device),
                                     _p(rgb), _p(z), _p(cnt), C.byref(job)))
        try:
            _check(lib().rt_render_wait(job))
            st = Stats()
            _check(lib().rt_job_stats(job, C.byref(st)))
            progress = lib().rt_render_progress(job)
        finally:
            lib().rt_job_destroy(job)
        return rgb, z, cnt, st, progress

    def render_tiles_packed_device(self, cam, params, tiles, device, packed_ptr, packed_bytes, stream=None, sync=True,
                                   want_stats=True):
        """This call's tiles as packed 8-byte pixel records (the all-gather contribution of a rank), see the header."""
        st = Stats()
        handle = _stream_handle(stream)
        _check(lib().rt_render_tiles_packed_device(self._h, C.byref(cam), C.byref(params), C.byref(tiles), int(device), handle,
                                                   C.c_void_p(packed_ptr), C.c_uint64(int(packed_bytes)), 1 if sync else 0,
                                                   C.byref(st) if want_stats else None))
        return st

    def render_counters(self, device=0, reset=False):
        """the device-side work counters accumulated since they were last cleared (rt_render_counters); waits for pending renders"""
        st = Stats()
        _check(lib().rt_render_counters(self._h, int(device), 1 if reset else 0, C.byref(st)))
        return st

    def render_check(self, device=0):
        """Collect the verdict of the asynchronous renders issued so far (raises on dropped rays)."""
        _check(lib().rt_render_check(self._h, int(device)))

    def render_tiles_device(self, cam, params, tiles, device, rgb_ptr, z_ptr, cnt_ptr, stream=None, sync=True,
                            want_stats=True):
        """Render this call's tiles into DEVICE buffers (e.g. torch tensors' data_ptr()).
        stream: None = the library's own stream; otherwise the handle of an EXPLICIT hipStream_t.  0 -- what torch
        reports for its legacy default stream -- is refused: through this argument NULL means "the library's
        stream", so work on the default stream would silently not be ordered with the render; run the caller's
        side under a torch.cuda.Stream and pass its handle (raytracing_folder_amd.dist does)."""
        st = Stats()
        handle = _stream_handle(stream)
        _check(lib().rt_render_tiles_device(self._h, C.byref(cam), C.byref(params), C.byref(tiles), int(device),
                                            handle, C.c_void_p(rgb_ptr),
                                            C.c_void_p(z_ptr), C.c_void_p(cnt_ptr), 1 if sync else 0,
                                            C.byref(st) if want_stats else None))
        return st
