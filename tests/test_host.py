"""CPU tests of the host layer and the C-ABI surface (no compute calls: there is no GPU here).
Golden data: tests/golden/*.npz, produced by the reference's own code (oracle/gen_golden.py)."""
import ctypes as C
import os
import re

import subprocess

import numpy as np
import pytest

from raytracing_folder_amd import capi, photons
from tests import scenes

ROOT = scenes.ROOT


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rt_mi355x.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"rt_status"}
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    lib = capi.lib()
    for name in declared:
        assert getattr(lib, name)
    assert lib.rt_abi_version() == 4


def test_struct_sizes_match_header():
    assert C.sizeof(capi.Camera) == 56 and C.sizeof(capi.Params) == 64
    assert C.sizeof(capi.TileRange) == 16 and C.sizeof(capi.Stats) == 29 * 8
    p = capi.default_params()
    assert (p.min_sample, p.max_sample, p.bounce, p.knn_k, p.shadow_samples) == (4, 8, 4, 400, 4)
    assert p.threshold == np.float32(1e-3) and p.gamma == 2.2 and p.knn_radius == 1.0
    assert (p.photon_count, p.photon_bounce) == (1000000, 8)       # MAX_NUM_OF_PHOTON, PHOTON_BOUNCE (FIN/main.cpp:27,29)
    assert C.sizeof(capi.SetupMs) == 5 * 8


def test_xml_loader_reproduces_reference_transforms(gold):
    s, cam = scenes.load_cornell()
    e = s.export()
    g = gold("node.npz")
    names = ["box_group", "wall_bottom", "wall_top", "wall_back", "wall_left", "wall_right", "teapot", "sphere1"]
    assert list(e["nodes"]["parent"]) == [-1, 0, 1, 1, 1, 1, 1, 0, 0, 0]
    assert list(e["nodes"]["obj_type"]) == [0, 0, 2, 2, 2, 2, 2, 3, 1, 1]
    for i, nm in enumerate(names):
        n = e["nodes"][i + 1]
        for f in ("tm", "itm", "pos"):
            assert n[f].tobytes() == g[f"{nm}_{f}"].tobytes(), (nm, f)
    root = e["nodes"][0]
    assert (root["tm"] == np.eye(3, dtype=np.float32).ravel()).all() and (root["pos"] == 0).all()
    # camera as LoadScene leaves it (dir, up orthonormal)
    assert list(cam.pos) == [0, -60, 12] and list(cam.dir) == [0, 1, 0] and list(cam.up) == [0, 0, 1]
    assert (cam.fov, cam.focaldist, cam.dof, cam.width, cam.height) == (30, 1, 0, 800, 600)
    m = e["materials"]
    assert np.allclose(m["diffuse"][3], [1.0, 0.3, 0.3]) and np.allclose(m["specular"][3], 0.7)
    assert (m["reflection"][4] == 1).all() and (m["refraction"][5] == 1).all() and m["ior"][5] == np.float32(1.52)
    assert (m["specular"][4] == 1).all()              # <specular vakye="1.0"/> -> default (1,1,1)
    l = e["lights"]
    assert l["type"][0] == capi.LIGHT_POINT and l["intensity"][0][0] == np.float32(100.5) and l["size"][0] == 0


def test_obj_loader_and_bvh_match_reference(gold):
    s, _ = scenes.load_cornell()
    m = s.export()["meshes"][0]
    g = gold("mesh_teapot_fin.npz")
    for k in ("v", "f", "vn", "fn", "elements"):
        assert (m[k] == g[k]).all(), k
    assert m["nodes"][1:].tobytes() == g["nodes"][1:].tobytes()
    nodes, el = capi.bvh_build(g["v"], g["f"], 4)
    assert nodes[1:].tobytes() == g["nodes"][1:].tobytes() and (el == g["elements"]).all()


def test_obj_loader_polygons_negative_indices_and_missing_normals(tmp_path):
    obj = tmp_path / "quad.obj"
    obj.write_text("# a quad and a pentagon, relative indices, no normals\n"
                   "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv -0.5 0.5 0\n"
                   "f 1 2 3 4\nf -5 -4 -3 -2 -1\n")
    xml = tmp_path / "s.xml"
    xml.write_text('<xml><scene><object type="obj" name="quad.obj" material="m"/>'
                   '<material type="blinn" name="m"/></scene><camera/></xml>')
    s = capi.Scene()
    s.load_xml(str(xml))
    m = s.export()["meshes"][0]
    assert m["f"].tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 2], [0, 2, 3], [0, 3, 4]]
    assert np.allclose(m["vn"], [[0, 0, 1]] * 5)       # ComputeNormals
    assert (m["fn"] == m["f"]).all()


def test_obj_materials_and_texture_vertices_match_reference(gold):
    """an OBJ node without a material attribute: usemtl regrouping, vt/ft, the .mtl fields and the
    multi-material LoadNode generates from them (rendered as its first sub-material)"""
    s = capi.Scene()
    s.load_xml(os.path.join(scenes.GOLD, "twotone.xml"))
    e = s.export()
    g = gold("mesh_twotone.npz")
    m = e["meshes"][1]                                   # meshes in node order: brickwall, twotone
    for k in ("v", "f", "vn", "fn", "vt", "ft", "elements"):
        assert (m[k] == g[k]).all(), k
    assert m["nodes"][1:].tobytes() == g["nodes"][1:].tobytes()
    assert g["mtl"]["mcfc"].tolist() == [5, 11] and g["map_Kd"].tolist() == ["", "texture_52x37.png"]
    n = e["nodes"]
    assert n["mesh"].tolist() == [-1, -1, 0, 1, 1, -1]
    brick, chrome = e["materials"][n["material"][2]], e["materials"][n["material"][3]]
    assert n["material"][4] == n["material"][5]           # the third OBJ node names a scene material
    gm = g["mtl"]                                         # reference order: chrome (first use), brick
    assert (chrome["diffuse"] == gm["Kd"][0]).all() and (chrome["specular"] == gm["Ks"][0]).all()
    assert chrome["glossiness"] == gm["Ns"][0] and chrome["ior"] == gm["Ni"][0]
    assert (chrome["reflection"] == gm["Ks"][0]).all()                            # illum 6
    assert (chrome["refraction"] == np.float32(1) - gm["Tf"][0]).all()
    assert (brick["diffuse"] == gm["Kd"][1]).all() and (brick["reflection"] == 0).all() and brick["glossiness"] == 25
    maps = e["material_maps"]
    assert maps["texture"][2 * n["material"][2]] == 0 and maps["texture"][2 * n["material"][3]] == capi.MAP_NONE
    assert e["textures"]["width"][0] == 52 and len(e["textures"]) == 1          # shared with the ball's texture
    bw = e["meshes"][0]
    assert bw["vt"].max() == 2 and bw["ft"].shape == bw["f"].shape


@pytest.mark.parametrize("tag", ["k8", "k50", "k400"])
def test_photon_pack_and_balance_match_reference(gold, tag):
    g = gold(f"photon_{tag}.npz")
    ph = g["photons_in"]
    packed = photons.pack_photons(ph[:, :3], ph[:, 3:6], ph[:, 6:9])
    packed["power"] *= g["scale"]
    assert packed.tobytes() == g["packed"].tobytes()
    bal = capi.photon_balance(np.concatenate([np.zeros(1, capi.PHOTON), packed]))
    assert bal[1:].tobytes() == g["balanced"][1:].tobytes()


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 100, 101, 1023, 1024, 1025, 4097, 20000, 20001])
def test_unreachable_photons_are_the_tail_of_the_balanced_heap(n):
    """rt_photon_unreachable (partial BalanceSegment along the root paths of the last heap slots) names exactly the photons the
    full PrepareForIrradianceEstimation puts at heap slots >= 2 * halfStoredPhotons, which LocatePhotons never visits
    (cyPhotonMap.h:217,371) -- also with many equal coordinates, where the partition's tie handling decides"""
    rng = np.random.default_rng(n)
    ph = np.zeros(n + 1, capi.PHOTON)
    pos = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
    if n > 8:
        pos[:, rng.integers(0, 3)] = np.round(pos[:, 0] * 2) / 2          # heavy ties along one axis
    ph["position"][1:] = pos
    ph["power"][1:] = np.arange(1, n + 1)                                 # identifies the record
    bal = capi.photon_balance(ph)
    half = n // 2 - 1
    reach = min(n, max(1, 2 * half - 1))
    want = sorted(bal["power"][reach + 1:].astype(int).tolist())
    got = capi.photon_unreachable(ph)
    assert sorted(ph["power"][got].astype(int).tolist()) == want and len(got) == n - reach
    assert (np.diff(got.astype(int)) > 0).all() if len(got) > 1 else True


def test_photon_dat_round_trip_and_reference_dump(gold, tmp_path):
    """rt_photons_read_dat / write_dat on the dump the reference ships, and the host balance of it
    against the reference's cyPhotonMap"""
    import hashlib
    g = gold("photon_caustic.npz")
    path = os.path.join(scenes.GOLD, "causticmap.dat")
    ph = capi.photons_read_dat(path)
    assert len(ph) == 48141 and ph[1:].tobytes() == open(path, "rb").read()
    capi.photons_write_dat(tmp_path / "copy.dat", ph)
    assert open(tmp_path / "copy.dat", "rb").read() == open(path, "rb").read()
    with open(tmp_path / "ragged.dat", "wb") as f:        # a trailing partial record is dropped
        f.write(ph[1:4].tobytes() + b"xyz")
    assert len(capi.photons_read_dat(tmp_path / "ragged.dat")) == 4
    open(tmp_path / "empty.dat", "wb").close()
    assert len(capi.photons_read_dat(tmp_path / "empty.dat")) == 1
    with pytest.raises(capi.RtError):
        capi.photons_read_dat(tmp_path / "missing.dat")
    bal = capi.photon_balance(ph)
    assert hashlib.sha256(bal[1:].tobytes()).hexdigest() == str(g["balanced_sha256"])


def test_byte_over_255_two_term_constant_gives_the_correctly_rounded_quotient():
    """the device forms Color24 -> Color (byte / 255.0f) as fma(c, r_hi, RN(c * r_lo)) with r_hi = RN(1/255),
    r_lo = RN(1/255 - r_hi): checked here for all 256 bytes in exact rational arithmetic (one rounding per operation)"""
    from fractions import Fraction as Fr
    f = np.float32
    r_hi, r_lo = f(float.fromhex("0x1.010102p-8")), f(float.fromhex("-0x1.fdfdfep-33"))
    assert r_hi == f(1) / f(255) and r_lo == f(1.0 / 255.0 - float(r_hi))

    def rn32(F):                                           # correctly rounded float32 of a Fraction, no ties allowed
        if F == 0:
            return f(0)
        sign = -1 if F < 0 else 1
        F = abs(F)
        g = f(float(F))
        cands = [np.nextafter(g, f(-np.inf)), g, np.nextafter(g, f(np.inf))]
        d = sorted((abs(Fr(float(c)) - F), i) for i, c in enumerate(cands))
        assert d[0][0] != d[1][0]
        return f(sign * float(cands[d[0][1]]))

    wrong_plain = 0
    for c in range(256):
        x = Fr(c)
        t = rn32(x * Fr(float(r_lo)))
        q = rn32(x * Fr(float(r_hi)) + Fr(float(t)))      # the fma: exact product + t, rounded once
        assert q == f(c) / f(255) == rn32(Fr(c, 255))
        wrong_plain += int(f(c) * r_hi != f(c) / f(255))
    assert wrong_plain > 100                               # the plain product is NOT enough


def test_xml_errors_are_reported(tmp_path):
    s = capi.Scene()
    with pytest.raises(capi.RtError) as e:
        s.load_xml(str(tmp_path / "missing.xml"))
    assert e.value.status == -5
    bad = tmp_path / "bad.xml"
    bad.write_text("<xml><scene></scene></xml>")
    with pytest.raises(capi.RtError, match="camera"):
        s.load_xml(str(bad))
    with pytest.raises(capi.RtError):
        s.set_nodes(np.zeros(2, capi.NODE))            # node 0 must have parent -1


def test_render_fails_loudly_without_a_gpu():
    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    s, cam = scenes.load_cornell(64, 48)
    with pytest.raises(capi.RtError) as e:
        s.render(cam, capi.default_params())
    assert e.value.status == -3                        # RT_ERR_NO_DEVICE: no CPU fallback
    with pytest.raises(capi.RtError):
        s.trace_rays(np.zeros((1, 6), np.float32))


def test_cpp_driver_of_the_beginrender_shim_fails_loudly_without_a_gpu(tmp_path):
    """the reference-side contract in C++ (rt::Renderer: LoadScene / BeginRender / IsRenderDone / saveImage)
    links against the product library with plain g++; without a device BeginRender reports the error"""
    import subprocess
    exe = scenes.build_shim_driver(tmp_path)
    r = subprocess.run([exe, os.path.join(scenes.GOLD, "cornell_textured.xml"), str(tmp_path / "a.png"), str(tmp_path / "b.png"),
                        str(tmp_path / "c.png")], capture_output=True, text=True)
    if capi.device_count() == 0:
        assert r.returncode == 4 and "no CPU path" in r.stderr and not (tmp_path / "a.png").exists()
    else:
        assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, str(tmp_path / "missing.xml"), "a", "b", "c"], capture_output=True, text=True)
    assert r.returncode == 3 and "LoadScene failed" in r.stderr


def test_cpp_multi_gpu_driver_builds_against_rccl_and_fails_loudly_without_a_gpu(tmp_path):
    """tests/dist_driver.cpp -- rt_render_tiles_packed_device -> ncclAllGather -> rt_tiles_unpack_device from plain C++ --
    compiles and links against include/rt_mi355x.h, /opt/rocm/include/rccl and librccl; without a gfx950 device it exits
    with an error instead of rendering anything on the CPU"""
    exe = scenes_mod().build_dist_driver(tmp_path)
    r = subprocess.run([exe, scenes_mod().CORNELL, "64", "48", "2", "1000", str(tmp_path / "x.png")], capture_output=True, text=True)
    assert r.returncode != 0 and "no gfx950 device" in r.stderr and not (tmp_path / "x.png").exists()


def test_reference_side_binding_links_into_the_reference_program_and_fails_loudly_without_a_gpu(tmp_path):
    """oracle/_ref/ref_binding_harness_fin (the reference's main.cpp + rt_binding.cpp, `make -C oracle refbinding`): the
    reference's LoadScene reads the product's Cornell file, the binding's BeginRender lowers the scene -- and reports that
    there is no GPU instead of rendering on the CPU"""
    import shutil
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_binding_harness_fin")
    if not os.path.exists(exe):
        pytest.skip("not built (needs the reference tree at build time)")
    data = os.path.dirname(scenes_mod().CORNELL)
    for f in ("cornell.xml", "teapot_tri.obj"):
        shutil.copy(os.path.join(data, f), tmp_path / f)
    r = subprocess.run([exe, "cornell.xml", "out.bin", "32", "24"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 5 and "no HIP device" in r.stderr and not (tmp_path / "out.bin").exists()


def scenes_mod():
    from tests import scenes
    return scenes


def test_synthetic_photon_map_is_well_formed():
    bal = photons.synth_cornell_photon_map(5000, seed=3)
    assert len(bal) == 5001
    p = bal["position"][1:]
    assert p[:, 0].min() >= -15 and p[:, 0].max() <= 15 and p[:, 2].min() >= 0 and p[:, 2].max() <= 24
    # heap order: the root splits the widest axis
    axis = bal["plane_and_dirz"][1] & 3
    assert axis in (0, 1, 2)


def test_image_reader_matches_lodepng(gold):
    """the product's own inflate + PNG unfilter on a zlib-level-9, all-filter-types PNG; golden texels
    are what the reference's lodepng decoded from the same file"""
    g = gold("texture.npz")
    img = capi.image_read_rgb(os.path.join(scenes.GOLD, "texture_52x37.png"))
    assert (img == g["image"]).all()
    # and the PNG writer (RenderImage::SavePNG replacement) round-trips through the reader
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        for arr in (g["image"], g["image"][:, :, 0]):
            path = os.path.join(td, "x.png")
            capi.image_write_png(path, arr)
            back = capi.image_read_rgb(path)
            assert (back == (arr if arr.ndim == 3 else np.repeat(arr[:, :, None], 3, 2))).all()


def test_xml_textures_and_map_transforms(gold):
    from oracle import orc
    g = gold("texture.npz")
    s = capi.Scene()
    s.load_xml(os.path.join(scenes.GOLD, "cornell_textured.xml"))
    e = s.export()
    tex = e["textures"]
    # TextureList order: background file texture first (shared by name), then the checkers as they appear
    assert tex["type"].tolist() == [capi.TEX_FILE, capi.TEX_CHECKER, capi.TEX_CHECKER]
    assert (tex["width"][0], tex["height"][0]) == (52, 37)
    t0 = e["texels"][tex["texel_offset"][0]: tex["texel_offset"][0] + 52 * 37 * 3].reshape(37, 52, 3)
    assert (t0 == g["image"]).all()
    names = ["floor", "back", "pot", "ball", "glass"]
    maps = e["material_maps"].reshape(len(names), 2)
    assert maps["texture"][:, 0].tolist() == [1, 0, 0, 0, capi.MAP_NONE]
    assert maps["texture"][:, 1].tolist() == [capi.MAP_NONE, 2, capi.MAP_NONE, capi.MAP_NONE, capi.MAP_NONE]
    # the "back" diffuse map carries exactly the transform the golden vectors were made with
    # (scale .25 .5 1, rotate z 30, translate .1 -.2): TransformTo agrees with the reference bit for bit
    assert orc.texmap_transform(maps[1, 0], g["uvw"]).tobytes() == g["map_transform"].tobytes()
    assert e["env_map"]["texture"][0] == 0 and e["bg_map"]["texture"][0] == 0
    assert np.allclose(e["materials"]["diffuse"][3], 0.8)


def test_zbuffer_and_sample_count_images_match_reference(gold):
    """rt_image_zbuffer / rt_image_sample_count (host code behind the C ABI, used by rt::RenderImage) against
    RenderImage::ComputeZBufferImage / ComputeSampleCountImage of the reference itself: bit-exact integer maps"""
    g = gold("zimage.npz")
    for tag in "abc":
        assert np.array_equal(capi.zbuffer_image(g[f"z_{tag}"]), g[f"zimg_{tag}"])
        sc, smax = capi.sample_count_image(g[f"cnt_{tag}"])
        assert np.array_equal(sc, g[f"scimg_{tag}"]) and smax == int(g[f"smax_{tag}"])
    # degenerate frames: nothing hit (all BIGFLOAT) -> zeros; a single depth -> 0/0, NaN -> 0 (UB in the reference)
    assert (capi.zbuffer_image(np.full((3, 4), 1.0e30, np.float32)) == 0).all()
    assert (capi.zbuffer_image(np.full((3, 4), 7.0, np.float32)) == 0).all()


def _png(path, w, h, depth, ctype, rows, plte=None):
    """minimal PNG writer for the reader test: `rows` are the packed scanlines (bytes), filter type 0"""
    import struct
    import zlib

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    raw = b"".join(b"\0" + bytes(r) for r in rows)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if plte is not None:
        png += chunk(b"PLTE", bytes(plte))
    png += chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)


def test_image_reader_low_bit_depths_and_hostile_headers(tmp_path):
    """1/2/4-bit grey and palette PNGs (the reference's own sample-count images are 1-bit grey: lodepng picks
    the smallest type) decode like lodepng's LCT_RGB conversion; absurd IHDR sizes and bad PPM magics are
    refused with an error instead of an exception or an overflow"""
    rng = np.random.default_rng(5)
    w, h = 13, 5
    for depth in (1, 2, 4):
        vals = rng.integers(0, 1 << depth, (h, w))
        per = 8 // depth
        rows = []
        for y in range(h):
            row = bytearray((w + per - 1) // per)
            for x in range(w):
                row[x // per] |= int(vals[y, x]) << (8 - depth - depth * (x % per))
            rows.append(row)
        path = str(tmp_path / f"g{depth}.png")
        _png(path, w, h, depth, 0, rows)
        got = capi.image_read_rgb(path)
        want = (vals * 255 // ((1 << depth) - 1)).astype(np.uint8)
        assert np.array_equal(got, np.repeat(want[:, :, None], 3, 2))
        plte = rng.integers(0, 256, (1 << depth, 3)).astype(np.uint8)
        _png(path, w, h, depth, 3, rows, plte.tobytes())
        assert np.array_equal(capi.image_read_rgb(path), plte[vals])
    # the reference-held 8-bit grey z image decodes (it is a fixture of tests/test_reference_images.py)
    assert capi.image_read_rgb(os.path.join(scenes.GOLD, "ref_prj13_boxzbuff.png")).shape == (600, 800, 3)
    big = str(tmp_path / "big.png")
    _png(big, 0x7FFFFFFF, 0x7FFFFFFF, 8, 2, [b"\0\0\0"])
    with pytest.raises(capi.RtError):
        capi.image_read_rgb(big)
    for magic in (b"P5", b"X6", b"PX"):
        ppm = str(tmp_path / "bad.ppm")
        with open(ppm, "wb") as f:
            f.write(magic + b"\n2 2\n255\n" + bytes(12))
        with pytest.raises(capi.RtError):
            capi.image_read_rgb(ppm)
    ppm = str(tmp_path / "ok.ppm")
    with open(ppm, "wb") as f:
        f.write(b"P6\n2 2\n255\n" + bytes(range(12)))
    assert capi.image_read_rgb(ppm).ravel().tolist() == list(range(12))


def test_reference_side_binding_compiles_against_the_reference_headers():
    """raytracing_folder_amd/binding/rt_binding.cpp (BeginRender / StopRender / saveImage on
    the C ABI, with real LowerMaterials / LowerLights / LowerTextures) is syntax-checked with g++ against the
    reference's own headers of both snapshots.  Only where the reference tree exists (the build container)."""
    import subprocess
    if not os.path.isdir("/root/reference/RayTracingFinal/RayTracingFinal/include"):
        pytest.skip("reference tree absent (GPU box)")
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "binding-check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "syntax OK" in r.stdout
