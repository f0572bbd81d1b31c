#!/usr/bin/env python3
"""gen_golden.py -- TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz.

Runs oracle/_ref/ref_harness_{fin,p13} (the reference's own headers compiled where they lie
under /root/reference, see oracle/Makefile) on seeded inputs and stores inputs + the
reference's outputs as small fixtures.  Only runs in the build container (the reference tree
does not exist on the GPU box); the fixtures are committed.

    python oracle/gen_golden.py            # regenerate everything
"""
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("RT_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "raytracing_folder_amd", "data")     # workload inputs the product ships (bench.py, smoke())
HARNESS = {m: os.path.join(ROOT, "oracle", "_ref", f"ref_harness_{m}") for m in ("fin", "p13")}
TEAPOT = os.path.join(REF, "RayTracingFinal", "RayTracingFinal", "data", "teapot.obj")

HITREC = np.dtype([("hit", "<i4"), ("z", "<f4"), ("p", "<f4", 3), ("N", "<f4", 3), ("front", "<i4")])
BVHNODE = np.dtype([("box", "<f4", 6), ("data", "<u4")])
PHOTON = np.dtype([("position", "<f4", 3), ("power", "<f4"), ("color", "u1", 3),
                   ("plane_and_dirz", "u1"), ("dir_x", "<i2"), ("dir_y", "<i2")])
assert HITREC.itemsize == 36 and BVHNODE.itemsize == 28 and PHOTON.itemsize == 24


def run(model, cmd, payload, *extra):
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(payload)
        subprocess.run([HARNESS[model], cmd, fin, fout, *extra], check=True,
                       stdout=subprocess.DEVNULL)
        with open(fout, "rb") as f:
            return f.read()


def rays_with_z(rng, n, spread=3.0):
    """rays: origins in a cube, directions of mixed length; some axis-parallel; z0 mostly BIG."""
    o = rng.uniform(-spread, spread, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d *= rng.choice([0.25, 1.0, 1.0, 4.0], size=(n, 1)).astype(np.float32)
    # aim most rays roughly at the origin so that they hit something
    aim = rng.random(n) < 0.7
    tgt = rng.uniform(-0.9, 0.9, (n, 3)).astype(np.float32)
    d[aim] = (tgt[aim] - o[aim]) * rng.uniform(0.2, 2.0, (aim.sum(), 1)).astype(np.float32)
    # exact zeros in direction components
    zc = rng.random(n) < 0.05
    d[zc, rng.integers(0, 3, zc.sum())] = 0.0
    # origins inside the unit sphere
    ins = rng.random(n) < 0.15
    o[ins] = rng.uniform(-0.5, 0.5, (ins.sum(), 3)).astype(np.float32)
    z0 = np.full(n, 1.0e30, np.float32)
    sm = rng.random(n) < 0.2
    z0[sm] = rng.uniform(0.1, 5.0, sm.sum()).astype(np.float32)
    return np.concatenate([o, d, z0[:, None]], axis=1).astype(np.float32)


def gen_prims(model):
    rng = np.random.default_rng(101)
    n = 4096
    r = rays_with_z(rng, n)
    out = run(model, "prims", struct.pack("<i", n) + r.tobytes())
    rec = np.frombuffer(out, HITREC)
    np.savez_compressed(os.path.join(GOLD, f"prims_{model}.npz"), rays=r,
                        sphere=rec[:n], plane=rec[n:])


def gen_box():
    rng = np.random.default_rng(102)
    n = 4096
    lo = rng.uniform(-2, 1, (n, 3)).astype(np.float32)
    hi = lo + rng.uniform(0, 2, (n, 3)).astype(np.float32)
    flat = rng.random(n) < 0.1
    hi[flat, 2] = lo[flat, 2]                       # zero-thickness boxes (axis-aligned triangles)
    r = rays_with_z(rng, n)
    tmax = np.where(rng.random(n) < 0.8, np.float32(1.0e30), rng.uniform(0.1, 6, n)).astype(np.float32)
    rec = np.concatenate([lo, hi, r[:, :6], tmax[:, None]], axis=1).astype(np.float32)
    out = run("fin", "box", struct.pack("<i", n) + rec.tobytes())
    np.savez_compressed(os.path.join(GOLD, "box.npz"), rec=rec, hit=np.frombuffer(out, "<i4"))


def parse_mesh(out):
    nv, nf, nvn, nnodes = struct.unpack_from("<4i", out, 0)
    off = 16
    def take(dt, count):
        nonlocal off
        a = np.frombuffer(out, dt, count, off)
        off += a.nbytes
        return a
    v = take("<f4", nv * 3).reshape(nv, 3)
    f = take("<u4", nf * 3).reshape(nf, 3)
    vn = take("<f4", nvn * 3).reshape(nvn, 3)
    fn = take("<u4", nf * 3).reshape(nf, 3)
    nodes = take(BVHNODE, nnodes)
    elements = take("<u4", nf)
    hits = np.frombuffer(out, HITREC, -1, off)
    return v, f, vn, fn, nodes, elements, hits


def write_tri_obj(path, v, f, vn, fn):
    """own OBJ export of the loaded teapot arrays (already fan-triangulated by the reference's
    loader); %.9g round-trips float32 exactly."""
    with open(path, "w") as o:
        o.write("# teapot mesh arrays as loaded by the reference's cyTriMesh (triangulated)\n")
        for p in v:
            o.write("v %.9g %.9g %.9g\n" % tuple(p))
        for p in vn:
            o.write("vn %.9g %.9g %.9g\n" % tuple(p))
        for a, b in zip(f, fn):
            o.write("f %d//%d %d//%d %d//%d\n" % (a[0] + 1, b[0] + 1, a[1] + 1, b[1] + 1, a[2] + 1, b[2] + 1))


def gen_mesh(model):
    rng = np.random.default_rng(103)
    n = 4096
    # teapot bounds are roughly [-8,9] x [-5,5] x [0,8]
    o = rng.uniform(-20, 20, (n, 3)).astype(np.float32)
    tgt = np.stack([rng.uniform(-8, 9, n), rng.uniform(-5, 5, n), rng.uniform(0, 8, n)], 1).astype(np.float32)
    d = tgt - o
    nrm = rng.random(n) < 0.5
    d[nrm] /= np.linalg.norm(d[nrm], axis=1, keepdims=True)
    ins = rng.random(n) < 0.1                       # origins inside the teapot's box
    o[ins] = tgt[ins]
    d[ins] = rng.normal(size=(ins.sum(), 3))
    z0 = np.full(n, 1.0e30, np.float32)
    sm = rng.random(n) < 0.1
    z0[sm] = rng.uniform(0.3, 1.2, sm.sum())
    r = np.concatenate([o, d.astype(np.float32), z0[:, None]], 1).astype(np.float32)
    out = run(model, "mesh", struct.pack("<i", n) + r.tobytes(), TEAPOT)
    v, f, vn, fn, nodes, elements, hits = parse_mesh(out)
    assert len(hits) == n
    np.savez_compressed(os.path.join(GOLD, f"mesh_teapot_{model}.npz"), rays=r, hits=hits,
                        **({"v": v, "f": f, "vn": vn, "fn": fn, "nodes": nodes, "elements": elements}
                           if model == "fin" else {}))
    if model == "fin":
        write_tri_obj(os.path.join(DATA, "teapot_tri.obj"), v, f, vn, fn)
        # the triangulated export must load back to identical arrays through the reference loader
        out2 = run(model, "mesh", struct.pack("<i", 0), os.path.join(DATA, "teapot_tri.obj"))
        v2, f2, vn2, fn2, nodes2, el2, _ = parse_mesh(out2)
        assert (v2 == v).all() and (f2 == f).all() and (vn2 == vn).all() and (fn2 == fn).all()
        assert nodes2.tobytes() == nodes.tobytes() and (el2 == elements).all()
    print(f"mesh[{model}]: nv={len(v)} nf={len(f)} nvn={len(vn)} nnodes={len(nodes)} hits={int(hits['hit'].sum())}")


def gen_mesh_mtl():
    """tests/golden/twotone.obj + .mtl through the reference loader with loadMtl = true: face
    regrouping by material, texture vertices/faces, the .mtl fields; and, for the PROJ13 triangle,
    the uvw each hit carries."""
    obj = os.path.join(GOLD, "twotone.obj")
    rng = np.random.default_rng(211)
    n = 2048
    o = rng.uniform(-4, 4, (n, 3)).astype(np.float32)
    o[:, 2] = rng.uniform(-3, 5, n)
    tgt = np.stack([rng.uniform(-1, 1, n), rng.uniform(-1, 1.6, n), rng.uniform(0, 2, n)], 1).astype(np.float32)
    d = (tgt - o).astype(np.float32)
    r = np.concatenate([o, d, np.full((n, 1), 1.0e30, np.float32)], 1).astype(np.float32)
    res = {}
    for model in ("fin", "p13"):
        with tempfile.TemporaryDirectory() as td:
            open(td + "/in.bin", "wb").write(struct.pack("<i", n) + r.tobytes())
            subprocess.run([HARNESS[model], "meshmtl", td + "/in.bin", td + "/out.bin", obj,
                            td + "/mtl.bin"], check=True, stdout=subprocess.DEVNULL)
            out = open(td + "/out.bin", "rb").read()
            mb = open(td + "/mtl.bin", "rb").read()
            uv = np.fromfile(td + "/mtl.bin.uvw", "<f4").reshape(n, 3)
        v, f, vn, fn, nodes, elements, hits = parse_mesh(out)
        nm, nvt = struct.unpack_from("<2i", mb, 0)
        off = 8
        mt = np.zeros(nm, [("Kd", "<f4", 3), ("Ks", "<f4", 3), ("Tf", "<f4", 3), ("Ns", "<f4"), ("Ni", "<f4"),
                           ("illum", "<i4"), ("mcfc", "<i4")])
        names = []
        for i in range(nm):
            mt[i] = np.frombuffer(mb, mt.dtype, 1, off)[0]
            off += mt.dtype.itemsize
            names.append([mb[off:off + 256].split(b"\0")[0].decode(), mb[off + 256:off + 512].split(b"\0")[0].decode()])
            off += 512
        vt = np.frombuffer(mb, "<f4", nvt * 3, off).reshape(nvt, 3); off += vt.nbytes
        ft = np.frombuffer(mb, "<u4", len(f) * 3, off).reshape(len(f), 3)
        res[model] = dict(v=v, f=f, vn=vn, fn=fn, nodes=nodes, elements=elements, hits=hits, uvw=uv, mtl=mt,
                          map_Kd=np.array([x[0] for x in names]), map_Ks=np.array([x[1] for x in names]), vt=vt, ft=ft)
    a, b = res["fin"], res["p13"]
    for k in ("v", "f", "vn", "fn", "elements", "vt", "ft", "map_Kd", "map_Ks"):
        assert (a[k] == b[k]).all(), k
    assert a["nodes"].tobytes() == b["nodes"].tobytes() and a["mtl"].tobytes() == b["mtl"].tobytes()
    assert (a["uvw"] == 0.5).all()                   # the FINAL triangle never writes uvw
    np.savez_compressed(os.path.join(GOLD, "mesh_twotone.npz"), rays=r, hits_fin=a["hits"], hits_p13=b["hits"],
                        uvw_p13=b["uvw"], **{k: a[k] for k in ("v", "f", "vn", "fn", "nodes", "elements", "vt", "ft", "mtl",
                                                              "map_Kd", "map_Ks")})
    print(f"mesh_twotone: nf={len(a['f'])} nm={len(a['mtl'])} hits fin={int(a['hits']['hit'].sum())} p13={int(b['hits']['hit'].sum())}")


NODE_OPS = {
    # (kind, a0..a3): 0 scale xyz, 1 rotate axis xyz + degrees, 2 translate xyz -- Cornell scene.xml
    "box_group": [(2, 0, 0, 12, 0)],
    "wall_bottom": [(0, 32, 32, 32, 0), (2, 0, 0, -12, 0)],
    "wall_top": [(0, 32, 32, 32, 0), (1, 1, 0, 0, 180), (2, 0, 0, 12, 0)],
    "wall_back": [(0, 32, 32, 32, 0), (1, 1, 0, 0, 90), (2, 0, 20, 0, 0)],
    "wall_left": [(0, 32, 32, 32, 0), (1, 0, 1, 0, 90), (2, -15, 0, 0, 0)],
    "wall_right": [(0, 32, 32, 32, 0), (1, 0, 1, 0, -90), (2, 15, 0, 0, 0)],
    "teapot": [(0, 0.8, 0.8, 0.8, 0), (1, 0, 0, 1, -30), (2, 2, 5, 0, 0)],
    "sphere1": [(0, 4, 4, 4, 0), (2, 8, -6, 4, 0)],
    "identity": [],
    "skew": [(0, 1, 2, 3, 0), (1, 1, 1, 0, 33), (2, 0.5, -1, 2, 0), (1, 0, 2, 1, -70)],
}


def gen_node():
    rng = np.random.default_rng(104)
    n = 256
    res = {}
    for name, ops in NODE_OPS.items():
        rays = np.concatenate([rng.uniform(-30, 30, (n, 3)), rng.normal(size=(n, 3))], 1).astype(np.float32)
        hits = np.concatenate([rng.uniform(-1, 1, (n, 3)), rng.normal(size=(n, 3))], 1).astype(np.float32)
        payload = struct.pack("<i", len(ops))
        for op in ops:
            payload += struct.pack("<i4f", int(op[0]), *[float(x) for x in op[1:]])
        payload += struct.pack("<i", n) + rays.tobytes() + hits.tobytes()
        out = np.frombuffer(run("fin", "node", payload), "<f4")
        res[name + "_ops"] = np.array(ops, np.float64).reshape(len(ops), 5)
        res[name + "_tm"] = out[0:9]
        res[name + "_itm"] = out[9:18]
        res[name + "_pos"] = out[18:21]
        res[name + "_rays"] = rays
        res[name + "_hits"] = hits
        res[name + "_rays_local"] = out[21:21 + 6 * n].reshape(n, 6)
        res[name + "_hits_parent"] = out[21 + 6 * n:21 + 12 * n].reshape(n, 6)
    np.savez_compressed(os.path.join(GOLD, "node.npz"), **res)


def gen_misc():
    rng = np.random.default_rng(105)
    n = 512
    rgb = rng.uniform(-0.2, 1.3, (n, 3)).astype(np.float32)
    rgb[:8] = np.array([[0, 0, 0], [1, 1, 1], [0.999999, 0.5, 0.25], [1 / 255, 2 / 255, 254 / 255],
                        [0.0039, 0.00393, 1.004], [-1, 2, 0.5], [0.2, 0.4, 0.6], [255, 1e9, -1e9]], np.float32)
    out = run("fin", "misc", struct.pack("<i", n) + rgb.tobytes())
    h2 = np.frombuffer(out, "<f4", n, 0)
    h3 = np.frombuffer(out, "<f4", n, 4 * n)
    c24 = np.frombuffer(out, "u1", 3 * n, 8 * n).reshape(n, 3)
    np.savez_compressed(os.path.join(GOLD, "misc.npz"), rgb=rgb, halton2=h2, halton3=h3, color24=c24)


def synth_photons(rng, n):
    """photons on the floor/walls of a 20 x 20 x 10 room with a dense patch (so that some
    queries find far more than k photons and some far fewer)."""
    pos = np.empty((n, 3), np.float32)
    which = rng.integers(0, 3, n)
    u, v = rng.uniform(-10, 10, n), rng.uniform(-10, 10, n)
    dense = rng.random(n) < 0.35
    u[dense] = rng.normal(2.0, 1.2, dense.sum())
    v[dense] = rng.normal(-3.0, 1.2, dense.sum())
    pos[:, 0] = np.where(which == 1, -10, u)
    pos[:, 1] = np.where(which == 2, 10, np.where(which == 1, u, v))
    pos[:, 2] = np.where(which == 0, 0, np.abs(v) * 0.5)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pw = rng.uniform(0.05, 1.0, (n, 3)) * rng.uniform(0.5, 30, (n, 1))
    return np.concatenate([pos, d, pw], 1).astype(np.float32)


def gen_photon(tag, npho, k, radius, nq, seed):
    rng = np.random.default_rng(seed)
    ph = synth_photons(rng, npho)
    scale = np.float32(4 * np.pi / npho)
    qi = rng.integers(0, npho, nq)
    qpos = ph[qi, :3] + rng.normal(0, 0.05, (nq, 3)).astype(np.float32)
    qn = rng.normal(size=(nq, 3))
    qn /= np.linalg.norm(qn, axis=1, keepdims=True)
    # most normals face the stored directions (dir . N < 0 accepted)
    flip = (ph[qi, 3:6] * qn).sum(1) > 0
    qn[flip & (rng.random(nq) < 0.8)] *= -1
    q = np.concatenate([qpos, qn], 1).astype(np.float32)
    payload = (struct.pack("<i", npho) + ph.tobytes() + struct.pack("<fif", scale, k, radius) +
               struct.pack("<i", nq) + q.tobytes())
    out = run("fin", "photon", payload)
    off = 0
    packed = np.frombuffer(out, PHOTON, npho, off); off += npho * 24
    balanced = np.frombuffer(out, PHOTON, npho + 1, off); off += (npho + 1) * 24
    half = struct.unpack_from("<i", out, off)[0]; off += 4
    dec = np.frombuffer(out, "<f4", npho * 6, off).reshape(npho, 6); off += npho * 24
    res = np.frombuffer(out, "<f4", nq * 6, off).reshape(nq, 6)
    np.savez_compressed(os.path.join(GOLD, f"photon_{tag}.npz"), photons_in=ph, scale=scale, k=k,
                        radius=np.float32(radius), queries=q, packed=packed, balanced=balanced,
                        half=half, decoded=dec, result=res)
    nz = int((res[:, :3].sum(1) > 0).sum())
    print(f"photon[{tag}]: n={npho} k={k} r={radius} half={half} queries with light: {nz}/{nq}")


def gen_photon_dat():
    """the photon dump the reference itself holds (PhotonMap/PhotonMap/causticmap.dat, 48 140 records of
    its Cornell scene written by main.cpp:397-400): copied as a data fixture, balanced and queried by the
    reference's cyPhotonMap."""
    import hashlib, shutil
    src = os.path.join(REF, "PhotonMap", "PhotonMap", "causticmap.dat")
    dst = os.path.join(GOLD, "causticmap.dat")
    shutil.copyfile(src, dst)
    os.chmod(dst, 0o644)
    raw = np.fromfile(dst, PHOTON)
    rng = np.random.default_rng(311)
    out = {}
    for tag, k, radius, nq in (("k100", 100, 1.5, 384), ("k400", 400, 4.0, 192)):
        qi = rng.integers(0, len(raw), nq)
        qpos = raw["position"][qi] + rng.normal(0, 0.05, (nq, 3)).astype(np.float32)
        # normals: the wall the photon sits on, guessed from the Cornell box, else random
        qn = rng.normal(size=(nq, 3))
        qn /= np.linalg.norm(qn, axis=1, keepdims=True)
        p = raw["position"][qi]
        qn[np.abs(p[:, 2]) < 1e-3] = (0, 0, 1)
        qn[np.abs(p[:, 2] - 24) < 1e-3] = (0, 0, -1)
        qn[np.abs(p[:, 0] + 15) < 1e-3] = (1, 0, 0)
        qn[np.abs(p[:, 0] - 15) < 1e-3] = (-1, 0, 0)
        qn[np.abs(p[:, 1] - 20) < 1e-3] = (0, -1, 0)
        q = np.concatenate([qpos, qn], 1).astype(np.float32)
        res = run("fin", "photondat", struct.pack("<ifi", k, radius, nq) + q.tobytes(), dst)
        n = struct.unpack_from("<i", res, 0)[0]
        assert n == len(raw)
        bal = np.frombuffer(res, PHOTON, n + 1, 4)
        half = struct.unpack_from("<i", res, 4 + (n + 1) * 24)[0]
        r = np.frombuffer(res, "<f4", nq * 6, 8 + (n + 1) * 24).reshape(nq, 6)
        out.update({f"q_{tag}": q, f"res_{tag}": r, f"k_{tag}": k, f"radius_{tag}": np.float32(radius)})
        digest = hashlib.sha256(bal[1:].tobytes()).hexdigest()
        print(f"photon_dat[{tag}]: n={n} half={half} lit {int((r[:, :3].sum(1) > 0).sum())}/{nq}")
    np.savez_compressed(os.path.join(GOLD, "photon_caustic.npz"), n=n, half=half, balanced_sha256=digest,
                        balanced_head=bal[:64], **out)


def write_png_all_filters(path, img):
    """PNG written here (not by the product): every scanline filter type in turn, zlib level 9
    (dynamic Huffman) -- exercises the product's own inflate + unfilter against lodepng's."""
    import zlib
    h, w, _ = img.shape
    raw = bytearray()
    prev = np.zeros(w * 3, np.int32)
    for y in range(h):
        cur = img[y].astype(np.int32).ravel()
        left = np.concatenate([np.zeros(3, np.int32), cur[:-3]])
        upleft = np.concatenate([np.zeros(3, np.int32), prev[:-3]])
        ft = y % 5
        if ft == 0: out = cur
        elif ft == 1: out = cur - left
        elif ft == 2: out = cur - prev
        elif ft == 3: out = cur - (left + prev) // 2
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            out = cur - pred
        raw.append(ft)
        raw += bytes((out % 256).astype(np.uint8))
        prev = cur
    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
    comp = zlib.compress(bytes(raw), 9)
    png += chunk(b"IDAT", comp[: len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:]) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)


def gen_texture():
    rng = np.random.default_rng(109)
    h, w = 37, 52
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 5 + yy * 3) % 256, (xx * yy) % 256, 255 - (xx * 7) % 256], 2).astype(np.uint8)
    img[10:20, 10:30] = rng.integers(0, 256, (10, 20, 3))          # noise: poorly compressible
    path = os.path.join(GOLD, "texture_52x37.png")
    write_png_all_filters(path, img)
    n = 2048
    uvw = rng.uniform(-2.5, 3.5, (n, 3)).astype(np.float32)
    uvw[:64, :2] = rng.choice([0.0, 0.5, 1.0, -1.0, 0.25, 0.999999, 1.000001], (64, 2))
    ops = [(0, 0.25, 0.5, 1.0, 0), (1, 0, 0, 1, 30), (2, 0.1, -0.2, 0, 0)]
    payload = struct.pack("<i", len(ops))
    for op in ops:
        payload += struct.pack("<i4f", int(op[0]), *[float(x) for x in op[1:]])
    c12 = np.array([0.1, 0.2, 0.3, 0.9, 0.8, 0.7], np.float32)
    payload += c12.tobytes() + struct.pack("<i", n) + uvw.tobytes()
    out = run("fin", "tex", payload, path)
    wi, hi = struct.unpack_from("<2i", out, 0)
    assert (wi, hi) == (w, h)
    off = 8
    def take(count):
        nonlocal off
        a = np.frombuffer(out, "<f4", count, off)
        off += a.nbytes
        return a
    texels = take(w * h * 3).reshape(h, w, 3)
    assert np.array_equal(np.rint(texels * 255).astype(np.uint8), img)       # lodepng decodes what was written
    np.savez_compressed(os.path.join(GOLD, "texture.npz"), image=img, uvw=uvw, ops=np.array(ops, np.float64), colors=c12,
                        file_sample=take(n * 3).reshape(n, 3), checker_sample=take(n * 3).reshape(n, 3),
                        map_transform=take(n * 3).reshape(n, 3), env_coord=take(n * 3).reshape(n, 3))
    print("texture: 52x37 PNG,", n, "samples")


def gen_zimage():
    """RenderImage::ComputeZBufferImage / ComputeSampleCountImage (scene.h:591-637) on seeded buffers."""
    rng = np.random.default_rng(401)
    res = {}
    for tag, (w, h) in (("a", (53, 31)), ("b", (16, 9)), ("c", (40, 25))):
        z = rng.uniform(5.0, 90.0, (h, w)).astype(np.float32)
        z[rng.random((h, w)) < 0.2] = np.float32(1.0e30)            # BIGFLOAT = missed pixels
        cnt = np.where(rng.random((h, w)) < 0.3, 255, 0).astype(np.uint8)
        if tag == "b":
            cnt[:] = 255                                             # smax == smin -> all zeros
        if tag == "c":
            cnt = rng.integers(0, 256, (h, w)).astype(np.uint8)      # the general formula
            z[0, 0] = np.float32(1.0e30); z[1, 1] = np.float32(5.0); z[2, 2] = np.float32(90.0)
        out = run("fin", "zimg", struct.pack("<2i", w, h) + z.tobytes() + cnt.tobytes())
        n = w * h
        res.update({f"z_{tag}": z, f"cnt_{tag}": cnt, f"zimg_{tag}": np.frombuffer(out, "u1", n, 0).reshape(h, w),
                    f"scimg_{tag}": np.frombuffer(out, "u1", n, n).reshape(h, w),
                    f"smax_{tag}": struct.unpack_from("<i", out, 2 * n)[0]})
        out13 = run("p13", "zimg", struct.pack("<2i", w, h) + z.tobytes() + cnt.tobytes())
        assert out13 == out                                          # both snapshots carry the same code
    np.savez_compressed(os.path.join(GOLD, "zimage.npz"), **res)
    print("zimage: 3 cases")


ILLUM_IN = np.dtype([("intensity", "<f4", 3), ("position", "<f4", 3), ("size", "<f4"), ("p", "<f4", 3),
                     ("seed", "<u4"), ("nscript", "<i4"), ("script", "<f4", 8)])
ILLUM_OUT = np.dtype([("result", "<f4", 3), ("ncalls", "<i4"), ("log", "<f4", (20, 7)), ("rand", "<i4", 64)])


def gen_illum():
    """PointLight::Illuminate of both snapshots (FIN/include/lights.h:67-131, P13/include/lights.h:65-91)
    with the harness's recording Shadow double and a captured rand() stream."""
    rng = np.random.default_rng(402)
    n = 384
    a = np.zeros(n, ILLUM_IN)
    a["intensity"] = rng.uniform(0.2, 120.0, (n, 3))
    a["position"] = rng.uniform(-10, 10, (n, 3)) + np.array([0, 0, 20])
    a["size"] = rng.choice([0.0, 0.5, 2.0, 5.0], n)
    a["p"] = rng.uniform(-15, 15, (n, 3))
    a["p"][:32, 0] = a["position"][:32, 0] - rng.uniform(1, 30, 32)       # dir.x > 0.8: the other basis branch
    a["seed"] = rng.integers(1, 2 ** 31, n)
    kinds = rng.integers(0, 4, n)
    for i in range(n):
        if kinds[i] == 0: sc = [1.0]                                        # fully lit: no escalation
        elif kinds[i] == 1: sc = [0.0]                                      # fully shadowed
        elif kinds[i] == 2: sc = list(rng.integers(0, 2, 8).astype(float))  # mixed -> FIN escalates to 16
        else: sc = [1.0, 1.0, 1.0, 0.0, 1.0, 0.0, 0.0, 1.0]
        a["nscript"][i] = len(sc)
        a["script"][i, :len(sc)] = sc
    res = {"cases": a}
    for model in ("fin", "p13"):
        out = np.frombuffer(run(model, "illum", struct.pack("<i", n) + a.tobytes()), ILLUM_OUT)
        assert len(out) == n
        res[f"out_{model}"] = out
        print(f"illum[{model}]: shadow calls per case: {sorted(set(out['ncalls'].tolist()))}")
    np.savez_compressed(os.path.join(GOLD, "illum.npz"), **res)


PB_IN = np.dtype([("diffuse", "<f4", 3), ("specular", "<f4", 3), ("reflection", "<f4", 3), ("refraction", "<f4", 3),
                  ("absorption", "<f4", 3), ("glossiness", "<f4"), ("ior", "<f4"), ("ray", "<f4", 6),
                  ("hit_p", "<f4", 3), ("hit_N", "<f4", 3), ("hit_z", "<f4"), ("front", "<i4"), ("c", "<f4", 3),
                  ("seed", "<u4"), ("reflection_glossiness", "<f4"), ("refraction_glossiness", "<f4")])
PB_OUT = np.dtype([("ret", "<i4"), ("ray", "<f4", 6), ("c", "<f4", 3), ("rand", "<i4", 8)])


def gen_pbounce():
    """MtlBlinn::RandomPhotonBounce, Attenuation, createCoordinateSystem (FIN/include/materials.h:50-256)."""
    rng = np.random.default_rng(403)
    n = 4096
    a = np.zeros(n, PB_IN)
    kind = rng.integers(0, 5, n)
    a["diffuse"] = rng.uniform(0.05, 0.9, (n, 3)); a["specular"] = rng.uniform(0, 0.8, (n, 3))
    a["glossiness"] = rng.choice([1.0, 20.0, 50.0], n); a["ior"] = 1.0
    mirror, glass, tinted, black = kind == 1, kind == 2, kind == 3, kind == 4
    a["diffuse"][mirror | glass] = 0; a["specular"][mirror | glass] = 0.8
    a["reflection"][mirror] = 0.8
    a["refraction"][glass | tinted] = rng.uniform(0.5, 0.9, ((glass | tinted).sum(), 3))
    a["ior"][glass | tinted] = rng.choice([1.52, 1.33, 2.4], (glass | tinted).sum())
    a["absorption"][tinted] = rng.uniform(0.0, 0.3, (tinted.sum(), 3))
    a["reflection"][tinted] = rng.uniform(0, 0.3, (tinted.sum(), 3))
    a["diffuse"][black] = 0; a["absorption"][black] = 0.2                     # nothing but absorption
    # glossy branches (materials.h:183-213): a third of the mirror / glass / tinted cases sample a hemisphere instead
    glossy = (mirror | glass | tinted) & (rng.random(n) < 0.35)
    a["reflection_glossiness"][glossy] = rng.choice([0.5, 5.0, 40.0], glossy.sum())
    a["refraction_glossiness"][glossy & (rng.random(n) < 0.6)] = rng.choice([0.5, 5.0, 40.0], 1)
    N = rng.normal(size=(n, 3)); N /= np.linalg.norm(N, axis=1, keepdims=True)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    a["front"] = (rng.random(n) < 0.7).astype(np.int32)
    facing = (d * N).sum(1) < 0                                                  # front hits: the ray runs against N
    flip = facing != (a["front"] == 1)
    d[flip] *= -1
    a["ray"][:, :3] = rng.uniform(-10, 10, (n, 3)); a["ray"][:, 3:] = d
    a["hit_p"] = rng.uniform(-10, 10, (n, 3)); a["hit_N"] = N
    a["hit_z"] = rng.uniform(0.1, 40, n)
    a["c"] = rng.uniform(0.1, 100, (n, 3))
    a["seed"] = rng.integers(1, 2 ** 31, n)
    out = run("fin", "pbounce", struct.pack("<i", n) + a.tobytes())
    recs = np.frombuffer(out, PB_OUT, n, 0)
    off = n * PB_OUT.itemsize
    att = np.frombuffer(out, "<f4", 3 * n, off).reshape(n, 3); off += att.nbytes
    cs = np.frombuffer(out, "<f4", 6 * n, off).reshape(n, 6)
    np.savez_compressed(os.path.join(GOLD, "pbounce.npz"), cases=a, out=recs, attenuation=att, coord=cs)
    print(f"pbounce: {int(recs['ret'].sum())}/{n} bounced")


LIGHTS_IN = np.dtype([("type", "<i4"), ("intensity", "<f4", 3), ("vec", "<f4", 3), ("p", "<f4", 3), ("shadow", "<f4")])
LIGHTS_OUT = np.dtype([("illum", "<f4", 3), ("dir", "<f4", 3), ("ncalls", "<i4"), ("log", "<f4", 7)])


def gen_lights():
    """AmbientLight / DirectLight::Illuminate and Light::Direction of all three lights (lights.h:28-57, 159)."""
    rng = np.random.default_rng(404)
    n = 192
    a = np.zeros(n, LIGHTS_IN)
    a["type"] = rng.integers(0, 3, n)
    a["intensity"] = rng.uniform(0.05, 3.0, (n, 3))
    a["vec"] = rng.normal(size=(n, 3)) * rng.choice([0.3, 1.0, 20.0], (n, 1))       # SetDirection normalises; positions as they are
    a["p"] = rng.uniform(-15, 15, (n, 3))
    a["shadow"] = rng.choice([0.0, 1.0], n)
    res = {"cases": a}
    for model in ("fin", "p13"):
        out = np.frombuffer(run(model, "lights", struct.pack("<i", n) + a.tobytes()), LIGHTS_OUT)
        assert len(out) == n
        res[f"out_{model}"] = out
    assert res["out_fin"].tobytes() == res["out_p13"].tobytes()
    np.savez_compressed(os.path.join(GOLD, "lights.npz"), cases=a, out=res["out_fin"])
    print("lights:", n, "cases")


def gen_refimages():
    """Images the reference itself holds, copied as DATA fixtures (reference-held outputs, not source):
    RayTracingProj13/prj13_boxzbuff.png -- the z image (RenderImage::SaveZImage) of the P13 Cornell scene,
    RayTracingProj5/RayTracingProj5/prj5_zbuff.png -- the z image of the scene RayTracingProj6/scene.xml describes."""
    import shutil
    for src, dst in ((("RayTracingProj13", "prj13_boxzbuff.png"), "ref_prj13_boxzbuff.png"),
                     (("RayTracingProj5", "RayTracingProj5", "prj5_zbuff.png"), "ref_prj5_zbuff.png")):
        shutil.copyfile(os.path.join(REF, *src), os.path.join(GOLD, dst))
        os.chmod(os.path.join(GOLD, dst), 0o644)
    print("reference-held z images copied")


# ---- the reference's main.cpp (oracle/_ref/ref_main_harness_*, see ref_main_harness.cpp) -------------------------
MAIN_HARNESS = {m: os.path.join(ROOT, "oracle", "_ref", f"ref_main_harness_{m}") for m in ("fin", "p13")}
SCENE_DIR = {"fin": os.path.join(REF, "RayTracingFinal", "RayTracingFinal", "data"),
             "p13": os.path.join(REF, "RayTracingProj13", "RayTracingProj13")}
MAINHIT = np.dtype([("hit", "<i4"), ("z", "<f4"), ("p", "<f4", 3), ("N", "<f4", 3), ("front", "<i4"), ("node", "<i4")])


def run_main(model, cmd, payload):
    """the reference's own program text, scene.xml loaded by its own LoadScene; cwd = the scene's directory (OBJ names
    are cwd-relative, xmlload.cpp:205)"""
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(payload)
        subprocess.run([MAIN_HARNESS[model], cmd, "scene.xml", fin, fout], check=True, cwd=SCENE_DIR[model])
        with open(fout, "rb") as f:
            return f.read()


def photons_payload(pos, dirn, power):
    a = np.concatenate([pos, dirn, power], axis=1).astype(np.float32)
    return struct.pack("<i", len(a)) + a.tobytes()


def take_balanced(out):
    n = struct.unpack_from("<i", out, 0)[0]
    if n == 0:
        return np.zeros(0, PHOTON), 4
    return np.frombuffer(out, PHOTON, n + 1, 4).copy(), 4 + (n + 1) * 24


def decode_photons(ph):
    """Photon::GetDirection / GetPower (cyPhotonMap.h:58,158-180) in numpy: only to turn a dump into AddPhoton inputs"""
    dx = ph["dir_x"].astype(np.float32) / np.float32(0x7FFF)
    dy = ph["dir_y"].astype(np.float32) / np.float32(0x7FFF)
    dz = np.sqrt(np.maximum(0.0, 1.0 - dx.astype(np.float64) ** 2 - dy.astype(np.float64) ** 2)).astype(np.float32)
    dz = np.where(ph["plane_and_dirz"] & 8, -dz, dz)
    power = ph["color"].astype(np.float32) / np.float32(255.0) * ph["power"][:, None]
    return np.stack([dx, dy, dz], 1), power


def main_photon_set():
    """<= 50 000 photons for the Shade / RenderPixel fixtures, made by the reference's own PhotonTracing (1 000 000 stored,
    seed 11: the density of the real map) and thinned so that BOTH regimes of EstimateIrradiance<400>(radius 1) occur: the
    full density (about 600 within the radius: the k-th distance decides) inside a ball around the floor between the
    spheres, sparse elsewhere."""
    out = run_main("fin", "photontrace", struct.pack("<Iiiii", 11, 1000000, 8, 0, 0))
    consumed = struct.unpack_from("<i", out, 0)[0]
    off = 4 + max(consumed, 0) * 4
    attempts, n = struct.unpack_from("<qi", out, off)
    ph = np.frombuffer(out, PHOTON, n, off + 12)
    pos = ph["position"]
    dense = np.linalg.norm(pos - np.array([0, -6, 0], np.float32), axis=1) < 8.5
    rng = np.random.default_rng(12)
    keep_d = rng.permutation(np.flatnonzero(dense))[:44000]
    keep_s = rng.permutation(np.flatnonzero(~dense))[:6000]
    sel = np.sort(np.concatenate([keep_d, keep_s]))
    d, pw = decode_photons(ph[sel])
    # the powers were scaled by 4*pi/1000000 (the real map's): the thinned set keeps them
    return pos[sel].astype(np.float32), d, pw.astype(np.float32), dict(attempts=attempts, stored=n, dense=len(keep_d), sparse=len(keep_s))


def main_rays(rng, cam_pos, n_cam, n_box):
    """camera-like rays (from the camera position through the view) and rays between points inside the box"""
    tgt = np.stack([rng.uniform(-17, 17, n_cam), rng.uniform(-10, 16, n_cam), rng.uniform(-1, 25, n_cam)], 1)
    d = tgt - cam_pos
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    cam = np.concatenate([np.broadcast_to(cam_pos, (n_cam, 3)), d], 1)
    o = np.stack([rng.uniform(-14, 14, n_box), rng.uniform(-28, 14, n_box), rng.uniform(0.5, 23.5, n_box)], 1)
    d = rng.normal(size=(n_box, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    aim = rng.random(n_box) < 0.5                                   # towards the spheres / the teapot
    tg = np.array([[8, -6, 4], [-8, -6, 4], [2, 5, 3], [0, -6, 0]], float)[rng.integers(0, 4, n_box)] + rng.normal(scale=2.5, size=(n_box, 3))
    da = tg - o
    da /= np.linalg.norm(da, axis=1, keepdims=True)
    d[aim] = da[aim]
    return np.concatenate([cam, np.concatenate([o, d], 1)]).astype(np.float32)


def gen_main_shade(model):
    """TraceNode + MtlBlinn::Shade + GenLight::Shadow of the reference's main.cpp on the Cornell scene.xml"""
    rng = np.random.default_rng(201 if model == "fin" else 202)
    top = 4 if model == "fin" else 6                     # BOUNCE of the snapshot (FIN/main.cpp:25, P13/main.cpp:25)
    cam_pos = np.array([0, -60, 12], float)
    if model == "fin":
        pos, d, pw, info = main_photon_set()
        counts = {4: (1100, 100), 3: (300, 700), 2: (200, 600), 1: (150, 500), 0: (100, 400)}
    else:
        pos, d, pw, info = np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), {}
        counts = {6: (900, 100), 5: (200, 300), 4: (200, 400), 3: (200, 400), 2: (150, 400), 1: (150, 350), 0: (100, 300)}
    rays, bounce = [], []
    for b, (nc, nb) in counts.items():
        r = main_rays(rng, cam_pos, nc, nb)
        rays.append(r)
        bounce.append(np.full(len(r), b, np.int32))
    rays, bounce = np.concatenate(rays), np.concatenate(bounce)
    n = len(rays)
    assert n >= 4096 and bounce.max() == top
    cases = np.zeros(n, np.dtype([("ray", "<f4", 6), ("bounce", "<i4")]))
    cases["ray"], cases["bounce"] = rays, bounce
    # first run: hit points only, to aim the Shadow rays from real surface points (where the biases matter)
    out = run_main(model, "shade", photons_payload(pos, d, pw) + struct.pack("<i", n) + cases.tobytes() + struct.pack("<i", 0))
    bal, off = take_balanced(out)
    hits = np.frombuffer(out, MAINHIT, n, off)
    hp = hits["p"][hits["hit"] == 1]
    light = np.array([0, 0, 22], np.float32)
    ns = 2048
    src = hp[rng.integers(0, len(hp), ns)]
    sh = np.zeros((ns, 7), np.float32)
    sh[:, :3] = src
    sh[:, 3:6] = light - src                             # PointLight::Illuminate: Shadow(Ray(p, position - p), 1)
    sh[:, 6] = 1.0
    k = ns // 4                                          # the rest: other targets, lengths and t_max (DirectLight: BIGFLOAT)
    sh[:k, 3:6] = rng.normal(size=(k, 3)) * rng.choice([0.3, 1.0, 20.0], size=(k, 1))
    sh[:k, 6] = rng.choice([1.0, 1.0e30], k)
    sh[k:2 * k, :3] = np.stack([rng.uniform(-14, 14, k), rng.uniform(-28, 14, k), rng.uniform(0.5, 23.5, k)], 1)
    sh[k:2 * k, 3:6] = light - sh[k:2 * k, :3]
    out = run_main(model, "shade", photons_payload(pos, d, pw) + struct.pack("<i", n) + cases.tobytes() + struct.pack("<i", ns) + sh.tobytes())
    bal, off = take_balanced(out)
    hits = np.frombuffer(out, MAINHIT, n, off).copy(); off += n * MAINHIT.itemsize
    rgb = np.frombuffer(out, np.float32, 3 * n, off).reshape(n, 3).copy(); off += 12 * n
    shadow = np.frombuffer(out, np.float32, ns, off).copy()
    assert off + 4 * ns == len(out)
    np.savez_compressed(os.path.join(GOLD, f"main_shade_{model}.npz"), photons=bal, rays=rays, bounce=bounce, hits=hits, rgb=rgb,
                        shadow_rays=sh, shadow=shadow)
    print(f"main_shade_{model}: {n} cases, {int(hits['hit'].sum())} hits, {len(bal)} photon records {info}, "
          f"shadow: {ns} rays, {int((shadow == 0).sum())} occluded, nodes hit {sorted(set(hits['node'][hits['hit'] == 1]))}")


def gen_main_pixels(model):
    """RenderPixel of the reference's main.cpp: RGB8 + z + sample-count byte of pixel segments"""
    segs = []
    if model == "fin":
        pos, d, pw, info = main_photon_set()
        frames = [((160, 120), [(0, 160 * 120)]),                                            # a whole small frame
                  ((800, 600), [(440 * 800, 800), (455 * 800 + 200, 400), (470 * 800, 800), (520 * 800 + 300, 300)])]
    else:
        pos = d = pw = np.zeros((0, 3), np.float32)
        # rows >= 327 only: P13's RenderPixel skips (and never counts) the rows above (P13/main.cpp:219)
        frames = [((800, 600), [(330 * 800, 800), (440 * 800 + 100, 600), (470 * 800 + 150, 500), (560 * 800, 400)])]
    res = {}
    bal = None
    for fi, ((w, h), seg) in enumerate(frames):
        a = np.array(seg, np.int32)
        out = run_main(model, "pixels", photons_payload(pos, d, pw) + struct.pack("<iii", w, h, len(a)) + a.tobytes())
        bal, off = take_balanced(out)
        ow, oh = struct.unpack_from("<ii", out, off); off += 8
        assert (ow, oh) == (w, h)
        rgb, z, cnt = [], [], []
        for start, count in seg:
            rgb.append(np.frombuffer(out, np.uint8, 3 * count, off).reshape(count, 3)); off += 3 * count
            z.append(np.frombuffer(out, np.float32, count, off)); off += 4 * count
            cnt.append(np.frombuffer(out, np.uint8, count, off)); off += count
        assert off + 8 == len(out)                       # (+ the seconds RenderPixel took: bench.py's figure, not a fixture)
        res[f"f{fi}_size"] = np.array([w, h], np.int32)
        res[f"f{fi}_segments"] = a
        res[f"f{fi}_rgb"], res[f"f{fi}_z"], res[f"f{fi}_count"] = np.concatenate(rgb), np.concatenate(z), np.concatenate(cnt)
        print(f"main_pixels_{model}: frame {w}x{h}, {sum(c for _, c in seg)} px, {int((res[f'f{fi}_count'] == 255).sum())} took the second batch, "
              f"{int((res[f'f{fi}_z'] > 1e29).sum())} background")
    if model == "fin":
        # the same thinned photon set as main_shade_fin.npz (main_photon_set is deterministic): stored there only
        shade = np.load(os.path.join(GOLD, "main_shade_fin.npz"))
        assert shade["photons"].tobytes() == bal.tobytes(), "run gen_main_shade('fin') first"
    np.savez_compressed(os.path.join(GOLD, f"main_pixels_{model}.npz"), n_frames=len(frames), **res)


def gen_main_photontrace():
    """generatePhotonMap's loop with the reference's RandomPhoton / TraceNode / PhotonTracing (mode 0) and CausticTracing
    (mode 1), rand() captured (RayTracingFinal: the snapshot the product's photon pass follows)"""
    res = {}
    for tag, mode, count, bounce, seed in (("photon", 0, 3000, 8, 7), ("caustic", 1, 6000, 5, 8)):
        out = run_main("fin", "photontrace", struct.pack("<Iiiii", seed, count, bounce, mode, 600000))
        consumed = struct.unpack_from("<i", out, 0)[0]
        assert consumed > 0, "capture too small"
        raw = np.frombuffer(out, np.int32, consumed, 4).copy()
        off = 4 + 4 * consumed
        attempts, n = struct.unpack_from("<qi", out, off); off += 12
        ph = np.frombuffer(out, PHOTON, n, off).copy()
        assert off + 24 * n == len(out)
        res.update({f"{tag}_seed": seed, f"{tag}_count": count, f"{tag}_bounce": bounce, f"{tag}_raw": raw, f"{tag}_attempts": attempts, f"{tag}_photons": ph})
        print(f"main_photontrace {tag}: {n} photons stored, {attempts} attempts, {consumed} rand() calls")
    np.savez_compressed(os.path.join(GOLD, "main_photontrace.npz"), **res)


OLD_HARNESS = {m: os.path.join(ROOT, "oracle", "_ref", f"ref_main_harness_{m}") for m in ("p12", "p6", "p3")}
OLD_SCENE = {"p12": (os.path.join(REF, "RayTracingProj13", "RayTracingProj13"), "scene.xml"),      # the Cornell file (RayTracingProj12's own loader reads it)
             "p6": (os.path.join(REF, "RayTracingProj6", "RayTracingProj6"), "scene.xml"),
             "p3": (os.path.join(REF, "RayTracingProj3", "RayTracingProj3"), "input2.xml")}


def gen_main_shade_old(model):
    """TraceNode + MtlBlinn::Shade + GenLight::Shadow of the main.cpp of RayTracingProj12 (live GI: its rand() draws captured
    per case), RayTracingProj6 and RayTracingProj3 (see oracle/ref_main_harness_old.cpp)"""
    rng = np.random.default_rng({"p12": 301, "p6": 302, "p3": 303}[model])
    if model == "p3":
        # RayTracingProj3's Shade takes V = camera.pos - p (main.cpp:152-190): every ray starts at the camera of input2.xml
        cam_pos = np.array([0, -60, 12], float)
        tgt = np.stack([rng.uniform(-20, 20, 1500), rng.uniform(-5, 20, 1500), rng.uniform(-2, 26, 1500)], 1)
        dd = tgt - cam_pos
        dd /= np.linalg.norm(dd, axis=1, keepdims=True)
        rays = np.concatenate([np.broadcast_to(cam_pos, (1500, 3)), dd], 1).astype(np.float32)
        bounce = np.zeros(1500, np.int32)
        light = np.array([0, 0, 22], np.float32)
    else:
        cam_pos = np.array([0, -60, 12], float)
        counts = ({8: (40, 10), 5: (60, 40), 4: (100, 100), 3: (150, 150), 2: (150, 150), 1: (150, 150), 0: (100, 100)} if model == "p12"
                  else {5: (500, 100), 4: (100, 150), 3: (100, 150), 2: (100, 150), 1: (100, 150), 0: (100, 100)})
        rr, bb = [], []
        for b, (nc, nb) in counts.items():
            r = main_rays(rng, cam_pos, nc, nb)
            rr.append(r)
            bb.append(np.full(len(r), b, np.int32))
        rays, bounce = np.concatenate(rr), np.concatenate(bb)
        light = np.array([0, 0, 22], np.float32)
    n = len(rays)
    cases = np.zeros(n, np.dtype([("ray", "<f4", 6), ("bounce", "<i4"), ("seed", "<u4")]))
    cases["ray"], cases["bounce"], cases["seed"] = rays, bounce, rng.integers(1, 2 ** 31, n)
    cwd, xml = OLD_SCENE[model]
    capture = 400000 if model == "p12" else 16

    def run_old(sh):
        with tempfile.TemporaryDirectory() as td:
            fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
            with open(fin, "wb") as f:
                f.write(struct.pack("<i", n) + cases.tobytes() + struct.pack("<i", len(sh)) + sh.tobytes() + struct.pack("<i", capture))
            subprocess.run([OLD_HARNESS[model], "shade", xml, fin, fout], check=True, cwd=cwd)
            return open(fout, "rb").read()
    out = run_old(np.zeros((0, 7), np.float32))
    hits = np.frombuffer(out, MAINHIT, n, 0)
    hp = hits["p"][hits["hit"] == 1]
    ns = 1024
    src = hp[rng.integers(0, len(hp), ns)]
    sh = np.zeros((ns, 7), np.float32)
    sh[:, :3], sh[:, 3:6], sh[:, 6] = src, light - src, 1.0
    k = ns // 4
    sh[:k, 3:6] = rng.normal(size=(k, 3)) * rng.choice([0.3, 1.0, 20.0], size=(k, 1))
    sh[:k, 6] = rng.choice([1.0, 1.0e30], k)
    out = run_old(sh)
    off = 0
    hits = np.frombuffer(out, MAINHIT, n, off).copy(); off += n * MAINHIT.itemsize
    rgb = np.frombuffer(out, np.float32, 3 * n, off).reshape(n, 3).copy(); off += 12 * n
    consumed = np.frombuffer(out, np.int32, n, off).copy(); off += 4 * n
    assert (consumed >= 0).all(), "rand() capture too small"
    raw = np.frombuffer(out, np.int32, int(consumed.sum()), off).copy(); off += 4 * int(consumed.sum())
    shadow = np.frombuffer(out, np.float32, ns, off).copy()
    assert off + 4 * ns == len(out)
    # the rand() values a case consumed are NOT stored: they are srand(seed)'s first `consumed` outputs, restated in
    # oracle/orc.py (glibc_rand) -- checked here against what the reference run really drew
    sys.path.insert(0, ROOT)
    from oracle import orc
    at = 0
    for i in np.flatnonzero(consumed > 0):
        assert (orc.glibc_rand(cases["seed"][i], consumed[i]) == raw[at:at + consumed[i]]).all(), i
        at += consumed[i]
    np.savez_compressed(os.path.join(GOLD, f"main_shade_{model}.npz"), rays=rays, bounce=bounce, seed=cases["seed"], hits=hits, rgb=rgb, consumed=consumed,
                        shadow_rays=sh, shadow=shadow)
    print(f"main_shade_{model}: {n} cases, {int(hits['hit'].sum())} hits, {int(consumed.sum())} rand() values kept (max {int(consumed.max())} per case), "
          f"shadow {ns} rays / {int((shadow == 0).sum())} occluded, nodes {sorted(set(hits['node'][hits['hit'] == 1]))}")


def gen_main_pixels_old(model):
    """the WHOLE frame of BASELINE config C1 (RayTracingProj3, input2.xml, 640 x 480) / C2 (RayTracingProj6, scene.xml,
    800 x 600) from the snapshot's own RenderPixel (oracle/ref_main_harness_old.cpp `pixels`): Color24 image and z buffer"""
    cwd, xml = OLD_SCENE[model]
    size = {"p3": (640, 480), "p6": (800, 600)}[model]
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(struct.pack("<ii", *size))                      # RayTracingProj3's input2.xml says 800 x 600: BASELINE config C1 is 640 x 480
        subprocess.run([OLD_HARNESS[model], "pixels", xml, fin, fout], check=True, cwd=cwd)
        out = open(fout, "rb").read()
    w, h = struct.unpack("<ii", out[:8])
    assert (w, h) == size and len(out) == 8 + 7 * w * h
    rgb = np.frombuffer(out, np.uint8, w * h * 3, 8).reshape(h, w, 3).copy()
    z = np.frombuffer(out, np.float32, w * h, 8 + 3 * w * h).reshape(h, w).copy()
    np.savez_compressed(os.path.join(GOLD, f"main_pixels_{model}.npz"), rgb=rgb, z=z)
    print(f"main_pixels_{model}: {w} x {h}, {int((z < 1e29).sum())} pixels hit, mean level {rgb.mean():.1f}")


def gen_main():
    for m in ("fin", "p13"):
        if not os.path.exists(MAIN_HARNESS[m]):
            sys.exit(f"{MAIN_HARNESS[m]} missing: run `make -C oracle refmain` in the build container first")
    gen_main_shade("fin")
    gen_main_shade("p13")
    gen_main_pixels("fin")
    gen_main_pixels("p13")
    gen_main_photontrace()
    for m in ("p12", "p6", "p3"):
        gen_main_shade_old(m)
    gen_main_pixels_old("p6")
    gen_main_pixels_old("p3")


def main():
    os.makedirs(GOLD, exist_ok=True)
    for m in ("fin", "p13"):
        if not os.path.exists(HARNESS[m]):
            sys.exit(f"{HARNESS[m]} missing: run `make -C oracle` in the build container first")
    gen_prims("fin")
    gen_prims("p13")
    gen_box()
    gen_mesh("fin")
    gen_mesh("p13")
    gen_mesh_mtl()
    gen_node()
    gen_misc()
    gen_texture()
    gen_photon("k400", 12000, 400, 1.0, 384, 106)
    gen_photon("k50", 3001, 50, 1.5, 256, 107)
    gen_photon("k8", 64, 8, 4.0, 64, 108)
    gen_photon_dat()
    gen_zimage()
    gen_illum()
    gen_pbounce()
    gen_lights()
    gen_refimages()
    gen_main()
    print("fixtures written to", GOLD)


if __name__ == "__main__":
    if len(sys.argv) > 1:          # regenerate only the named fixtures, e.g. `gen_golden.py mesh_mtl`
        for name in sys.argv[1:]:
            fn, _, arg = name.partition(":")      # e.g. main_shade:fin
            globals()["gen_" + fn](*([arg] if arg else []))
    else:
        main()
