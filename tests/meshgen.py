"""Procedural meshes for the large-scene parity cases (stand-in for the absent christmas_balls.obj,
SURVEY.md section 8 config C5): tessellated spheres scattered over a ground quad."""
import numpy as np


def uv_sphere(nu, nv, center, radius):
    """nu x nv quads -> 2*nu*(nv-1) triangles (poles are fans)"""
    v = [[0, 0, 1]]
    for j in range(1, nv):
        th = np.pi * j / nv
        for i in range(nu):
            ph = 2 * np.pi * i / nu
            v.append([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)])
    v.append([0, 0, -1])
    v = np.array(v) * radius + np.array(center)
    f = []
    for i in range(nu):
        f.append([0, 1 + i, 1 + (i + 1) % nu])
    for j in range(nv - 2):
        a, b = 1 + j * nu, 1 + (j + 1) * nu
        for i in range(nu):
            i2 = (i + 1) % nu
            f.append([a + i, b + i, b + i2])
            f.append([a + i, b + i2, a + i2])
    last = len(v) - 1
    a = 1 + (nv - 2) * nu
    for i in range(nu):
        f.append([last, a + (i + 1) % nu, a + i])
    return v, np.array(f)


def balls_scene(n_balls=128, nu=20, nv=21, seed=1):
    """n_balls spheres of 2*nu*(nv-1) = 800 triangles each (102 400 for the defaults) + a ground quad"""
    rng = np.random.default_rng(seed)
    vs, fs, off = [], [], 0
    for _ in range(n_balls):
        c = [rng.uniform(-14, 14), rng.uniform(-25, 18), rng.uniform(1.0, 8.0)]
        v, f = uv_sphere(nu, nv, c, rng.uniform(0.6, 1.6))
        vs.append(v); fs.append(f + off); off += len(v)
    g = np.array([[-20, -30, 0], [20, -30, 0], [20, 25, 0], [-20, 25, 0]], float)
    vs.append(g); fs.append(np.array([[0, 1, 2], [0, 2, 3]]) + off)
    return np.concatenate(vs).astype(np.float32), np.concatenate(fs).astype(np.uint32)


def write_obj(path, v, f):
    with open(path, "w") as o:
        for p in v:
            o.write("v %.9g %.9g %.9g\n" % tuple(p))
        for t in f:
            o.write("f %d %d %d\n" % (t[0] + 1, t[1] + 1, t[2] + 1))
