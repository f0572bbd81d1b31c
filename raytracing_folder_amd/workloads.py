"""Workload inputs of the render path, as the product ships them (bench.py, __graft_entry__.smoke() and the
tests all build their scenes here):

  * the Cornell box of the reference's photon-mapping snapshots (BASELINE config C4): data/cornell.xml +
    data/teapot_tri.obj, loaded through the product's own XML/OBJ loader; data/cornell_gi.xml is the variant
    RayTracingProj12's main() loads (config C3, live path-traced GI);
  * the stand-in for the absent christmas_balls.obj: 128 tessellated spheres
    (102 402 triangles with the ground quad) in two meshes, one of them mirrors, under a PNG sky that is both
    environment and background -- written as OBJ + PNG + XML and loaded the same way.
"""
import os
import tempfile

import numpy as np

from . import capi

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
CORNELL_XML = os.path.join(DATA, "cornell.xml")
CORNELL_GI_XML = os.path.join(DATA, "cornell_gi.xml")


def load_cornell(width=None, height=None):
    """(scene, camera) of the Cornell box; width/height override the file's 800 x 600"""
    s = capi.Scene()
    s.load_xml(CORNELL_XML)
    cam = s.camera()
    if width:
        cam.width, cam.height = int(width), int(height)
    return s, cam


def load_cornell_gi(width=None, height=None):
    """(scene, camera) of BASELINE config C3: the scene RayTracingProj12's main() loads (its scene-2.xml: glass teapot, glossy
    sphere, point light 0.5 -- that snapshot's PointLight has no fall-off, include/lights.h:86-88), for shade model P12"""
    s = capi.Scene()
    s.load_xml(CORNELL_GI_XML)
    cam = s.camera()
    if width:
        cam.width, cam.height = int(width), int(height)
    return s, cam


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


def sky_image(width=256, height=128):
    """a smooth sky: zenith blue -> pale horizon -> brown ground, with a warm spot (a PNG environment stands in for
    the reference's absent HDRI; TexturedColor::SampleEnvironment maps directions onto it, scene.h:426-432)"""
    y, x = np.mgrid[0:height, 0:width].astype(np.float64)
    t = y / (height - 1)
    top, hor, gnd = np.array([0.25, 0.45, 0.9]), np.array([0.85, 0.9, 1.0]), np.array([0.35, 0.3, 0.25])
    up = np.clip(t / 0.55, 0, 1)[..., None]
    img = np.where((t < 0.55)[..., None], top * (1 - up) + hor * up, gnd)
    sun = np.exp(-(((x - 0.7 * width) / (0.05 * width)) ** 2 + ((y - 0.3 * height) / (0.08 * height)) ** 2))[..., None]
    img = np.clip(img + sun * np.array([1.0, 0.9, 0.6]), 0, 1)
    return (img * 255).astype(np.uint8)


def make_balls_scene(width, height, n_balls=64, directory=None):
    """C5 stand-in: two meshes of n_balls tessellated spheres each (the first with the ground quad), matte and
    mirror materials, ambient + point light, PNG sky as environment AND background.  Returns (scene, camera)."""
    d = directory or tempfile.mkdtemp(prefix="rt_balls_")
    va, fa = balls_scene(n_balls=n_balls, seed=1)               # n balls + the ground quad
    vb, fb = balls_scene(n_balls=n_balls, seed=2)
    vb, fb = vb[:-4], fb[:-2]                                   # the second group without a second ground
    write_obj(os.path.join(d, "balls_a.obj"), va, fa)
    write_obj(os.path.join(d, "balls_b.obj"), vb, fb)
    capi.image_write_png(os.path.join(d, "sky.png"), sky_image())
    with open(os.path.join(d, "scene.xml"), "w") as f:
        f.write(f"""<xml><scene>
  <background value="1" texture="sky.png"/><environment value="1" texture="sky.png"/>
  <object type="obj" name="balls_a.obj" material="matte"/>
  <object type="obj" name="balls_b.obj" material="mirror"/>
  <material type="blinn" name="matte"><diffuse r="0.8" g="0.5" b="0.3"/><specular value="0.4"/><glossiness value="30"/></material>
  <material type="blinn" name="mirror"><diffuse value="0.1"/><specular value="0.9"/><glossiness value="80"/><reflection value="0.8"/></material>
  <light type="ambient" name="amb"><intensity value="0.2"/></light>
  <light type="point" name="sun"><intensity value="1800"/><position x="10" y="-30" z="40"/></light>
</scene><camera><position x="0" y="-52" z="20"/><target x="0" y="-4" z="3"/><up x="0" y="0" z="1"/>
  <fov value="35"/><width value="{int(width)}"/><height value="{int(height)}"/></camera></xml>""")
    s = capi.Scene()
    s.load_xml(os.path.join(d, "scene.xml"))
    return s, s.camera()
