"""Shared scene helpers for the tests (TEST INFRASTRUCTURE: may use oracle/)."""
import os

import numpy as np

from oracle import orc
from raytracing_folder_amd import capi, workloads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CORNELL = workloads.CORNELL_XML


load_cornell = workloads.load_cornell
load_cornell_gi = workloads.load_cornell_gi
oracle_scene = orc.scene_from_export        # orc.Scene over the very arrays the product exports
oracle_camera = orc.camera_from
oracle_params = orc.params_from


def identity_node(parent=-1, obj=capi.OBJ_NONE, material=-1, mesh=-1, scale=1.0, pos=(0, 0, 0)):
    n = np.zeros(1, capi.NODE)
    n["tm"][0, [0, 4, 8]] = scale
    n["itm"][0, [0, 4, 8]] = np.float32(1.0) / np.float32(scale)
    n["pos"] = pos
    n["parent"], n["obj_type"], n["material"], n["mesh"] = parent, obj, material, mesh
    return n


def camera_rays(cam, n, seed=0):
    """n primary rays of random (pixel, sample index) pairs, built by the oracle's RenderPixel
    restatement."""
    rng = np.random.default_rng(seed)
    oc = oracle_camera(cam)
    xs, ys, js = rng.integers(0, cam.width, n), rng.integers(0, cam.height, n), rng.integers(0, 8, n)
    return np.stack([orc.primary_ray(oc, x, y, j) for x, y, j in zip(xs, ys, js)])


def rebuild(export, materials=None, lights=None):
    """A new product scene from exported arrays, with materials / lights optionally replaced."""
    s = capi.Scene()
    s.set_nodes(export["nodes"])
    for i, m in enumerate(export["meshes"]):
        s.set_mesh(i, m["v"], m["f"], m["vn"], m["fn"], m["nodes"], m["elements"], m.get("vt"), m.get("ft"))
    s.set_materials(export["materials"] if materials is None else materials)
    s.set_lights(export["lights"] if lights is None else lights)
    return s


def build_shim_driver(tmp_path):
    """g++ build of tests/shim_driver.cpp (the reference's BeginRender / StopRender / saveImage contract in
    C++, on top of the product library); returns the executable's path"""
    import subprocess
    exe = os.path.join(str(tmp_path), "shim_driver")
    lib = os.path.join(ROOT, "raytracing_folder_amd", "lib")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "shim_driver.cpp"), "-L" + lib, "-lrt_mi355x", "-Wl,-rpath," + lib, "-lpthread"],
                   check=True, capture_output=True)
    return exe


def build_dist_driver(tmp_path):
    """g++ build of tests/dist_driver.cpp: the C++ multi-GPU host (rt_render_tiles_packed_device -> ncclAllGather ->
    rt_tiles_unpack_device) against the C ABI header and /opt/rocm/include/rccl; returns the executable's path"""
    import subprocess
    exe = os.path.join(str(tmp_path), "dist_driver")
    lib = os.path.join(ROOT, "raytracing_folder_amd", "lib")
    subprocess.run(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "dist_driver.cpp"), "-L" + lib, "-lrt_mi355x", "-Wl,-rpath," + lib,
                    "-L/opt/rocm/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"],
                   check=True, capture_output=True)
    return exe
