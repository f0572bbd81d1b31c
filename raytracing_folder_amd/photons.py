"""Photon-map inputs for the render path: the reference's 24-byte photon record in numpy, and a
synthetic photon distribution for the Cornell box (stand-in for generatePhotonMap's output until
the photon pass itself runs on the GPU -- SURVEY.md section 8 row f1).

pack_photons follows PhotonMap::AddPhoton -> Photon::SetDirection / SetPower
(FIN/include/cyPhotonMap.h:139-156,184-192; FIN = /root/reference/RayTracingFinal/RayTracingFinal):
direction x,y as short(d*0x7FFF) (truncation), the sign of z in bit 3 of planeAndDirZ, power = max
channel, colour = Color24(c/power) = clamp(int(c/power*255)).
"""
import numpy as np

from .capi import PHOTON, photon_balance


def pack_photons(pos, direction, power):
    pos, direction, power = (np.ascontiguousarray(a, np.float32).reshape(-1, 3) for a in (pos, direction, power))
    out = np.zeros(len(pos), PHOTON)
    out["position"] = pos
    out["dir_x"] = (direction[:, 0] * np.float32(0x7FFF)).astype(np.int16)
    out["dir_y"] = (direction[:, 1] * np.float32(0x7FFF)).astype(np.int16)
    out["plane_and_dirz"] = np.where(direction[:, 2] > 0, 0, 0x8).astype(np.uint8)
    pw = power.max(axis=1)              # power = r; if (power < g) power = g; if (power < b) power = b
    out["power"] = pw
    with np.errstate(divide="ignore", invalid="ignore"):
        c = (power / pw[:, None]).astype(np.float32) * np.float32(255)
    out["color"] = np.clip(np.nan_to_num(c, nan=0.0).astype(np.int32), 0, 255).astype(np.uint8)
    return out


def synth_cornell_photons(n, seed=20171203, flux=100.5):
    """n photons on the five walls of the reference's Cornell box (floor z=0, ceiling z=24, back
    y=20, left x=-15, right x=15; the open side faces the camera at y=-60).  Density falls off with
    the distance from the light at (0,0,22) roughly like a first diffuse bounce would; directions
    arrive from the hemisphere above each wall; colours take the wall tint.  Powers are scaled the
    way generatePhotonMap does, by 4*pi/n (FIN/main.cpp:396).  Returns the UNBALANCED 1-based
    array (index 0 unused)."""
    rng = np.random.default_rng(seed)
    wall = rng.choice(5, size=n, p=[0.30, 0.16, 0.22, 0.16, 0.16])
    u, v = rng.random(n), rng.random(n)
    # concentrate towards the box centre (where the light is) with a mild power law
    cu = 0.5 + (u - 0.5) * np.abs(2 * u - 1) ** 0.35
    cv = 0.5 + (v - 0.5) * np.abs(2 * v - 1) ** 0.35
    pos = np.zeros((n, 3))
    nrm = np.zeros((n, 3))
    tint = np.ones((n, 3))
    x = -15 + 30 * cu
    y = -30 + 50 * cv
    z = 24 * cv
    yb = -30 + 50 * cu
    m = wall == 0; pos[m] = np.stack([x[m], y[m], np.zeros(m.sum())], 1); nrm[m] = (0, 0, 1)
    m = wall == 1; pos[m] = np.stack([x[m], y[m], np.full(m.sum(), 24.0)], 1); nrm[m] = (0, 0, -1)
    m = wall == 2; pos[m] = np.stack([x[m], np.full(m.sum(), 20.0), z[m]], 1); nrm[m] = (0, -1, 0)
    m = wall == 3; pos[m] = np.stack([np.full(m.sum(), -15.0), yb[m], z[m]], 1); nrm[m] = (1, 0, 0); tint[m] = (1.0, 0.5, 0.5)
    m = wall == 4; pos[m] = np.stack([np.full(m.sum(), 15.0), yb[m], z[m]], 1); nrm[m] = (-1, 0, 0); tint[m] = (0.5, 0.5, 1.0)
    # incoming direction: cosine-weighted around -normal
    r1, r2 = rng.random(n), rng.random(n)
    phi = 2 * np.pi * r1
    st, ct = np.sqrt(r2), np.sqrt(1 - r2)
    a = np.where(np.abs(nrm[:, [0]]) > 0.5, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    t1 = np.cross(nrm, a)
    t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(nrm, t1)
    d = -(nrm * ct[:, None] + t1 * (st * np.cos(phi))[:, None] + t2 * (st * np.sin(phi))[:, None])
    power = tint * flux * rng.uniform(0.3, 1.0, (n, 1)) * (4 * np.pi / n)
    packed = pack_photons(pos, d, power)
    return np.concatenate([np.zeros(1, PHOTON), packed])


def synth_cornell_photon_map(n, seed=20171203):
    """Balanced (heap-ordered kd-tree) photon map ready for rt_scene_set_photons."""
    return photon_balance(synth_cornell_photons(n, seed))
