"""raytracing_folder_amd -- MI355X-native render path for the Roia2529/RayTracing-folder scenes.

Only the hot path lives here: the C ABI library (csrc/, built into lib/librt_mi355x.so), its ctypes
binding (capi) and the tile-sharding / RCCL gather helper (dist).  Names follow the reference's
domain: scenes, nodes, meshes, photons, tiles.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
