"""gloo tests (CPU, world sizes 2, 3 and 8) of the tile sharding + gather used for N>1 GPUs.  The packed layout
the torch helpers build here is the one k_resolve writes on the GPU (rt_render_tiles_packed_device) and
k_unpack_tiles reads (rt_tiles_unpack_device); tests/test_gpu_parity.py checks the HIP side against them."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_folder_amd import dist as rtd

W, H = 100, 37          # deliberately not a multiple of the 32x8 tile


def _frame(W=W, H=H):
    y, x = np.mgrid[0:H, 0:W]
    rgb = np.stack([x % 251, y % 241, (x * 7 + y * 13) % 256], 2).astype(np.uint8)
    z = (x * 0.25 + y * 100.0).astype(np.float32)
    cnt = ((x + y) % 2 * 255).astype(np.uint8)
    return torch.from_numpy(rgb), torch.from_numpy(z), torch.from_numpy(cnt)


def _worker(rank, world, port, q, W=W, H=H):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rgb, z, cnt = _frame(W, H)
    tx, ty, n = rtd.tile_grid(W, H)
    # this rank "renders" only its own tiles: everything else stays zero
    own = torch.zeros((ty * 8, tx * 32), dtype=torch.bool)
    for t in range(rank, n, world):
        own[(t // tx) * 8:(t // tx) * 8 + 8, (t % tx) * 32:(t % tx) * 32 + 32] = True
    own = own[:H, :W]
    r2, z2, c2 = rgb * own[..., None], z * own, cnt * own
    fr, fz, fc = rtd.gather_frame(r2, z2, c2, rank, world)
    ok = bool((fr == rgb).all() and (fz == z).all() and (fc == cnt).all())
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,size", [(2, (W, H)), (3, (W, H)), (8, (1920, 1080))])
def test_gather_frame_gloo(world, size):
    """world 8 at the BASELINE frame size: 8100 tiles of 32x8, 1013 per rank (the last four ranks one short)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q) + tuple(size)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(r for r, _ in res) == list(range(world)) and all(ok for _, ok in res)


def test_pack_unpack_roundtrip_single_process():
    rgb, z, cnt = _frame()
    world = 4
    packs = torch.stack([rtd.pack_own_tiles(rgb, z, cnt, r, world) for r in range(world)])
    fr, fz, fc = rtd.unpack_gathered(packs, W, H, world)
    assert (fr == rgb).all() and (fz == z).all() and (fc == cnt).all()
    tx, ty, n = rtd.tile_grid(1920, 1080)
    assert (tx, ty, n) == (60, 135, 8100)
    assert sum(len(rtd.tiles_of_rank(r, 8, n)) for r in range(8)) == n


def test_packed_sizes_match_the_c_abi():
    """rt_tiles_packed_size (host code of the C ABI) agrees with the torch packing used in the gloo tests"""
    from raytracing_folder_amd import capi
    for (w, h), world in (((100, 37), 3), ((1920, 1080), 8), ((64, 48), 5)):
        tx, ty, n = rtd.tile_grid(w, h)
        for rank in range(world):
            nbytes, k = capi.tiles_packed_size(w, h, capi.TileRange(32, 8, rank, world))
            assert k == len(rtd.tiles_of_rank(rank, world, n)) and nbytes == k * 32 * 8 * rtd.BYTES_PER_PIXEL
            assert k <= (n + world - 1) // world
