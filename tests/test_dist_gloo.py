"""world_size-2 gloo test (CPU) of the tile sharding + gather used for N>1 GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_folder_amd import dist as rtd

W, H = 100, 37          # deliberately not a multiple of the 32x8 tile


def _frame():
    y, x = np.mgrid[0:H, 0:W]
    rgb = np.stack([x % 251, y % 241, (x * 7 + y * 13) % 256], 2).astype(np.uint8)
    z = (x * 0.25 + y * 100.0).astype(np.float32)
    cnt = ((x + y) % 2 * 255).astype(np.uint8)
    return torch.from_numpy(rgb), torch.from_numpy(z), torch.from_numpy(cnt)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rgb, z, cnt = _frame()
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


@pytest.mark.parametrize("world", [2, 3])
def test_gather_frame_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
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
