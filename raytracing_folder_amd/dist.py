"""Multi-GPU sharding of a frame: interleaved tiles + one all-gather of the finished tiles.

The reference parallelises over pixels with one shared atomic counter (FIN/main.cpp:65-85; FIN =
/root/reference/RayTracingFinal/RayTracingFinal).  Here the image is cut into tile_w x tile_h
tiles numbered row-major; rank r of R renders tiles r, r+R, r+2R, ... (interleaved, so that the
expensive glass/mirror pixels spread over all ranks), every rank holds the whole scene and photon
map, and the only exchange is ONE all_gather of each rank's finished tiles -- 8 bytes per pixel
(RGB8 + float z + sample-count byte) -- over RCCL/xGMI (backend "nccl" on ROCm) or gloo on CPU.
torch is plumbing here: device buffers, the stream handle and torch.distributed.
"""
import torch
import torch.distributed as dist

from . import capi

BYTES_PER_PIXEL = 8          # 3 (Color24) + 4 (float z) + 1 (sample count)


def tile_grid(width, height, tile_w=32, tile_h=8):
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    return tiles_x, tiles_y, tiles_x * tiles_y


def tiles_of_rank(rank, world, n_tiles):
    return torch.arange(rank, n_tiles, world)


def _padded_tiles(rgb, z, cnt, tile_w, tile_h):
    """(H,W,3) u8, (H,W) f32, (H,W) u8  ->  (n_tiles, tile_h, tile_w, 8) u8 view-copy"""
    h, w = z.shape
    tx, ty, _ = tile_grid(w, h, tile_w, tile_h)
    ph, pw = ty * tile_h, tx * tile_w
    px = torch.zeros((ph, pw, BYTES_PER_PIXEL), dtype=torch.uint8, device=z.device)
    px[:h, :w, 0:3] = rgb
    px[:h, :w, 3:7] = z.contiguous().view(torch.uint8).reshape(h, w, 4)
    px[:h, :w, 7] = cnt
    return px.reshape(ty, tile_h, tx, tile_w, BYTES_PER_PIXEL).permute(0, 2, 1, 3, 4).reshape(tx * ty, tile_h, tile_w, BYTES_PER_PIXEL)


def pack_own_tiles(rgb, z, cnt, rank, world, tile_w=32, tile_h=8):
    """Compact buffer of this rank's tiles, padded to the per-rank maximum so that every rank
    contributes the same number of bytes to the all-gather."""
    tiles = _padded_tiles(rgb, z, cnt, tile_w, tile_h)
    n = tiles.shape[0]
    per_rank = (n + world - 1) // world
    out = torch.zeros((per_rank, tile_h, tile_w, BYTES_PER_PIXEL), dtype=torch.uint8, device=z.device)
    mine = tiles[rank::world]
    out[: mine.shape[0]] = mine
    return out.contiguous()


def unpack_gathered(gathered, width, height, world, tile_w=32, tile_h=8):
    """gathered: (world, per_rank, tile_h, tile_w, 8) u8 -> rgb (H,W,3) u8, z (H,W) f32, cnt (H,W) u8"""
    tx, ty, n = tile_grid(width, height, tile_w, tile_h)
    tiles = torch.zeros((n, tile_h, tile_w, BYTES_PER_PIXEL), dtype=torch.uint8, device=gathered.device)
    for r in range(world):
        k = len(range(r, n, world))
        tiles[r::world] = gathered[r, :k]
    px = tiles.reshape(ty, tx, tile_h, tile_w, BYTES_PER_PIXEL).permute(0, 2, 1, 3, 4).reshape(ty * tile_h, tx * tile_w, BYTES_PER_PIXEL)
    px = px[:height, :width]
    rgb = px[..., 0:3].contiguous()
    z = px[..., 3:7].contiguous().view(torch.float32).reshape(height, width)
    cnt = px[..., 7].contiguous()
    return rgb, z, cnt


def gather_frame(rgb, z, cnt, rank, world, tile_w=32, tile_h=8):
    """All-gather the finished tiles; every rank returns the complete frame."""
    h, w = z.shape
    if world == 1:
        return rgb, z, cnt
    mine = pack_own_tiles(rgb, z, cnt, rank, world, tile_w, tile_h)
    gathered = torch.empty((world * mine.shape[0],) + tuple(mine.shape[1:]), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(gathered, mine)          # rank r's tiles land in rows [r*per_rank, (r+1)*per_rank)
    return unpack_gathered(gathered.reshape((world,) + tuple(mine.shape)), w, h, world, tile_w, tile_h)


class ShardedRenderer:
    """One process per GPU: this rank's tiles are rendered straight into the buffer it contributes to the
    all-gather (rt_render_tiles_packed_device: k_resolve writes the 8-byte pixel records tile by tile), ONE
    all_gather_into_tensor moves them (RCCL over xGMI), and one small HIP kernel (rt_tiles_unpack_device)
    un-interleaves the gathered tiles into the RenderImage planes -- no Python-side packing in the step.
    `gather_ms` collects, per step, the time from the end of this rank's render to the finished frame."""

    def __init__(self, scene, cam, params, rank, world, device_index, tile_w=32, tile_h=8, host_gather=False):
        self.scene, self.cam, self.params = scene, cam, params
        self.host_gather = host_gather          # gather over CPU tensors (gloo rehearsal on one GPU)
        self.rank, self.world, self.device_index = rank, world, device_index
        self.tile_w, self.tile_h = tile_w, tile_h
        dev = torch.device("cuda", device_index)
        h, w = cam.height, cam.width
        self.rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
        self.z = torch.zeros((h, w), dtype=torch.float32, device=dev)
        self.cnt = torch.zeros((h, w), dtype=torch.uint8, device=dev)
        _, _, n = tile_grid(w, h, tile_w, tile_h)
        self.per_rank = (n + world - 1) // world
        # ranks whose share is one tile short leave the last slot zero: every rank contributes the same bytes
        self.packed = torch.zeros((self.per_rank, tile_h, tile_w, BYTES_PER_PIXEL), dtype=torch.uint8, device=dev)
        self.gathered = torch.zeros((world * self.per_rank, tile_h, tile_w, BYTES_PER_PIXEL), dtype=torch.uint8, device=dev)
        self.gather_ms = []
        # everything of a step -- render, all-gather, un-interleave -- is ordered on ONE explicit stream (torch's legacy
        # default stream has handle 0, which the C ABI reads as "the library's own stream": not ordered with ours)
        self.stream = torch.cuda.Stream(device=dev)

    def _stream(self):
        return self.stream.cuda_stream

    def render_own_tiles(self, want_stats=True, sync=True):
        """this rank's tiles into the image planes (single-GPU path, and the profiling leg of bench.py)"""
        tiles = capi.TileRange(self.tile_w, self.tile_h, self.rank, self.world)
        return self.scene.render_tiles_device(self.cam, self.params, tiles, self.device_index, self.rgb.data_ptr(),
                                              self.z.data_ptr(), self.cnt.data_ptr(), stream=self._stream(), sync=sync,
                                              want_stats=want_stats)

    def render_own_tiles_packed(self, want_stats=True, sync=True):
        tiles = capi.TileRange(self.tile_w, self.tile_h, self.rank, self.world)
        return self.scene.render_tiles_packed_device(self.cam, self.params, tiles, self.device_index, self.packed.data_ptr(),
                                                     self.packed.numel(), stream=self._stream(), sync=sync, want_stats=want_stats)

    def step(self, sync=True):
        """One frame.  sync=True: the call returns with the finished frame and this rank's statistics (the host waits for the
        render before it starts the exchange).  sync=False: everything -- render, all-gather, un-interleave -- is only ENQUEUED
        on this renderer's stream (the library orders a frame behind the one before it on the GPU, the collective follows in
        stream order); no statistics, no host round trip between frames: call finish() before reading the frame.  The first
        frame of a renderer should be a synchronous one (it sizes the queues from measurement; an asynchronous render
        without history takes worst-case queues)."""
        if self.world == 1:
            st = self.render_own_tiles(want_stats=sync, sync=sync)
            self.gather_ms.append(0.0)
            return (st if sync else None), (self.rgb, self.z, self.cnt)
        import time
        with torch.cuda.stream(self.stream):
            st = self.render_own_tiles_packed(want_stats=sync, sync=sync)       # sync: this rank's tiles are final
            t0 = time.perf_counter()
            if self.host_gather:
                self.stream.synchronize()
                mine = self.packed.cpu()
                gathered = torch.empty((self.world * self.per_rank,) + tuple(mine.shape[1:]), dtype=torch.uint8)
                dist.all_gather_into_tensor(gathered, mine)
                self.gathered.copy_(gathered)
            else:
                dist.all_gather_into_tensor(self.gathered, self.packed)     # rank r's tiles land in rows [r*per_rank, (r+1)*per_rank)
            capi.tiles_unpack_device(self.device_index, self._stream(), self.gathered.data_ptr(), self.world, self.per_rank,
                                     self.cam.width, self.cam.height, self.tile_w, self.tile_h,
                                     self.rgb.data_ptr(), self.z.data_ptr(), self.cnt.data_ptr())
            if sync:
                self.stream.synchronize()
        if sync:
            self.gather_ms.append((time.perf_counter() - t0) * 1e3)
        return (st if sync else None), (self.rgb, self.z, self.cnt)

    def finish(self):
        """wait for the frames enqueued with step(sync=False) and collect their verdict (raises if a queue overflowed)"""
        self.stream.synchronize()
        self.scene.render_check(self.device_index)
