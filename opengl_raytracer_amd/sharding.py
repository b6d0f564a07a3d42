"""Multi-GPU sharding: one process per GPU, scene replicated, no data-path collective
during rendering, one gather of RGBA8 rows (or whole frames) to rank 0 over
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" on CPU).

Two shardings, both of independent units:
  * image tiles  -- rank r renders row tiles r, r+N, r+2N, ... of one frame
                    (interleaved, so rays that die early are spread over ranks);
  * frame batch  -- rank r renders views r, r+N, ... of a batch of camera views.

The renderer is whatever object has the RayTracer surface (set_row_tiles, row_map,
render, ...).  How its pixels reach a torch tensor: on the GPU the
tensor is bound as the context's framebuffer (no copy), elsewhere pixels are copied.
"""
import numpy as np
import torch
import torch.distributed as dist


def tile_rows_of_rank(height, tile_rows, rank, world):
    """Frame rows of rank `rank`: tiles rank, rank+world, ... of `tile_rows` rows each."""
    ntiles = (height + tile_rows - 1) // tile_rows
    rows = []
    for t in range(rank, ntiles, world):
        rows.extend(range(t * tile_rows, min((t + 1) * tile_rows, height)))
    return np.asarray(rows, dtype=np.int64)


def choose_tile_rows(height, world, preferred=16):
    """The largest tile height <= preferred that gives every rank the same number of rows
    (so the gather needs no padding); falls back to `preferred`."""
    for tr in range(preferred, 0, -1):
        if height % (tr * world) == 0:
            return tr
    return preferred


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class TileShardedFrame:
    """One frame cut into interleaved row tiles over the ranks of the default process group."""

    def __init__(self, renderer, width, height, device, tile_rows=None, bind_device_framebuffer=True):
        self.r = renderer
        self.width, self.height = int(width), int(height)
        self.device = torch.device(device)
        self.rank, self.world = _world()
        self.tile_rows = tile_rows or choose_tile_rows(self.height, self.world)
        self.r.set_row_tiles(self.tile_rows, self.rank, self.world)
        self.rows = np.asarray(self.r.row_map(), dtype=np.int64)
        assert np.array_equal(self.rows, tile_rows_of_rank(self.height, self.tile_rows, self.rank, self.world))
        # every rank's slab is padded to the largest one so that the gather is uniform
        self.max_rows = max(tile_rows_of_rank(self.height, self.tile_rows, q, self.world).shape[0]
                            for q in range(self.world))
        self.local = torch.zeros((self.max_rows, self.width, 4), dtype=torch.uint8, device=self.device)
        self.bound = bool(bind_device_framebuffer and self.device.type == "cuda")
        if self.bound:
            self.r.bind_stream(torch.cuda.current_stream(self.device).cuda_stream)
            self.r.bind_framebuffer(self.local.data_ptr())
        if self.rank == 0:
            self.gathered = [torch.zeros_like(self.local) for _ in range(self.world)]
            # frame row -> position in the concatenated slabs
            perm = np.zeros(self.height, dtype=np.int64)
            for q in range(self.world):
                rows_q = tile_rows_of_rank(self.height, self.tile_rows, q, self.world)
                perm[rows_q] = q * self.max_rows + np.arange(rows_q.shape[0])
            self.perm = torch.from_numpy(perm).to(self.device)
            self.frame = torch.zeros((self.height, self.width, 4), dtype=torch.uint8, device=self.device)

    def render(self, passes):
        """Renders this rank's tiles and gathers the frame on rank 0 (returns it there, else None)."""
        self.r.render(passes)
        if not self.bound:
            px = torch.from_numpy(np.ascontiguousarray(self.r.read_pixels()))
            self.local[: px.shape[0]].copy_(px)
        if self.world == 1:
            return self.local[: self.height]  # one rank owns every tile, rows already in order
        dist.gather(self.local, self.gathered if self.rank == 0 else None, dst=0)
        if self.rank != 0:
            return None
        slabs = torch.cat(self.gathered, dim=0)
        torch.index_select(slabs, 0, self.perm, out=self.frame)
        return self.frame


class FrameBatch:
    """A batch of camera views, view k rendered by rank k % world; frames gathered to rank 0."""

    def __init__(self, renderer, width, height, cameras, device, bind_device_framebuffer=True):
        self.r = renderer
        self.width, self.height = int(width), int(height)
        self.device = torch.device(device)
        self.rank, self.world = _world()
        self.cameras = list(cameras)
        self.mine = list(range(self.rank, len(self.cameras), self.world))
        self.max_views = (len(self.cameras) + self.world - 1) // self.world
        self.local = torch.zeros((self.max_views, self.height, self.width, 4), dtype=torch.uint8, device=self.device)
        self.bound = bool(bind_device_framebuffer and self.device.type == "cuda")
        if self.bound:
            self.r.bind_stream(torch.cuda.current_stream(self.device).cuda_stream)
        if self.rank == 0:
            self.gathered = [torch.zeros_like(self.local) for _ in range(self.world)]

    def render(self, passes):
        for j, k in enumerate(self.mine):
            self.r.set_camera(*self.cameras[k])
            if self.bound:
                self.r.bind_framebuffer(self.local[j].data_ptr())
                self.r.render(passes)
            else:
                self.r.render(passes)
                self.local[j].copy_(torch.from_numpy(np.ascontiguousarray(self.r.read_pixels())))
        if self.bound:
            self.r.bind_framebuffer(0)
        if self.world == 1:
            return self.local[: len(self.cameras)]
        dist.gather(self.local, self.gathered if self.rank == 0 else None, dst=0)
        if self.rank != 0:
            return None
        # view k sits at slab k % world, slot k // world
        stacked = torch.stack(self.gathered, dim=1).reshape(-1, self.height, self.width, 4)
        return stacked[: len(self.cameras)]
