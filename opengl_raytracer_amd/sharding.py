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

Frames in flight: a frame is a chain of some hundred dependent kernels, and a rank's share of
a frame leaves most of a GPU idle between them.  Both classes therefore accept a LIST of renderers
(contexts with the same scene): consecutive frames (or views) go to them in turn, each on a stream
of its own, so that the chains of consecutive frames overlap on the GPU.  A frame returned by
render() is complete once its stream has been waited for (`wait()` does that for all of them).
"""
import contextlib
import time

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


def _as_list(renderers):
    return list(renderers) if isinstance(renderers, (list, tuple)) else [renderers]


class StepMarks:
    """Three marks on the timeline of one render() call -- before the rank's share is rendered, after it, after
    the gather -- so that a step can be split into the share's time and the collective's.  On the GPU they are
    events on the stream the call works on; elsewhere, clock readings."""

    def __init__(self, device):
        self.cuda = torch.device(device).type == "cuda"
        self.device = device
        self.marks = []

    def mark(self):
        if self.cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            self.marks.append(ev)
        else:
            self.marks.append(time.perf_counter())

    def ms(self):
        """(share_ms, gather_ms) once the work has finished (the caller synchronises first)."""
        a, b, c = self.marks[:3]
        if self.cuda:
            return a.elapsed_time(b), b.elapsed_time(c)
        return (b - a) * 1e3, (c - b) * 1e3


class _Streams:
    """The streams of renderers that run on the GPU with bound framebuffers: a single renderer works on
    the stream that is current when the object is made; several get a stream of their own each (none of
    them HIP's default stream, which does not run beside the others)."""

    def __init__(self, renderers, device, bound):
        self.device, self.bound = device, bound
        self.streams = []
        if bound:
            cur = torch.cuda.current_stream(device)
            self.streams = [cur] if len(renderers) == 1 else [torch.cuda.Stream(device) for _ in renderers]
            for r, s in zip(renderers, self.streams):
                if s != cur:
                    s.wait_stream(cur)
                r.bind_stream(s.cuda_stream)

    def on(self, i):
        return torch.cuda.stream(self.streams[i]) if self.bound else contextlib.nullcontext()

    def follow(self, only=None):
        """Every renderer's stream (or that of renderer `only`) waits for the stream that is current (work the
        caller enqueued there on buffers the renderers are about to overwrite: the gather of the last batch,
        the caller's reads of a frame)."""
        if self.bound:
            cur = torch.cuda.current_stream(self.device)
            for s in (self.streams if only is None else [self.streams[only]]):
                if s != cur:
                    s.wait_stream(cur)

    def wait(self):
        """The stream that is current waits for every renderer's stream."""
        if self.bound:
            cur = torch.cuda.current_stream(self.device)
            for s in self.streams:
                if s != cur:
                    cur.wait_stream(s)


class TileShardedFrame:
    """One frame cut into interleaved row tiles over the ranks of the default process group.
    `renderer`: one renderer, or a list of them for that many frames in flight."""

    def __init__(self, renderer, width, height, device, tile_rows=None, bind_device_framebuffer=True,
                 gather_always=False):
        self.rs = _as_list(renderer)
        self.r = self.rs[0]
        self.width, self.height = int(width), int(height)
        self.device = torch.device(device)
        self.rank, self.world = _world()
        # (gather_always: a process group of ONE rank gathers too -- a rehearsal of the collective path)
        self.gather = self.world > 1 or (gather_always and dist.is_available() and dist.is_initialized())
        self.tile_rows = tile_rows or choose_tile_rows(self.height, self.world)
        for r in self.rs:
            r.set_row_tiles(self.tile_rows, self.rank, self.world)
        self.rows = np.asarray(self.r.row_map(), dtype=np.int64)
        assert np.array_equal(self.rows, tile_rows_of_rank(self.height, self.tile_rows, self.rank, self.world))
        # every rank's slab is padded to the largest one so that the gather is uniform
        self.max_rows = max(tile_rows_of_rank(self.height, self.tile_rows, q, self.world).shape[0]
                            for q in range(self.world))
        self.bound = bool(bind_device_framebuffer and self.device.type == "cuda")
        assert self.bound or len(self.rs) == 1, "frames in flight need device framebuffers"
        # per renderer: its slab, and on rank 0 the gathered slabs and the assembled frame
        self.locals = [torch.zeros((self.max_rows, self.width, 4), dtype=torch.uint8, device=self.device) for _ in self.rs]
        self.local = self.locals[0]
        if self.rank == 0:
            self.gathered = [[torch.zeros_like(self.local) for _ in range(self.world)] for _ in self.rs]
            # frame row -> position in the concatenated slabs
            perm = np.zeros(self.height, dtype=np.int64)
            for q in range(self.world):
                rows_q = tile_rows_of_rank(self.height, self.tile_rows, q, self.world)
                perm[rows_q] = q * self.max_rows + np.arange(rows_q.shape[0])
            self.perm = torch.from_numpy(perm).to(self.device)
            self.frames = [torch.zeros((self.height, self.width, 4), dtype=torch.uint8, device=self.device) for _ in self.rs]
        # (the streams are made after every tensor they will touch: a stream made here waits for the current
        # stream, on which the tensors above are being filled)
        self.streams = _Streams(self.rs, self.device, self.bound)
        if self.bound:
            for r, loc in zip(self.rs, self.locals):
                r.bind_framebuffer(loc.data_ptr())
        self.turn = 0

    def render(self, passes, marks=None):
        """Renders this rank's tiles and gathers the frame on rank 0 (returns it there, else None).
        With several renderers the frame is complete once wait() (or a device synchronize) has returned, and
        its tensor is written again len(renderers) calls later; what the caller has enqueued on the current
        stream by then (its reads of that frame) is waited for before the tensor is overwritten.
        marks: a StepMarks that receives the three marks of this call."""
        i = self.turn
        self.turn = (self.turn + 1) % len(self.rs)
        r, local = self.rs[i], self.locals[i]
        self.streams.follow(i)
        with self.streams.on(i):
            if marks:
                marks.mark()
            r.render(passes)
            if not self.bound:
                px = torch.from_numpy(np.ascontiguousarray(r.read_pixels()))
                local[: px.shape[0]].copy_(px)
            if marks:
                marks.mark()
            if not self.gather:
                if marks:
                    marks.mark()
                return local[: self.height]  # one rank owns every tile, rows already in order
            dist.gather(local, self.gathered[i] if self.rank == 0 else None, dst=0)
            if marks:
                marks.mark()
            if self.rank != 0:
                return None
            slabs = torch.cat(self.gathered[i], dim=0)
            torch.index_select(slabs, 0, self.perm, out=self.frames[i])
            return self.frames[i]

    def wait(self):
        self.streams.wait()


class FrameBatch:
    """A batch of camera views, view k rendered by rank k % world; frames gathered to rank 0."""

    def __init__(self, renderer, width, height, cameras, device, bind_device_framebuffer=True, gather_always=False):
        self.rs = _as_list(renderer)
        self.r = self.rs[0]
        self.width, self.height = int(width), int(height)
        self.device = torch.device(device)
        self.rank, self.world = _world()
        self.gather = self.world > 1 or (gather_always and dist.is_available() and dist.is_initialized())
        self.cameras = list(cameras)
        self.mine = list(range(self.rank, len(self.cameras), self.world))
        self.max_views = (len(self.cameras) + self.world - 1) // self.world
        self.local = torch.zeros((self.max_views, self.height, self.width, 4), dtype=torch.uint8, device=self.device)
        self.bound = bool(bind_device_framebuffer and self.device.type == "cuda")
        assert self.bound or len(self.rs) == 1, "views in flight need device framebuffers"
        self.streams = _Streams(self.rs, self.device, self.bound)
        if self.rank == 0:
            self.gathered = [torch.zeros_like(self.local) for _ in range(self.world)]

    def render(self, passes, marks=None):
        # this rank's views go to the renderers in turn (a renderer's next view waits for its last one only);
        # none of them starts before the gather of the last batch, which reads the tensor they write, is done
        if marks:
            marks.mark()
        self.streams.follow()
        for j, k in enumerate(self.mine):
            i = j % len(self.rs)
            r = self.rs[i]
            with self.streams.on(i):
                r.set_camera(*self.cameras[k])
                if self.bound:
                    r.bind_framebuffer(self.local[j].data_ptr())
                    r.render(passes)
                else:
                    r.render(passes)
                    self.local[j].copy_(torch.from_numpy(np.ascontiguousarray(r.read_pixels())))
        if self.bound:
            for r in self.rs:
                r.bind_framebuffer(0)
            self.streams.wait()
        if marks:
            marks.mark()
        if not self.gather:
            if marks:
                marks.mark()
            return self.local[: len(self.cameras)]
        dist.gather(self.local, self.gathered if self.rank == 0 else None, dst=0)
        if marks:
            marks.mark()
        if self.rank != 0:
            return None
        # view k sits at slab k % world, slot k // world
        stacked = torch.stack(self.gathered, dim=1).reshape(-1, self.height, self.width, 4)
        return stacked[: len(self.cameras)]

    def wait(self):
        self.streams.wait()
