"""Multi-GPU path: inputs sharded across ranks, one exchange step (tile gather to the root), root composes.

The reference has no multi-GPU support (single wgpu device); this is the MI355X-native
scaling axis named by BASELINE.json: per-input work (colour conversion + Lanczos to the on-screen size)
is independent per input (smelter-render/src/state/render_loop.rs:24-41, transformations/layout.rs:250-275),
so input i lives on GPU i % world; every frame each rank turns its inputs into dst-sized RGBA8 tiles and
sends them point-to-point to rank 0 (RCCL send/recv: each peer uses its own xGMI link to the root, so the
gather is per-link bound, not ring bound); rank 0 runs the single compose+output kernel.

The exchange logic is backend-agnostic (torch.distributed): `nccl` (= RCCL) on GPUs, `gloo` in the CPU tests.
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np


def rust_round(x: float) -> int:
    """f32::round — half away from zero (layout.rs:258-261)."""
    x = float(np.float32(x))
    return int(np.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)


@dataclass(frozen=True)
class ShardPlan:
    n_inputs: int
    world: int
    root: int = 0

    def owner(self, input_idx: int) -> int:
        return input_idx % self.world

    def inputs_of(self, rank: int) -> List[int]:
        return [i for i in range(self.n_inputs) if self.owner(i) == rank]

    def remote_inputs(self) -> List[int]:
        return [i for i in range(self.n_inputs) if self.owner(i) != self.root]


def pitch_of(width_px: int) -> int:
    return (width_px * 4 + 255) & ~255


def post_gather(dist, plan: ShardPlan, rank: int, tiles: Dict[int, "object"]):
    """Posts one exchange step without waiting: non-root ranks send their tiles to the root, the root receives every remote
    tile.  `tiles[i]` is the tile tensor of input i: filled on owner(i), receive buffer on the root.  Returns the work handles;
    `w.wait()` orders the current stream after completion."""
    ops = []
    if rank == plan.root:
        for i in plan.remote_inputs():
            if i in tiles:
                ops.append(dist.P2POp(dist.irecv, tiles[i], plan.owner(i), tag=i))
    else:
        for i in plan.inputs_of(rank):
            if i in tiles:
                ops.append(dist.P2POp(dist.isend, tiles[i], plan.root, tag=i))
    return dist.batch_isend_irecv(ops) if ops else []


def gather_tiles(dist, plan: ShardPlan, rank: int, tiles: Dict[int, "object"]):
    """One exchange step, posted and waited on (see post_gather)."""
    ops = []
    if rank == plan.root:
        for i in plan.remote_inputs():
            if i in tiles:
                ops.append(dist.P2POp(dist.irecv, tiles[i], plan.owner(i), tag=i))
    else:
        for i in plan.inputs_of(rank):
            if i in tiles:
                ops.append(dist.P2POp(dist.isend, tiles[i], plan.root, tag=i))
    if not ops:
        return []
    works = dist.batch_isend_irecv(ops)
    for w in works:
        w.wait()
    return works


class ShardedCompositor:
    """Per-frame driver of the sharded path for one output scene."""

    _FRAME_STATE = ("root_layouts", "root_packed", "label", "slot_of_input", "input_of_slot")  # what a frame in flight keeps

    def __init__(self, ctx, hip, plan: ShardPlan, rank: int, layouts, res, input_source_slot: Sequence[int], label_surface,
                 torch, dist, ingest_fn: Optional[Callable] = None, compose_fn: Optional[Callable] = None, device=None, comm=None):
        """comm: a hip.Comm (smr_comm_create_rank) — the exchange then goes through the C ABI (smr_gather_tiles: RCCL send / recv
        enqueued on the ctx stream, no torch.distributed in the data path); None: torch.distributed point-to-point (`dist`),
        which is also what the CPU tests run over gloo."""
        self.ctx, self.hip, self.plan, self.rank, self.dist, self.torch = ctx, hip, plan, rank, dist, torch
        self.comm = comm
        if comm is None and dist is not None and ctx is not None and torch.cuda.is_available() and plan.world > 1:
            # torch.distributed orders its send / recv against torch's CURRENT stream, the library enqueues on the ctx stream:
            # unless they are the same stream a tile could be sent before the ingest kernel wrote it, or composed before it
            # arrived.  (The C-ABI exchange, comm=..., runs on the ctx stream itself and has no such requirement.)
            if getattr(ctx, "stream_handle", None) != torch.cuda.current_stream().cuda_stream:
                raise ValueError("ShardedCompositor over torch.distributed needs the context on torch's current stream: create it with "
                                 "hip.Context(device, stream=s.cuda_stream) under torch.cuda.stream(s) / torch.cuda.set_stream(s) — or pass comm=")
        self.layouts = list(layouts)
        self.res = list(res)
        self.slot_of_input = list(input_source_slot)
        self.input_of_slot = {s: k for k, s in enumerate(self.slot_of_input)}
        self.label = label_surface
        device = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu")
        # geometry of every input's tile: the layout that samples it (resample_scaled_children, layout.rs:238-278)
        self.tile_geom: Dict[int, tuple] = {}
        for L in self.layouts:
            if L.type == 0 and L.source_index in self.input_of_slot:
                k = self.input_of_slot[L.source_index]
                self.tile_geom[k] = (max(rust_round(L.width), 1), max(rust_round(L.height), 1), tuple(L.crop))
        needed = plan.inputs_of(rank) if rank != plan.root else list(range(plan.n_inputs))
        # two sets of tiles: frame k+1 is resampled / received into one set while frame k's is still being sent / composed
        self.tile_sets = [{}, {}]
        self.surface_sets = [{}, {}]
        for par in range(2):
            for k in needed:
                if k not in self.tile_geom:
                    continue
                dw, dh, _ = self.tile_geom[k]
                t = torch.zeros((dh, pitch_of(dw)), dtype=torch.uint8, device=device)
                self.tile_sets[par][k] = t
                if ctx is not None:
                    s = ctx.wrap(t.data_ptr(), pitch_of(dw), dw, dh)
                    # tiles come out of the resampler with alpha == 255 (planar YUV / NV12 inputs): the compositor may use them
                    # as base layers (SMR_SOURCE_OPAQUE_SURFACE) exactly as it does with the raw frames on a single GPU
                    s.opaque = True
                    self.surface_sets[par][k] = s
        self.tiles, self.tile_surfaces = self.tile_sets[0], self.surface_sets[0]  # the set of the frame being worked on
        self.frame_no = 0
        self.pending = None  # (works, parity, out) of the frame whose exchange is in flight
        self.batched = ingest_fn is None and ctx is not None  # default device path: all local inputs in one launch
        self.ingest_fn = ingest_fn or self._ingest
        self.compose_fn = compose_fn or self._compose
        self.root_layouts, self.root_packed = None, None
        if rank == plan.root:
            self._rewrite_for_root(pack=True)

    def _rewrite_for_root(self, pack: bool):
        # root-side layout list: inputs are replaced by their (already resampled) tiles, crop = whole tile
        self.root_layouts = []
        for L in self.layouts:
            if L.type == 0 and L.source_index in self.input_of_slot:
                dw, dh, _ = self.tile_geom[self.input_of_slot[L.source_index]]
                L = replace(L, crop=(0.0, 0.0, float(dw), float(dh)))
            self.root_layouts.append(L)
        self.root_packed = self.hip.pack_layouts(self.root_layouts) if (pack and self.hip is not None) else None

    def set_layouts(self, layouts, input_source_slot: Optional[Sequence[int]] = None):
        """The layout list of the next frame of an animated scene (every rank evaluates the same scene at the same pts, so all
        ranks call this with the same list and no geometry travels).  Layouts may move, rotate, change masks or order; the size
        of the tile each input is resampled to must stay what the tile buffers were allocated for — true for Tiles / absolute
        position transitions, which move children without resizing them (tiles_component/interpolation.rs:17-64)."""
        layouts = list(layouts)
        if input_source_slot is not None:  # a scene update re-ordered the children: source slot s now shows another input
            self.slot_of_input = list(input_source_slot)
            self.input_of_slot = {s: k for k, s in enumerate(self.slot_of_input)}
        for L in layouts:
            if L.type == 0 and L.source_index in self.input_of_slot:
                k = self.input_of_slot[L.source_index]
                geom = (max(rust_round(L.width), 1), max(rust_round(L.height), 1), tuple(L.crop))
                if k not in self.tile_geom or self.tile_geom[k][:2] != geom[:2]:
                    raise ValueError(f"input {k}: tile size changed from {self.tile_geom.get(k, (None,))[:2]} to {geom[:2]}; "
                                     "build a new ShardedCompositor for a scene that resizes its inputs")
                self.tile_geom[k] = geom
        self.layouts = layouts
        if self.rank == self.plan.root:
            self._rewrite_for_root(pack=False)  # packed per frame by render_layouts

    # -- default device implementations
    def _ingest(self, k, frame, tile_tensor):
        dw, dh, crop = self.tile_geom[k]
        kind = self.ctx.ingest_resample(frame, crop, self.tile_surfaces[k])
        if kind == 0:
            raise RuntimeError("sharded path expects scaled inputs (direct 1:1 inputs need no resample shard)")

    def _compose(self, tiles, out):
        srcs = []
        for slot, r in enumerate(self.res):
            srcs.append(self.tile_surfaces[self.input_of_slot[slot]] if slot in self.input_of_slot else self.label)
        self.ctx.render_layouts(self.root_layouts, srcs, out.w, out.h, out=out, packed=self.root_packed)

    def _ingest_local(self, frames_row):
        mine = [k for k in self.plan.inputs_of(self.rank) if k in self.tile_geom]
        if self.batched:
            kinds = self.ctx.ingest_resample_batch([frames_row[k] for k in mine], [self.tile_geom[k][2] for k in mine],
                                                   [self.tile_surfaces[k] for k in mine])
            if any(kd == 0 for kd in kinds):
                raise RuntimeError("sharded path expects scaled inputs (direct 1:1 inputs need no resample shard)")
        else:
            for k in mine:
                self.ingest_fn(k, frames_row[k], self.tiles[k])

    def _gather_c_abi(self, par: int):
        """smr_gather_tiles on the ctx stream: tile k from rank owner(k) into the root's surface of the same set."""
        ks = sorted(self.tile_geom)
        owners = [self.plan.owner(k) for k in ks]
        surf = self.surface_sets[par]
        src = [surf.get(k) if self.plan.owner(k) == self.rank else None for k in ks]
        dst = [surf.get(k) if self.rank == self.plan.root else None for k in ks]
        self.comm.gather(self.plan.root, owners, src, dst)

    def step(self, frames_row: Dict[int, object], out):
        """One frame, start to finish: resample the local inputs, exchange, compose on the root."""
        self.flush()
        self._ingest_local(frames_row)
        if self.comm is not None:
            self._gather_c_abi(0 if self.tiles is self.tile_sets[0] else 1)
        else:
            gather_tiles(self.dist, self.plan, self.rank, self.tiles)
        if self.rank == self.plan.root:
            self.compose_fn(self.tiles, out)

    def step_pipelined(self, frames_row: Dict[int, object], out):
        """One frame per call, one frame of latency: frame k is resampled and its exchange posted, then frame k-1 (whose tiles
        have been travelling meanwhile) is composed — the xGMI transfer of one frame overlaps the kernels of its neighbours.
        The two tile sets alternate; call flush() after the last frame."""
        par = self.frame_no & 1
        self.tiles, self.tile_surfaces = self.tile_sets[par], self.surface_sets[par]
        self._ingest_local(frames_row)
        if self.comm is not None:
            # stream order on the root: ingest(k) . compose(k-1) . receive(k) — the frame composed now arrived during the previous
            # call; the senders' ingest(k) runs while the root composes
            prev, self.pending = self.pending, ([], par, out, {name: getattr(self, name) for name in self._FRAME_STATE})
            self.frame_no += 1
            self._finish(prev)
            self._gather_c_abi(par)
            return
        works = post_gather(self.dist, self.plan, self.rank, self.tiles)
        prev, self.pending = self.pending, (works, par, out, {name: getattr(self, name) for name in self._FRAME_STATE})
        self.frame_no += 1
        self._finish(prev)

    def _finish(self, pending):
        if pending is None:
            return
        works, par, out, state = pending
        for w in works:
            w.wait()  # the current stream now waits for this frame's sends / receives
        if self.rank == self.plan.root:
            # the frame is composed with the tile set, the layout list, the slot -> input map and the non-input surface (`label`)
            # it was prepared with (set_layouts / a new label may have moved on to the next frame)
            now = {name: getattr(self, name) for name in self._FRAME_STATE}
            self.tiles, self.tile_surfaces = self.tile_sets[par], self.surface_sets[par]
            self.__dict__.update(state)
            self.compose_fn(self.tiles, out)
            self.__dict__.update(now)
            nxt = self.frame_no & 1
            self.tiles, self.tile_surfaces = self.tile_sets[nxt], self.surface_sets[nxt]

    def flush(self):
        prev, self.pending = self.pending, None
        self._finish(prev)


class LocalRanks:
    """One process, `world` ShardedCompositors — one per context, i.e. what a single renderer thread holding several GPUs (or several
    contexts of one GPU) runs — over ONE local communicator (smr_comm_create_local: peer copies + events, smr_comm.hip).

    A rank-mode communicator is called by every rank with its own half of the gather; a local one moves everything in one call.  `view(rank)`
    is the `comm=` of rank's ShardedCompositor: it collects the ranks' halves and issues the single smr_gather_tiles when the last rank of a
    frame has posted.  Drive the compositors in lockstep, the root LAST (its call composes frame k - 1 and then posts frame k's gather):

        ranks = LocalRanks(hip.Comm.local(ctxs))
        sc = [ShardedCompositor(ctxs[r], hip, plan, r, ..., comm=ranks.view(r)) for r in range(world)]
        for frame in frames:
            for r in ranks.order(plan.root): sc[r].step_pipelined(frame[r], out if r == plan.root else None)
    """

    class _View:
        def __init__(self, parent, rank):
            self.parent, self.rank = parent, rank

        def gather(self, root, owners, src, dst):
            self.parent._post(self.rank, root, owners, src, dst)

    def __init__(self, comm):
        self.comm, self.world = comm, comm.world
        self.posted = {}

    def view(self, rank: int):
        return LocalRanks._View(self, rank)

    def order(self, root: int = 0):
        return [r for r in range(self.world) if r != root] + [root]

    def _post(self, rank, root, owners, src, dst):
        if rank in self.posted:
            raise RuntimeError(f"rank {rank} posted two gathers before the others posted one: drive the compositors in lockstep")
        self.posted[rank] = (list(src), list(dst))
        if len(self.posted) < self.world:
            return
        n = len(owners)
        src_all = [self.posted[owners[i]][0][i] for i in range(n)]
        dst_all = [self.posted[root][1][i] for i in range(n)]
        self.posted = {}
        self.comm.gather(root, list(owners), src_all, dst_all)
