"""The N > 1 path on CPU: world_size-2 (and 3) gloo runs of the shard plan + tile gather + root compose
plumbing (smelter_amd/dist.py).  The device kernels are replaced by deterministic stand-ins; what is
under test is exactly what differs from the single-GPU path: input ownership, tile geometry, the
point-to-point exchange and the root-side layout rewrite."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from smelter_amd import dist as smr_dist
from tests import scenes


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tile_pattern(k, dh, pitch):
    return ((np.arange(dh * pitch, dtype=np.int64).reshape(dh, pitch) * (k + 3) + k * 17) % 251).astype(np.uint8)


def _worker(rank, world, port, n_inputs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        layouts, res = scenes.cfg3_scene(480, 270, 960, 540, n_inputs)
        slots = [i for i, r in enumerate(res) if r == (480, 270)]
        plan = smr_dist.ShardPlan(n_inputs=n_inputs, world=world)
        seen = {}
        composed = []

        def ingest(k, frame, tile):
            assert plan.owner(k) == rank and frame[0] == f"frame{k}"
            tile.copy_(torch.from_numpy(_tile_pattern(k + 7 * frame[1], tile.shape[0], tile.shape[1])))

        def compose(tiles, out):  # `out` carries the frame number the root believes it is composing
            composed.append(out)
            for k, t in tiles.items():
                ok = bool((t.numpy() == _tile_pattern(k + 7 * out, t.shape[0], t.shape[1])).all())
                seen[k] = seen.get(k, True) and ok

        comp = smr_dist.ShardedCompositor(None, None, plan, rank, layouts, res, slots, "label", torch, dist, ingest_fn=ingest,
                                          compose_fn=compose, device="cpu")
        for step in range(3):
            comp.step({k: (f"frame{k}", step) for k in plan.inputs_of(rank)}, step)
        # pipelined: frame k's exchange overlaps frame k-1's compose; every frame must still be composed from its own tiles
        for step in range(3, 9):
            comp.step_pipelined({k: (f"frame{k}", step) for k in plan.inputs_of(rank)}, step)
        comp.flush()
        if rank == 0:
            assert composed == list(range(9)), composed
        # animated scene: every rank sets the frame's layout list before the frame; the root must compose frame k with the list
        # (and the non-input surface) of frame k although frame k + 1's were set before frame k's exchange completed
        from dataclasses import replace
        moved = {}
        first_tex = next(i for i, L in enumerate(layouts) if L.type == 0 and L.source_index in comp.input_of_slot)
        comp.compose_fn = lambda tiles, out: (compose(tiles, out), moved.__setitem__(out, (comp.root_layouts[first_tex].left, comp.label)))
        for step in range(9, 15):
            comp.set_layouts([replace(L, left=L.left + 0.25 * step) if L.type == 0 else L for L in layouts])
            comp.label = f"layer{step}"
            comp.step_pipelined({k: (f"frame{k}", step) for k in plan.inputs_of(rank)}, step)
        comp.flush()
        if rank == 0:
            assert composed == list(range(15)), composed
            base = layouts[first_tex].left
            assert moved == {s: (base + 0.25 * s, f"layer{s}") for s in range(9, 15)}, moved
        try:
            comp.set_layouts([replace(L, width=L.width * 2) if L.type == 0 else L for L in layouts])
            resized_ok = False
        except ValueError:
            resized_ok = True
        assert resized_ok
        if rank == 0:
            # root-side layouts sample whole tiles 1:1
            geom_ok = all(L.crop == (0.0, 0.0, float(comp.tile_geom[comp.input_of_slot[L.source_index]][0]),
                                     float(comp.tile_geom[comp.input_of_slot[L.source_index]][1]))
                          for L in comp.root_layouts if L.type == 0 and L.source_index in comp.input_of_slot)
            q.put((sorted(seen.items()), geom_ok, sorted(comp.tiles.keys())))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_inputs", [(2, 8), (3, 5)])
def test_sharded_gather_gloo(world, n_inputs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_inputs, q)) for r in range(world)]
    for p in procs:
        p.start()
    seen, geom_ok, keys = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert keys == list(range(n_inputs))
    assert seen == [(k, True) for k in range(n_inputs)]  # every tile (local and remote) arrived intact at the root
    assert geom_ok


def test_shard_plan():
    p = smr_dist.ShardPlan(8, 4)
    assert [p.inputs_of(r) for r in range(4)] == [[0, 4], [1, 5], [2, 6], [3, 7]]
    assert p.remote_inputs() == [1, 2, 3, 5, 6, 7]
    p1 = smr_dist.ShardPlan(8, 1)
    assert p1.inputs_of(0) == list(range(8)) and p1.remote_inputs() == []
    assert smr_dist.rust_round(2.5) == 3 and smr_dist.rust_round(1265.78) == 1266 and smr_dist.pitch_of(1266) == 5120
