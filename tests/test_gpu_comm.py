"""The multi-GPU exchange step behind the C ABI (smr_comm_*, smr_gather_tiles): inputs sharded over contexts, tiles gathered on
the root, root composes — bit-identical to rendering everything on one context.  A one-GPU box covers the code paths with two
contexts on the same device (local comm) and a one-rank RCCL communicator; the two-device test runs where two GPUs are visible."""
import numpy as np
import pytest

from tests import scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from smelter_amd import hip as h
    return h


def _scene(ctx, hip, iw, ih, W, H, n):
    layouts, res = scenes.cfg2_scene(iw, ih, W, H, n)
    planes = [scenes.test_input(i, iw, ih, noise_seed=300 + i) for i in range(n)]
    return layouts, res, planes


def _sharded_render(hip, ctxs, layouts, res, planes, iw, ih, W, H, comm, root=0):
    """input i -> ctxs[i % len(ctxs)]: smr_ingest_resample_batch per shard, smr_gather_tiles, smr_render_layouts on the root."""
    from dataclasses import replace
    from smelter_amd.dist import rust_round
    n, world = len(planes), len(ctxs)
    geom = {}
    for L in layouts:
        if L.type == 0:
            geom[L.source_index] = (max(rust_round(L.width), 1), max(rust_round(L.height), 1), tuple(L.crop))
    src, dst = [None] * n, [None] * n
    for r, c in enumerate(ctxs):
        mine = [i for i in range(n) if i % world == r]
        frames = [c.frame(hip.FRAME_PLANAR_YUV420, iw, ih, list(planes[i])) for i in mine]
        tiles = [c.surface(geom[i][0], geom[i][1]) for i in mine]
        kinds = c.ingest_resample_batch(frames, [geom[i][2] for i in mine], tiles)
        assert all(k > 0 for k in kinds)
        for i, t in zip(mine, tiles):
            src[i] = t
    for i in range(n):
        dst[i] = src[i] if i % world == root else ctxs[root].surface(geom[i][0], geom[i][1])
    comm.gather(root, [i % world for i in range(n)], src, dst)
    for t in dst:
        t.opaque = True
    root_layouts = [replace(L, crop=(0.0, 0.0, float(geom[L.source_index][0]), float(geom[L.source_index][1]))) if L.type == 0 else L for L in layouts]
    out = ctxs[root].frame(hip.FRAME_PLANAR_YUV420, W, H)
    ctxs[root].render_layouts(root_layouts, dst, W, H, out=out)
    for c in ctxs:
        c.sync()
    return out.download()


def _single_render(hip, ctx, layouts, planes, iw, ih, W, H):
    frames = [ctx.frame(hip.FRAME_PLANAR_YUV420, iw, ih, list(p)) for p in planes]
    out = ctx.frame(hip.FRAME_PLANAR_YUV420, W, H)
    ctx.render_layouts(layouts, frames, W, H, out=out)
    return out.download()


def test_local_comm_two_contexts_on_one_device(hip):
    iw, ih, W, H, n = 320, 180, 480, 272, 4
    a, b = hip.Context(0), hip.Context(0)
    layouts, res, planes = _scene(a, hip, iw, ih, W, H, n)
    comm = hip.Comm.local([a, b])
    assert comm.world == 2
    want = _single_render(hip, a, layouts, planes, iw, ih, W, H)
    for root in (0, 1):
        got = _sharded_render(hip, [a, b], layouts, res, planes, iw, ih, W, H, comm, root=root)
        for g, w_ in zip(got, want):
            assert (g == w_).all(), root
    comm.close()
    b.close()
    a.close()


def test_local_comm_two_devices(hip):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one process driving both: what a single renderer thread would hold)")
    iw, ih, W, H, n = 320, 180, 480, 272, 4
    a, b = hip.Context(0), hip.Context(1)
    layouts, res, planes = _scene(a, hip, iw, ih, W, H, n)
    comm = hip.Comm.local([a, b])
    want = _single_render(hip, a, layouts, planes, iw, ih, W, H)
    got = _sharded_render(hip, [a, b], layouts, res, planes, iw, ih, W, H, comm)
    for g, w_ in zip(got, want):
        assert (g == w_).all()
    comm.close()
    b.close()
    a.close()


def test_rank_comm_of_one_and_the_sharded_driver_over_it(hip):
    """smr_comm_create_rank with world 1 (librccl is opened, a communicator is initialised, the gather has nothing to move), under
    the same ShardedCompositor bench.py --gpus N drives."""
    import torch
    from smelter_amd import dist as smr_dist
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    c = hip.Context(0, stream=side.cuda_stream)
    comm = hip.Comm.rank(c, 1, 0, hip.Comm.unique_id())
    assert comm.world == 1
    iw, ih, W, H, n = 320, 180, 480, 272, 4
    layouts, res, planes = _scene(c, hip, iw, ih, W, H, n)
    want = _single_render(hip, c, layouts, planes, iw, ih, W, H)
    frames = {i: c.frame(hip.FRAME_PLANAR_YUV420, iw, ih, list(planes[i])) for i in range(n)}
    plan = smr_dist.ShardPlan(n_inputs=n, world=1)
    sharded = smr_dist.ShardedCompositor(c, hip, plan, 0, layouts, res, list(range(n)), None, torch, None, comm=comm)
    outs = [c.frame(hip.FRAME_PLANAR_YUV420, W, H) for _ in range(2)]
    sharded.step(frames, outs[0])
    sharded.step_pipelined(frames, outs[1])
    sharded.flush()
    c.sync()
    for o in outs:
        for g, w_ in zip(o.download(), want):
            assert (g == w_).all()
    comm.close()
    c.close()
    torch.cuda.set_stream(torch.cuda.default_stream())


def test_rank_comm_across_processes(hip):
    """The RCCL transport at world 2: `torchrun --nproc-per-node 2 tests/rank_gather_worker.py` — smr_comm_create_rank on every rank, tiles
    gathered by smr_gather_tiles (ncclSend / ncclRecv on the context streams), the gathered tiles and the composed frame byte for byte what one
    context renders alone.  Needs two GPUs; the worker carries its own 60 s watchdog so a stuck exchange fails instead of hanging."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one process per GPU over RCCL)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29617",
           os.path.join(root, "tests", "rank_gather_worker.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert "== one context" in p.stdout


def test_sharded_pipeline_with_two_ranks_on_one_device_over_an_animated_scene(hip):
    """The N > 1 driver on hardware as far as one GPU allows: TWO ShardedCompositors (rank 0 = root, rank 1), one context each on the same
    device, over smr_comm_create_local — the full `step_pipelined` protocol (ingest k, compose k - 1, gather k posted; two tile sets, two
    output frames) for 36 frames of configs[3]'s geometry (8 x 4K YUV420 inputs -> 4K, tiles 1280 x 720) while the layout list moves every
    frame.  Every output equals, byte for byte, what ONE context renders from the same frames and that frame's layouts.
    The per-input independence the sharding relies on: smelter-render/src/state/render_loop.rs:24-41, transformations/layout.rs:250-275."""
    import torch
    from dataclasses import replace
    from smelter_amd import dist as smr_dist
    iw, ih, W, H, n, frames_n = 3840, 2160, 3840, 2160, 8, 36
    a, b = hip.Context(0), hip.Context(0)
    ctxs = [a, b]
    base, res = scenes.cfg2_scene(iw, ih, W, H, n)  # Tiles of 8 inputs on 4K: a 3 x 3 grid of 1280 x 720 tiles
    rng = np.random.default_rng(77)
    planes = [scenes.random_yuv420(iw, ih, rng) if i % 2 else scenes.test_input(i, iw, ih, noise_seed=500 + i) for i in range(n)]
    frames = [[c.frame(hip.FRAME_PLANAR_YUV420, iw, ih, list(planes[i])) for i in range(n)] for c in ctxs]

    def layouts_at(k):  # the grid drifts: every tile moves (integer and fractional offsets, the size stays), two tiles swap depth order
        out = []
        for L in base:
            if L.type == 0:
                dx = 0.25 * k * (1 + L.source_index % 3) - 3.0 * (L.source_index % 2)
                dy = (k % 7) - 0.5 * (L.source_index % 4)
                L = replace(L, left=L.left + dx, top=L.top + dy)
            out.append(L)
        if k % 5 == 4:
            out[-1], out[-2] = out[-2], out[-1]
        return out
    comm = hip.Comm.local(ctxs)
    ranks = smr_dist.LocalRanks(comm)
    plan = smr_dist.ShardPlan(n_inputs=n, world=2)
    sc = [smr_dist.ShardedCompositor(ctxs[r], hip, plan, r, layouts_at(0), res, list(range(n)), None, torch, None, comm=ranks.view(r)) for r in range(2)]
    outs = [a.frame(hip.FRAME_PLANAR_YUV420, W, H) for _ in range(frames_n)]
    for k in range(frames_n):
        for r in ranks.order(plan.root):
            sc[r].set_layouts(layouts_at(k))
            sc[r].step_pipelined({i: frames[r][i] for i in range(n)}, outs[k] if r == plan.root else None)
    for r in ranks.order(plan.root):
        sc[r].flush()
    for c in ctxs:
        c.sync()
    ref = a.frame(hip.FRAME_PLANAR_YUV420, W, H)
    for k in range(frames_n):
        a.render_layouts(layouts_at(k), frames[0], W, H, out=ref)
        a.sync()
        for g, w_ in zip(outs[k].download(), ref.download()):
            assert np.array_equal(g, w_), f"frame {k}"
    comm.close()
    b.close()
    a.close()
