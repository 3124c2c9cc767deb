"""Worker of tests/test_gpu_comm.py::test_rank_comm_across_processes: one process per GPU (torchrun), smr_comm_create_rank over RCCL, every
rank resamples its shard of the inputs, smr_gather_tiles moves the tiles to rank 0, rank 0 composes and compares — tiles AND frame — byte
for byte with the same scene rendered on its own context alone.  Exit code 0 = identical; anything else fails the test.  A watchdog ends a
stuck exchange after 60 s (rc 3) instead of hanging the suite."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    beat = [time.monotonic()]

    def watchdog():
        while True:
            time.sleep(0.5)
            if time.monotonic() - beat[0] > 60.0:
                print(f"[rank {rank}] watchdog: no progress for 60 s", file=sys.stderr, flush=True)
                os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    from smelter_amd import dist as smr_dist
    from smelter_amd import hip
    from tests import scenes
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    ctx = hip.Context(local, stream=side.cuda_stream)
    uid = torch.zeros(hip.COMM_ID_BYTES, dtype=torch.uint8, device="cuda")
    if rank == 0:
        uid = torch.tensor(list(hip.Comm.unique_id()), dtype=torch.uint8, device="cuda")
    dist.broadcast(uid, src=0)
    beat[0] = time.monotonic()
    comm = hip.Comm.rank(ctx, world, rank, bytes(uid.cpu().tolist()))
    beat[0] = time.monotonic()
    iw, ih, W, H, n = 320, 180, 480, 272, 4
    layouts, res = scenes.cfg2_scene(iw, ih, W, H, n)
    planes = [scenes.test_input(i, iw, ih, noise_seed=300 + i) for i in range(n)]
    plan = smr_dist.ShardPlan(n_inputs=n, world=world)
    frames = {i: ctx.frame(hip.FRAME_PLANAR_YUV420, iw, ih, list(planes[i])) for i in plan.inputs_of(rank)}
    sharded = smr_dist.ShardedCompositor(ctx, hip, plan, rank, layouts, res, list(range(n)), None, torch, dist, comm=comm)
    outs = [ctx.frame(hip.FRAME_PLANAR_YUV420, W, H) for _ in range(2)] if rank == 0 else [None, None]
    sharded.step(frames, outs[0])
    beat[0] = time.monotonic()
    sharded.step_pipelined(frames, outs[1])
    sharded.flush()
    ctx.sync()
    beat[0] = time.monotonic()
    rc = 0
    if rank == 0:
        alone = hip.Context(local)
        all_frames = [alone.frame(hip.FRAME_PLANAR_YUV420, iw, ih, list(p)) for p in planes]
        want = alone.frame(hip.FRAME_PLANAR_YUV420, W, H)
        alone.render_layouts(layouts, all_frames, W, H, out=want)
        alone.sync()
        for o in outs:
            for g, w_ in zip(o.download(), want.download()):
                if not np.array_equal(g, w_):
                    rc = 1
        # the gathered tiles themselves against the same tiles resampled here
        from smelter_amd.dist import rust_round
        geom = {L.source_index: (max(rust_round(L.width), 1), max(rust_round(L.height), 1), tuple(L.crop)) for L in layouts if L.type == 0}
        for i in range(n):
            t = alone.surface(geom[i][0], geom[i][1])
            alone.ingest_resample(all_frames[i], geom[i][2], t)
            alone.sync()
            got = sharded.surface_sets[0].get(i)  # (the first step's tile set: what arrived over RCCL for the remote inputs)
            if got is None or not np.array_equal(got.download(), t.download()):
                rc = 1
        print(f"[rank 0] sharded over {world} ranks {'==' if rc == 0 else '!='} one context", flush=True)
        alone.close()
    flag = torch.tensor([rc], dtype=torch.int32, device="cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    comm.close()
    ctx.close()
    dist.destroy_process_group()
    sys.exit(int(flag.item()))


if __name__ == "__main__":
    main()
