"""world_size-2 check of the multi-GPU layer on CPU (gloo): sharding partitions the clips like the
reference does and the single end-of-run gather delivers every rank's clip to rank 0."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tooncrafter_amd.dist import gather_clips, init, shard_indices


def test_shard_indices_partition():
    for n in (0, 1, 3, 8, 17):
        for world in (1, 2, 4, 8):
            got = [shard_indices(n, world, r) for r in range(world)]
            flat = [i for g in got for i in g]
            assert flat == list(range(n))
            assert max(len(g) for g in got) - min(len(g) for g in got) <= 1
            # reference behaviour (inference.py:314-320): equal contiguous slices, remainder dropped
            ref = [shard_indices(n, world, r, drop_remainder=True) for r in range(world)]
            per = n // world
            assert all(g == list(range(per * r, per * (r + 1))) for r, g in enumerate(ref))


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    r, w = init(backend="gloo")
    assert (r, w) == (rank, world)
    mine = shard_indices(5, world, rank)
    video = torch.full((1, 3, 4, 8, 8), float(rank + 1)) + torch.arange(8.0)       # rank-specific content
    # row f4: the writer-side gather moves uint8 frames (converted on every rank before the collective)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from emu_ops import EmuOps
    from tooncrafter_amd import ops, output
    ops.set_backend(EmuOps(round_bf16=True))
    clip = torch.full((1, 3, 2, 4, 4), -1.0 + 0.5 * rank)
    frames = output.gather_frames(clip, dst=0)
    if rank == 0:
        assert [tuple(f.shape) for f in frames] == [(1, 2, 4, 4, 3)] * world and frames[0].dtype == torch.uint8
        assert [int(f.flatten()[0]) for f in frames] == [int(((-1.0 + 0.5 * r + 1) / 2) * 255) for r in range(world)]
    else:
        assert frames is None
    got = gather_clips(video, dst=0)
    if rank == 0:
        ok = len(got) == world and all(torch.equal(g, torch.full((1, 3, 4, 8, 8), float(i + 1)) + torch.arange(8.0))
                                       for i, g in enumerate(got))
        ret.put((ok, mine))
    else:
        assert got is None
        ret.put((True, mine))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world2_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for ok, _ in res)
    assert sorted(i for _, m in res for i in m) == list(range(5))
