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
    # ragged shards (5 clips over 2 ranks = 3 + 2): padded gather, one collective per rank, exact counts back
    mine_clips = torch.stack([torch.full((3, 2, 4, 4), float(i)) for i in mine]) if mine else torch.empty((0, 3, 2, 4, 4))
    rag = gather_clips(mine_clips, dst=0, ragged=True)
    if rank == 0:
        flat = [int(c[0, 0, 0, 0]) for part in rag for c in part]
        assert [len(p) for p in rag] == [len(shard_indices(5, world, r)) for r in range(world)] and flat == list(range(5))
    else:
        assert rag is None
    dist.barrier()
    dist.destroy_process_group()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` outside torchrun starts two ranks itself (torch.distributed.run on 127.0.0.1);
    checked on CPU with the launcher self-test leg (gloo): rendezvous, gather to rank 0, ONE JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launcher-selftest"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"selftest": "launcher", "n_gpus": 2, "gathered": [0.0, 1.0]}


def test_bench_launches_eight_ranks():
    """The first real 8-GPU lease must not also be the first 8-rank rendezvous: `python bench.py --gpus 8
    --launcher-selftest` (world 8 over gloo on this CPU box) -- torch.distributed.run on 127.0.0.1, eight ranks, the one
    gather to rank 0 (reference scripts/evaluation/ddp_wrapper.py:8-47, inference.py:314-320: clips sharded, no data-path
    collective but the final gather), ONE JSON line."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--launcher-selftest"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"selftest": "launcher", "n_gpus": 8, "gathered": [float(r) for r in range(8)]}


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


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus 2` on a box with fewer than 2 GPUs must fail at once and say why -- not start ranks that
    sit in the collective-init timeout (here: 0 GPUs)."""
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-300:])
    assert "refusing to start 2 ranks" in r.stderr
    assert time.time() - t0 < 120
