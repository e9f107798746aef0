"""world_size-2 gloo test of the multi-GPU path (clip sharding + fixed-shape all-gather of detections) on CPU."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stmask_amd import dist as sdist


def _fake_dets(clip, n):
    g = torch.Generator().manual_seed(clip)
    return {"box": torch.rand(n, 4, generator=g), "score": torch.rand(n, generator=g),
            "class": torch.randint(1, 41, (n,), generator=g), "box_ids": torch.arange(n),
            "mask_coeff": torch.randn(n, 32, generator=g)}


def _worker(rank, world, port, n_clips, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sdist.shard_clips(n_clips, rank, world)
    per = (n_clips + world - 1) // world
    dets = [_fake_dets(c, 3 + c) for c in mine]
    while len(dets) < per:  # pad the last shard
        dets.append(_fake_dets(999, 0))
    packed = sdist.pack_detections(dets, top_k=16)
    full = sdist.all_gather_detections(packed)
    order = sdist.global_clip_order(n_clips, world)
    ok = full.shape == (world * per, 16, sdist.DET_COLS)
    for c in range(n_clips):
        got = sdist.unpack_detections(full[order[c]])
        ref = _fake_dets(c, 3 + c)
        ok = ok and torch.equal(got["box"], ref["box"]) and torch.equal(got["class"], ref["class"])
        ok = ok and torch.equal(got["mask_coeff"], ref["mask_coeff"]) and torch.equal(got["box_ids"], ref["box_ids"])
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_all_gather_two_ranks():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 5, ret)) for r in range(2)]
    [p.start() for p in procs]
    [p.join(120) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert ret[0] and ret[1]


def test_sharding_is_a_partition():
    for n, w in [(8, 1), (8, 2), (8, 8), (5, 4), (3, 8)]:
        shards = [sdist.shard_clips(n, r, w) for r in range(w)]
        assert sorted(sum(shards, [])) == list(range(n))
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    assert sdist.all_gather_detections(torch.ones(2, 4, sdist.DET_COLS)).shape == (2, 4, sdist.DET_COLS)
