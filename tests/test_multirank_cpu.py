"""CPU, world_size 2 over gloo: the clip sharding and the one end-of-job collective (all-gather of the
fixed-size token records) that the N>1 bench path uses over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from clearconverse_amd.batch import broadcast_weights, gather_transcripts, shard_clip_indices


def test_sharding_is_a_partition():
    for n, w in [(256, 8), (5, 2), (3, 4)]:
        parts = [shard_clip_indices(n, r, w) for r in range(w)]
        assert sorted(i for p in parts for i in p) == list(range(n))
        assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_clip_indices(6, rank, world)
    records = [dict(tokens=[50363 + i, 100 + i, 7 * i]) for i in mine]       # a deterministic "transcript" per clip
    out = gather_transcripts(records, sample_len=8, eot=50256, device="cpu")
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                                  # the timing reduction of bench.py
    q.put((rank, out.tolist(), float(t)))
    dist.destroy_process_group()


def test_all_gather_of_token_records_world2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, rows, tmax in got:
        assert tmax == 2.0
        assert len(rows) == 6
        order = [0, 2, 4, 1, 3, 5]                                            # rank-major concatenation
        for row, clip in zip(rows, order):
            assert row[0] == 3 and row[1:4] == [50363 + clip, 100 + clip, 7 * clip] and row[4:] == [50256] * 5
    assert got[0][1] == got[1][1]


def _bcast_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sds = None
    if rank == 0:
        g = torch.Generator().manual_seed(7)
        sds = {"dims": {"n_mels": 80, "n_vocab": 51864}, "source": "synthetic-seed7",
               "net_a": {"w": torch.randn(5, 3, generator=g), "b": torch.randn(7, generator=g).to(torch.float16), "powerset": 1},
               "net_b": {"idx": torch.arange(11, dtype=torch.int64), "z": torch.randn(2, 2, 2, generator=g)}}
    got = broadcast_weights(sds, src=0, device="cpu")
    assert all(v.device.type == "cpu" for d in got.values() if isinstance(d, dict) for v in d.values() if torch.is_tensor(v))
    q.put((rank, {m: ({k: (v.tolist(), str(v.dtype)) if torch.is_tensor(v) else v for k, v in d.items()} if isinstance(d, dict) else d)
                  for m, d in got.items()}))
    dist.destroy_process_group()


def test_weight_broadcast_world2():
    """C1: rank 0's nested state dicts (tensors of several dtypes + plain entries) arrive identically on rank 1."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1]
    assert got[1]["dims"] == {"n_mels": 80, "n_vocab": 51864} and got[1]["source"] == "synthetic-seed7"
    assert got[1]["net_a"]["powerset"] == 1 and got[1]["net_a"]["b"][1] == "torch.float16"
    assert got[1]["net_b"]["idx"][0] == list(range(11))


def test_weight_broadcast_without_process_group_is_identity():
    sds = {"m": {"w": torch.ones(3)}}
    assert broadcast_weights(sds) is sds
    with pytest.raises(ValueError):
        broadcast_weights(None)
