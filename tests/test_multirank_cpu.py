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


def test_bench_spawns_its_own_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher (how the driver calls it) must start N ranks itself, before anything
    touches the GPU, on 127.0.0.1, and hand back the launcher's exit code."""
    import importlib
    import subprocess
    import sys
    bench = importlib.import_module("bench")
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # a failing rank -> non-zero exit of the parent
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_parent_fails_when_a_rank_fails():
    """End to end on a box without a GPU: the ranks die at torch.cuda.set_device, the parent must exit non-zero and print no JSON."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without GPUs (the ranks would run the real bench)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert not any(line.startswith('{"metric"') for line in r.stdout.splitlines())
