"""world_size-2 gloo test (CPU) of the multi-GPU layer: world sharding + the one collective
(all-gather of per-world episode statistics)."""
import importlib
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
stats = importlib.import_module("gym-exploration-2d_amd.stats")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    st = importlib.import_module("gym-exploration-2d_amd.stats")
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    start, count = st.shard_worlds(total, rank, ws)
    # synthetic per-world stats: value encodes the global world id, so the gathered order is checkable
    ids = torch.arange(start, start + count, dtype=torch.float32)
    local = {"stat_return": -ids, "stat_episodes": torch.ones(count, dtype=torch.int32),
             "stat_steps": (ids * 2).to(torch.int32),
             "stat_outcomes": torch.stack([ids, ids * 0, ids * 0 + 1], 1).to(torch.int32)}
    local["stat_steps"] = local["stat_steps"] + (1 << 25)  # beyond fp32's exact integers: the records carry int32
    packed = st.pack_episode_stats(local)
    g = st.all_gather_episode_stats(packed, total_worlds=total)
    if rank == 0:
        q.put(g.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_worlds_partition():
    for total, ws in ((4096, 8), (10, 3), (7, 8), (8192, 2)):
        parts = [stats.shard_worlds(total, r, ws) for r in range(ws)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == total
        for (s0, c0), (s1, _) in zip(parts, parts[1:]):
            assert s0 + c0 == s1


def _gather_world2(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    g = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert g.shape == (total, 6) and g.dtype == np.int32
    ids = np.arange(total)
    u = stats.unpack_episode_stats(torch.from_numpy(g))
    assert np.array_equal(u["return_sum"].numpy(), -ids.astype(np.float64))
    assert np.array_equal(u["steps"].numpy(), 2 * ids + (1 << 25)) and np.array_equal(u["n_goal"].numpy(), ids)
    s = stats.summarize(torch.from_numpy(g))
    assert s["episodes"] == total
    assert s["mean_steps"] == float((2 * ids + (1 << 25)).sum()) / total


def test_allgather_episode_stats_gloo_world2():
    _gather_world2(64)


def test_allgather_episode_stats_gloo_world2_ragged():
    """total % world_size != 0: shard_worlds gives rank 0 one world more; the gather pads and trims."""
    _gather_world2(63)


def test_single_process_passthrough():
    x = torch.zeros(4, 6, dtype=torch.int32)
    assert stats.all_gather_episode_stats(x) is x
