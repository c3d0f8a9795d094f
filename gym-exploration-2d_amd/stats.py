"""Multi-GPU: worlds are independent, so ranks shard them with no data-path exchange.  The single
collective is an all-gather of the per-world episode statistics (24 B/world) over RCCL
(torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).  The reference has no
counterpart (single process); its per-episode statistics are those of
experiments/src/env_utils.py:41-75 (return, steps, outcome flags of the previous episode).

Record per world: {f32 return_sum, i32 episodes, i32 steps, i32 n_goal, i32 n_collision, i32 n_timeout}
carried as six 32-bit words (the return's bit pattern in word 0), so the counters stay exact however long
the run is (fp32 would lose exactness past 2^24).
"""
import torch

RECORD_WORDS = 6


def shard_worlds(total_worlds, rank, world_size):
    """Contiguous block partition of `total_worlds` (first ranks take the remainder)."""
    base, rem = divmod(int(total_worlds), int(world_size))
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def pack_episode_stats(stats):
    """[N, 6] int32 records (word 0 = bit pattern of the fp32 return sum)."""
    ret = stats["stat_return"].float().contiguous().view(torch.int32)
    out = stats["stat_outcomes"].to(torch.int32)
    cols = [ret, stats["stat_episodes"].to(torch.int32), stats["stat_steps"].to(torch.int32), out[:, 0], out[:, 1], out[:, 2]]
    return torch.stack(cols, dim=1).contiguous()


def all_gather_episode_stats(local, group=None, total_worlds=None):
    """all-gather [N_local, 6] -> [sum of N_local, 6].  Shards may differ by one world (shard_worlds): every rank pads
    to ceil(total / world_size) rows for the one equal-size collective and the padding is trimmed afterwards.
    total_worlds = None means equal shards."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return local
    ws = dist.get_world_size(group)
    n_local = int(local.shape[0])
    if total_worlds is None:
        rows, counts = n_local, [n_local] * ws
    else:
        rows = -(-int(total_worlds) // ws)
        counts = [shard_worlds(total_worlds, r, ws)[1] for r in range(ws)]
        if counts[dist.get_rank(group)] != n_local:
            raise ValueError("rank holds %d worlds, shard_worlds(%d, rank, %d) says %d"
                             % (n_local, total_worlds, ws, counts[dist.get_rank(group)]))
    send = local.contiguous()
    if rows != n_local:
        send = torch.zeros((rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[:n_local] = local
    out = torch.empty((ws * rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, send, group=group)
    if all(c == rows for c in counts):
        return out
    return torch.cat([out[r * rows:r * rows + counts[r]] for r in range(ws)], dim=0)


def unpack_episode_stats(packed):
    """[n, 6] int32 records -> dict of exact columns (return as float64)."""
    p = packed.contiguous()
    return {"return_sum": p[:, 0].contiguous().view(torch.float32).double(), "episodes": p[:, 1].long(), "steps": p[:, 2].long(),
            "n_goal": p[:, 3].long(), "n_collision": p[:, 4].long(), "n_timeout": p[:, 5].long()}


def summarize(gathered):
    u = unpack_episode_stats(gathered)
    eps_n = int(u["episodes"].sum())
    eps = max(float(eps_n), 1.0)
    goal, coll, tout = int(u["n_goal"].sum()), int(u["n_collision"].sum()), int(u["n_timeout"].sum())
    agents = max(float(goal + coll + tout), 1.0)
    return {"episodes": float(eps_n), "mean_return": float(u["return_sum"].sum()) / eps,
            "mean_steps": float(int(u["steps"].sum())) / eps, "frac_goal": goal / agents,
            "frac_collision": coll / agents, "frac_timeout": tout / agents}
