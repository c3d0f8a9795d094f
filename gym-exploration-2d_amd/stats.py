"""Multi-GPU: worlds are independent, so ranks shard them with no data-path exchange.  The single
collective is an all-gather of the per-world episode statistics (16 B/world) over RCCL
(torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).  The reference has no
counterpart (single process); its per-episode statistics are those of
experiments/src/env_utils.py:41-75 (return, steps, outcome flags of the previous episode).
"""
import torch


def shard_worlds(total_worlds, rank, world_size):
    """Contiguous block partition of `total_worlds` (first ranks take the remainder)."""
    base, rem = divmod(int(total_worlds), int(world_size))
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def pack_episode_stats(stats):
    """[N, 6] float32: return_sum, episodes, steps, n_goal, n_collision, n_timeout (24 B/world)."""
    cols = [stats["stat_return"].float(), stats["stat_episodes"].float(), stats["stat_steps"].float()]
    out = stats["stat_outcomes"].float()
    return torch.stack(cols + [out[:, 0], out[:, 1], out[:, 2]], dim=1).contiguous()


def all_gather_episode_stats(local, group=None):
    """all-gather [N_local, 6] -> [world_size * N_local, 6] (equal shard sizes)."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return local
    ws = dist.get_world_size(group)
    out = torch.empty((ws * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def summarize(gathered):
    g = gathered.double().sum(0)
    eps = max(float(g[1]), 1.0)
    agents = max(float(g[3] + g[4] + g[5]), 1.0)
    return {"episodes": float(g[1]), "mean_return": float(g[0]) / eps, "mean_steps": float(g[2]) / eps,
            "frac_goal": float(g[3]) / agents, "frac_collision": float(g[4]) / agents,
            "frac_timeout": float(g[5]) / agents}
