#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched CollisionAvoidanceEnv on N MI355X (BASELINE.json metric).

Workload (N=1): BASELINE.json configs[2] -- 4096 worlds x 10 agents, every agent RVO/ORCA (ego LP on
device) + OtherAgentsStates sensor, synthetic random-goal episodes (SURVEY.md 8(d) rule), auto-reset
from a pool of 8x4096 scenarios.  One "step" = one env.step() of all worlds of a rank.  Steps are
issued through cagym_rollout (ROLL env steps per launch, agent records on chip, every step writes
its full observation / reward / flag tensors to HBM slice t of a trajectory buffer).  ROLL defaults to 512: a launch
ends with its slowest workgroup, and longer roll-outs average the per-step variation of the LP work out.

Multi-GPU (--gpus N under torch.distributed.run): worlds are independent, each rank owns its own 4096
worlds (weak scaling); the only collective is the RCCL all-gather of per-world episode statistics,
issued once per rollout launch on a side stream.

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B_ALG = {"noncoop": 133.0 + 384.0, "rvo": 517.0}  # algorithmic bytes / agent-step
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def cpu_baseline(M, policy_id, seconds=12.0, worlds=2048, threads=None):
    """The CPU restatement (oracle, "port") timed on this host's cores: the world loop of cao_step is shared between
    OpenMP threads (worlds are independent), bounded sample."""
    threads = threads or min(os.cpu_count() or 1, 16)
    os.environ["OMP_NUM_THREADS"] = str(threads)  # read by libgomp when the oracle library starts its first team
    from oracle import oracle as orc
    scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
    orc.build()
    a6 = scen.random_worlds_fast(worlds, M, seed=99)
    env = orc.OracleEnv(N=worlds, M=M, game_over_mode=1)
    env.set_scenario(a6, policy_id, scen.DYN_UNICYCLE, coop=np.full((worlds, M), 0.5))
    env.reset()
    env.run(8)  # thread team start-up, page faults
    steps = 0
    t0 = time.perf_counter()
    while True:
        env.run(64)  # 64 x (step all worlds, restart finished ones on the same scenario: reset cost included)
        steps += 64
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    return {"value": worlds * steps / el, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": "%d worlds x %d agents, %d steps, %.1f s, C oracle (oracle/cagym_oracle.c) with its world loop on "
                      "%d OpenMP threads, finished worlds restart inside the C loop, same policy/scenario rule"
                      % (worlds, M, steps, el, threads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--worlds", type=int, default=4096, help="worlds per GPU")
    ap.add_argument("--agents", type=int, default=10)
    ap.add_argument("--policy", default="rvo", choices=["rvo", "noncoop"])
    ap.add_argument("--roll", type=int, default=512, help="env steps per launch (64: 313, 128: 334, 256: 345, 512: 360 M env-steps/s: "
                    "a launch ends with its slowest workgroup, longer roll-outs average the per-step variation out)")
    ap.add_argument("--per-step-launch", action="store_true", help="one cagym_step launch per env step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pool-factor", type=int, default=8, help="scenario pool size = factor x worlds")
    ap.add_argument("--scenarios", default="host", choices=["host", "device"],
                    help="host: numpy rejection sampler + upload; device: cagym_generate_scenarios (same rule)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    # one process per GPU; CAGYM_BENCH_BACKEND=gloo + several ranks on one card is the 1-GPU rehearsal of the N>1 path
    backend = os.environ.get("CAGYM_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    if world_size > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)

    scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
    stats_mod = importlib.import_module("gym-exploration-2d_amd.stats")
    BEnv = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
    N, M = args.worlds, args.agents
    pol = scen.POLICY_RVO if args.policy == "rvo" else scen.POLICY_NONCOOP
    S = args.pool_factor * N
    env = BEnv(N, M, n_scenarios=S, game_over_mode="all", device=device)
    if args.scenarios == "device":
        env.generate_scenarios(seed=1234 + 7919 * rank, ego_policy=pol, other_policies=(pol, pol), p_b=0.0)
    else:
        a6 = scen.random_worlds_fast(S, M, seed=1234 + 7919 * rank)
        env.set_scenarios(a6, pol, scen.DYN_UNICYCLE, coop=np.full((S, M), 0.5))
    env.reset()

    ROLL = max(1, min(args.roll, args.steps))
    traj = env.alloc_rollout(ROLL)
    side = torch.cuda.Stream(device=device) if world_size > 1 else None
    gathered = None

    def run(n_steps):
        nonlocal gathered
        done = 0
        launches = 0
        while done < n_steps:
            k = min(ROLL, n_steps - done)
            if args.per_step_launch:
                for _ in range(k):
                    env.step(auto_reset=True)
                    launches += 1
            else:
                env.rollout(k, auto_reset=True, out=traj)
                launches += 1
            done += k
            if world_size > 1:
                # episode-stats all-gather on a side stream, overlapping the next launch
                local = stats_mod.pack_episode_stats(env.episode_stats())
                side.wait_stream(torch.cuda.current_stream(device))
                local.record_stream(side)  # allocated on the main stream, read by the collective on the side stream
                with torch.cuda.stream(side):
                    gathered = stats_mod.all_gather_episode_stats(local)
        if side is not None:
            torch.cuda.current_stream(device).wait_stream(side)
        return launches

    def barrier():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    run(args.warmup)
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    launches = run(args.steps)
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    if world_size > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    dev_ms = ev0.elapsed_time(ev1)

    st = stats_mod.summarize(gathered if gathered is not None else stats_mod.pack_episode_stats(env.episode_stats()))
    if rank == 0:
        total_worlds = N * world_size
        value = total_worlds * args.steps / elapsed
        steps_per_launch = args.steps / launches
        launch_ms = dev_ms / launches
        balg = B_ALG[args.policy]
        # the library's own rule (cagym_api.hip): M = 10 runs 4 worlds per workgroup while the launch is co-resident
        cus = torch.cuda.get_device_properties(device).multi_processor_count
        spec = {10: "256, 10, %d" % (4 if (N + 3) // 4 <= 5 * cus else 5), 4: "256, 4, 0", 20: "256, 20, 2"}
        kernel_name = "%s<%s, true>" % ("k_step2" if args.per_step_launch else "k_rollout2",
                                        spec.get(M, "256, 0, 0" if M <= 12 else "512, 0, 0"))
        achieved = balg * N * M * steps_per_launch / (launch_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate
        # rocprofv3 --pmc passes; profiles/r1/bench_4096x10_rvo_pmc_hbm.txt): 418.5 B per agent-step at 512 steps per
        # launch, 423.9 B at 64 (per-step outputs 417.7 B + 395 B of state in/out per agent and launch); measured on this
        # workload only and not re-measured live, so null for any other shape.
        traffic = None
        traffic_per_agent_step = 417.7 + 395.0 / steps_per_launch
        if (N, M, args.policy, args.per_step_launch) == (4096, 10, "rvo", False):
            traffic = traffic_per_agent_step * N * M * steps_per_launch / (launch_ms * 1e-3) / 1e9  # GB/s, comparable to `achieved`
        # measured device-to-device copy ceiling next to the vendor HBM figure: read + write of 1 GiB
        src = torch.empty(1 << 30, dtype=torch.uint8, device=device)
        dst = torch.empty_like(src)
        dst.copy_(src)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0.record()
        for _ in range(5):
            dst.copy_(src)
        c1.record()
        torch.cuda.synchronize(device)
        copy_gbs = 5 * 2 * (1 << 30) / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del src, dst
        line = {
            "metric": "env-steps/sec (whole node), 4096 worlds x 10 agents",
            "value": value, "unit": "env-steps/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d worlds x %d agents per GPU, %s policy + OtherAgentsStates sensor, "
                                   "UnicycleDynamics, random-goal episodes with auto-reset (BASELINE configs[2])"
                                   % (N, M, "RVO/ORCA on-device LP" if args.policy == "rvo" else "NonCooperative"),
                       "worlds_per_gpu": N, "agents": M, "steps_per_launch": steps_per_launch,
                       "launch_mode": "cagym_step per step" if args.per_step_launch else "cagym_rollout",
                       "parallelism": "worlds sharded x%d, no data-path collective" % world_size},
            "agent_steps_per_s": value * M,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": kernel_name,
                         "launch_ms": launch_ms, "alg_bytes_per_agent_step": balg,
                         "measured_d2d_copy_GBs": copy_gbs,
                         "alg_bytes_per_launch": balg * N * M * steps_per_launch,
                         "traffic_bytes_per_launch": None if traffic is None else traffic_per_agent_step * N * M * steps_per_launch},
            "episodes": st,
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(M, pol)
        print(json.dumps(line))
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
