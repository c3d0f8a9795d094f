#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched CollisionAvoidanceEnv on N MI355X (BASELINE.json metric).

Workload (N=1, --config cfg3, the default): BASELINE.json configs[2] -- 4096 worlds x 10 agents, every agent RVO/ORCA
(ego LP on device) + OtherAgentsStates sensor, synthetic random-goal episodes (SURVEY.md 8(d) rule), auto-reset from a
pool of 8x4096 scenarios.  One "step" = one env.step() of all worlds of a rank.  Steps are issued through cagym_rollout
(up to --roll env steps per launch, agent records on chip, every step writes its full observation / reward / flag
tensors to HBM slice t of a trajectory buffer).

Timing (repeats, median): after --warmup untimed steps the block of EXACTLY --steps steps is timed
`repeats` times, each time bracketed by barrier + torch.cuda.synchronize() on both sides and max-reduced over ranks;
`ms_per_step` / `value` are the MEDIAN block (min / max beside it).

Multi-GPU: `python bench.py --gpus N` with no RANK in the environment starts N one-GPU ranks itself
(torch.distributed.run, before anything touches the GPU in this process); under a launcher (RANK set) it is one rank.
Worlds are independent, each rank owns its own 4096 worlds (weak scaling); the only collective is the RCCL all-gather
of per-world episode statistics, issued once per rollout launch on a side stream.

Other rows of SURVEY 8(d): --config cfg2 (4096 x 4 NonCooperative), cfg4 (8192 x 10: GA3C-CADRL agent 0 + 9 RVO among
rectangles, LaserScan), cfg5 (2048 x 20 information-gain env part + planner primitives).

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
# algorithmic bytes per agent-step: 68 B state in + 60 B out + 5 B reward/flags = 133, OAS + ego 384 (K = 9)
B_STATE, B_OAS9, B_OAS19, B_LASER = 133.0, 384.0, 784.0, 64.0


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=256)
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--worlds", type=int, default=None, help="worlds per GPU (default: the config's)")
    ap.add_argument("--agents", type=int, default=None)
    ap.add_argument("--policy", default=None, choices=["rvo", "noncoop"])
    ap.add_argument("--roll", type=int, default=512, help="max env steps per launch")
    ap.add_argument("--repeats", type=int, default=0, help="timed blocks (0 = at least 5, more while the blocks are short)")
    ap.add_argument("--per-step-launch", action="store_true", help="one cagym_step_autoreset launch per env step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pool-factor", type=int, default=8, help="scenario pool size = factor x worlds")
    ap.add_argument("--scenarios", default="host", choices=["host", "device"],
                    help="host: numpy rejection sampler + upload; device: cagym_generate_scenarios (same rule)")
    return ap.parse_args(argv)


def self_launch(args):
    """--gpus N without a launcher: start N ranks (one per GPU) as children of this process, which has not touched
    the GPU (no torch import yet), and pass their output and exit code through."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def cpu_baseline(M, policy_id, seconds=12.0, worlds=2048, threads=None, config="cfg3"):
    """The CPU restatement (oracle, "port") timed on this host's cores: the world loop of cao_step is shared between
    OpenMP threads (worlds are independent), bounded sample."""
    import numpy as np
    from oracle import oracle as orc
    scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
    orc.build()
    want = threads or min(os.cpu_count() or 1, 16)
    threads = int(orc.lib().cao_set_threads(int(want)))  # the team size actually in force (omp_get_max_threads)
    note = ""
    if config in ("cfg4", "cfg5"):
        K = 10 if config == "cfg4" else 8
        worlds = 512
        a6, ob, nob, _ = scen.obstacle_worlds(worlds, M, K, seed=99)
        pol = np.full((worlds, M), scen.POLICY_RVO if config == "cfg4" else scen.POLICY_NONCOOP, dtype=np.int32)
        if config == "cfg4":
            pol[:, 0] = scen.POLICY_NONCOOP
            note = " (agent 0 NonCooperative in place of the GA3C network, 9 RVO among rectangles, LaserScan on)"
        else:
            pol[:, :3] = scen.POLICY_NONCOOP
            pol[:, 3:5] = scen.POLICY_STATIC
            note = " (env part only: no planner)"
        env = orc.OracleEnv(N=worlds, M=M, max_obstacles=K, game_over_mode=0 if config == "cfg4" else 1, laserscan=config == "cfg4")
        env.set_scenario(a6, pol, scen.DYN_UNICYCLE, coop=np.full((worlds, M), 0.5), obstacles=ob, n_obst=nob)
    else:
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
                      "%d OpenMP threads (omp_get_max_threads), finished worlds restart inside the C loop, same policy/scenario rule%s"
                      % (worlds, M, steps, el, threads, note)}


def main():
    args = parse_args()
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world_size:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or without a launcher)"
                 % (args.gpus, world_size, args.gpus))
    import numpy as np
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    defaults = {"cfg2": (4096, 4, "noncoop"), "cfg3": (4096, 10, "rvo"), "cfg4": (8192, 10, "rvo"), "cfg5": (2048, 20, "noncoop")}[args.config]
    M = args.agents or defaults[1]
    policy = args.policy or defaults[2]
    pol = scen.POLICY_RVO if policy == "rvo" else scen.POLICY_NONCOOP
    if args.config in ("cfg4", "cfg5"):
        # BASELINE quotes these two on a FIXED number of worlds sharded over the GPUs (strong scaling)
        total = args.worlds or defaults[0]
        N = stats_mod.shard_worlds(total, rank, world_size)[1]
        scaling = "strong"
    else:
        N = args.worlds or defaults[0]  # per GPU (weak scaling)
        scaling = "weak"
    S = args.pool_factor * N
    extra = {}
    side = torch.cuda.Stream(device=device) if world_size > 1 else None
    gathered = None

    def gather_stats(env):
        nonlocal gathered
        if world_size > 1:
            # episode-stats all-gather on a side stream, overlapping the next launch
            local = stats_mod.pack_episode_stats(env.episode_stats())
            side.wait_stream(torch.cuda.current_stream(device))
            local.record_stream(side)  # allocated on the main stream, read by the collective on the side stream
            with torch.cuda.stream(side):
                gathered = stats_mod.all_gather_episode_stats(local, total_worlds=None if scaling == "weak" else total)

    if args.config in ("cfg2", "cfg3"):
        env = BEnv(N, M, n_scenarios=S, game_over_mode="all", device=device)
        if args.scenarios == "device":
            env.generate_scenarios(seed=1234 + 7919 * rank, ego_policy=pol, other_policies=(pol, pol), p_b=0.0)
        else:
            a6 = scen.random_worlds_fast(S, M, seed=1234 + 7919 * rank)
            env.set_scenarios(a6, pol, scen.DYN_UNICYCLE, coop=np.full((S, M), 0.5))
        env.reset()
        ROLL = max(1, min(args.roll, args.steps))
        traj = env.alloc_rollout(ROLL)
        balg = B_STATE + (B_OAS9 if M <= 10 else B_OAS19)
        workload = ("%d worlds x %d agents per GPU, %s policy + OtherAgentsStates sensor, UnicycleDynamics, random-goal "
                    "episodes with auto-reset (BASELINE configs[%d])"
                    % (N, M, "RVO/ORCA on-device LP" if policy == "rvo" else "NonCooperative", 2 if args.config == "cfg3" else 1))
        launch_mode = "cagym_step_autoreset per step" if args.per_step_launch else "cagym_rollout"
        kernel_name = env.kernel_name(rollout=not args.per_step_launch, auto_reset=True)

        def run(n_steps):
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
                gather_stats(env)
            if side is not None:
                torch.cuda.current_stream(device).wait_stream(side)
            return launches
    elif args.config == "cfg4":
        # agent 0 GA3C-CADRL (state kernel + fused forward kernel per step) + 9 RVO agents among 2-10 rectangles,
        # LaserScan on every agent (scanned inside the step launch), game over when agent 0 is done, auto-reset
        GA3C = importlib.import_module("gym-exploration-2d_amd.ga3c").GA3CCADRLPolicy
        K = 10
        S = 2 * N
        a6, ob, nob, _ = scen.obstacle_worlds(S, M, K, seed=1234 + 7919 * rank)
        pol4 = np.full((S, M), scen.POLICY_RVO, dtype=np.int32)
        pol4[:, 0] = scen.POLICY_GA3C
        env = BEnv(N, M, n_scenarios=S, max_obstacles=K, laserscan=True, game_over_mode="agent0", device=device)
        env.set_scenarios(a6, pol4, scen.DYN_UNICYCLE, coop=np.full((S, M), 0.5), obstacles=ob, n_obst=nob)
        env.reset()
        ga3c = GA3C(env)
        ext = torch.zeros((N, M, 2), dtype=torch.float32, device=device)
        balg = B_STATE + B_OAS9 + B_LASER + 11250.0 / M  # env kernel: + scan out + the world's bit-packed raster once per step
        workload = ("%d worlds x %d agents%s: agent 0 GA3C-CADRL (fused LSTM-64 + 3 x FC-256 forward per step), 9 RVO/ORCA agents "
                    "among 2-10 rectangles (obstacle half-planes), LaserScan + OtherAgentsStates on every agent, auto-reset "
                    "(BASELINE configs[3])" % (N, M, " on this rank" if world_size > 1 else ""))
        launch_mode = "per step: cagym_ga3c_act (device-side selection + state vectors + fused forward, no host sync) + cagym_step_autoreset (laser scan inside)"
        kernel_name = env.kernel_name(rollout=False, auto_reset=True)

        def run(n_steps):
            for _ in range(n_steps):
                ga3c.act(ext)
                env.step(ext, auto_reset=True)
            gather_stats(env)
            if side is not None:
                torch.cuda.current_stream(device).wait_stream(side)
            return n_steps

        def cfg4_extra():
            def loop(fn, reps=50):
                fn()
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize(device)
                return (time.perf_counter() - t0) / reps
            t_nn = loop(lambda: ga3c.act(ext))
            t_env = loop(lambda: env.step(ext, auto_reset=True))
            return {"ga3c_evals_per_s": N / t_nn, "ga3c_ms_per_step": 1e3 * t_nn, "ga3c_tflops_fp32": N * 0.67e6 / t_nn / 1e12,
                    "env_kernel_ms_per_step": 1e3 * t_env,
                    "env_kernel_hbm_frac": balg * N * M / t_env / (HBM_PEAK_GBS * 1e9)}
        extra["cfg4"] = cfg4_extra
    else:  # cfg5: env part (3 IG agents driven externally + 2 static targets + 15 NonCooperative) + planner primitives
        IG = importlib.import_module("gym-exploration-2d_amd.ig").InfoGain
        K = 8
        a6, ob, nob, _ = scen.obstacle_worlds(S, M, K, seed=1234 + 7919 * rank)
        pol5 = np.full((S, M), scen.POLICY_NONCOOP, dtype=np.int32)
        pol5[:, :3] = scen.POLICY_IGMCTS
        pol5[:, 3:5] = scen.POLICY_STATIC
        dyn5 = np.full((S, M), scen.DYN_UNICYCLE, dtype=np.int32)
        dyn5[:, :3] = scen.DYN_FIRSTORDER
        env = BEnv(N, M, n_scenarios=S, max_obstacles=K, game_over_mode="all", device=device)
        env.set_scenarios(a6, pol5, dyn5, obstacles=ob, n_obst=nob)
        env.reset()
        ext = torch.zeros((N, M, 2), dtype=torch.float32, device=device)
        ext[:, :3, 0] = 2.0
        balg = B_STATE + B_OAS19
        workload = ("%d worlds x %d agents%s, env part: 3 information-gain agents (external (v, omega), FirstOrderDynamics), 2 static "
                    "targets, 15 NonCooperative, rectangles, OtherAgentsStates [19, 10] (BASELINE configs[4]); the planner part is "
                    "reported as visibility queries/s and roll-outs/s" % (N, M, " on this rank" if world_size > 1 else ""))
        launch_mode = "cagym_step_autoreset per step"
        kernel_name = env.kernel_name(rollout=False, auto_reset=True)

        def run(n_steps):
            for _ in range(n_steps):
                env.step(ext, auto_reset=True)
            gather_stats(env)
            if side is not None:
                torch.cuda.current_stream(device).wait_stream(side)
            return n_steps

        def cfg5_extra():
            ig = IG(env)
            rng = np.random.default_rng(0)
            Q = N * 32
            poses = torch.from_numpy(np.concatenate([rng.uniform(-12, 12, (Q, 2)), rng.uniform(-np.pi, np.pi, (Q, 1))], 1)).to(device)
            world = torch.arange(Q, device=device, dtype=torch.int32) % N

            def loop(fn, reps=10):
                fn()
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize(device)
                return (time.perf_counter() - t0) / reps
            t_vis = loop(lambda: ig.visible_cells(poses, world))
            Qr, nsims, H = N * 3, 10, 4  # experiments/src/dmcts.py budget: Nsims 10, horizon 4, xdt 5
            zeros = torch.zeros((Qr, 60), dtype=torch.int64, device=device)
            t_ro = loop(lambda: ig.rollouts(poses[:Qr], zeros, zeros, world[:Qr], torch.full((Qr,), H), torch.full((Qr,), 0.5), nsims, 7))
            return {"visibility_queries_per_s": Q / t_vis, "rollouts_per_s_horizon4": Qr * nsims / t_ro,
                    "rollout_visibility_queries_per_s": Qr * nsims * H / t_ro,
                    "l2_hit_rate": None, "l2_hit_rate_source": "profiles/ (rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum pass), not measured in this run"}
        extra["cfg5"] = cfg5_extra

    def barrier():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    def timed_block():
        """EXACTLY --steps steps between two barrier + synchronize brackets; returns (wall s maxed over ranks, device ms, launches)."""
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        launches = run(args.steps)
        ev1.record()
        barrier()
        el = time.perf_counter() - t0
        if world_size > 1:
            t = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, ev0.elapsed_time(ev1), launches

    run(args.warmup)
    barrier()
    blocks = [timed_block()]
    if args.repeats > 0:
        repeats = args.repeats
    else:  # at least 5 blocks; short blocks are repeated until about 2 s are on the clock (same count on every rank)
        repeats = int(min(400, max(5, 2.0 / max(blocks[0][0], 1e-6))))
        if world_size > 1:
            r = torch.tensor([repeats], dtype=torch.int64, device=device)
            dist.broadcast(r, 0)
            repeats = int(r.item())
    while len(blocks) < repeats:
        blocks.append(timed_block())
    walls = sorted(b[0] for b in blocks)
    elapsed = walls[len(walls) // 2]  # median block
    dev_sorted = sorted(b[1] for b in blocks)
    dev_ms = dev_sorted[len(dev_sorted) // 2]
    launches = blocks[0][2]

    st = stats_mod.summarize(gathered if gathered is not None else stats_mod.pack_episode_stats(env.episode_stats()))
    if rank == 0:
        total_worlds = N * world_size if scaling == "weak" else total
        value = total_worlds * args.steps / elapsed
        steps_per_launch = args.steps / launches
        launch_ms = dev_ms / launches
        achieved = balg * N * M * steps_per_launch / (launch_ms * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters are NOT collected in this run (they need rocprofv3 --pmc passes);
        # the figure below replays profiles/ (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes) for the
        # headline shape only and is labelled as such; null for any other shape.
        traffic = None
        traffic_per_agent_step = 417.7 + 395.0 / steps_per_launch
        if (N, M, policy, args.per_step_launch) == (4096, 10, "rvo", False):
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
            "metric": "env-steps/sec (whole node), %d worlds x %d agents" % (total_worlds if scaling == "strong" else N, M),
            "value": value, "unit": "env-steps/s", "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "worlds_per_gpu": N, "agents": M, "steps_per_launch": steps_per_launch,
                       "launch_mode": launch_mode,
                       "parallelism": "worlds sharded x%d, no data-path collective" % world_size},
            "repeats": len(blocks), "timing": "median of %d timed blocks of %d steps" % (len(blocks), args.steps),
            "ms_per_step_min": 1e3 * walls[0] / args.steps, "ms_per_step_max": 1e3 * walls[-1] / args.steps,
            "rccl_ranks": world_size if (world_size > 1 and backend == "nccl") else 0,
            "collective_backend": backend if world_size > 1 else None,
            "agent_steps_per_s": value * M,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": None if traffic is None else
                         "replayed from profiles/r2 PMC passes of this shape (FETCH_SIZE x2 + WRITE_SIZE per agent-step), not measured in this run",
                         "kernel": kernel_name,
                         "launch_ms": launch_ms, "launch_ms_min": dev_sorted[0] / launches, "launch_ms_max": dev_sorted[-1] / launches,
                         "alg_bytes_per_agent_step": balg,
                         "measured_d2d_copy_GBs": copy_gbs,
                         "alg_bytes_per_launch": balg * N * M * steps_per_launch,
                         "traffic_bytes_per_launch": None if traffic is None else traffic_per_agent_step * N * M * steps_per_launch},
            "episodes": st,
        }
        for name, fn in extra.items():
            line[name] = fn()
        if args.config == "cfg4":  # the timed step holds three launches: the roofline object prices the env kernel alone
            line["roofline"].update({"achieved": line["cfg4"]["env_kernel_hbm_frac"] * HBM_PEAK_GBS, "frac": line["cfg4"]["env_kernel_hbm_frac"],
                                     "launch_ms": line["cfg4"]["env_kernel_ms_per_step"], "launch_ms_min": None, "launch_ms_max": None,
                                     "note": "env kernel alone (cagym_step_autoreset incl. laser scan), timed in its own loop"})
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(M, pol, config=args.config)
        print(json.dumps(line), flush=True)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
