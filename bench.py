#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched CollisionAvoidanceEnv on N MI355X (BASELINE.json metric).

Workload (N=1, --config cfg3, the default): BASELINE.json configs[2] -- 4096 worlds x 10 agents, every agent RVO/ORCA
(ego LP on device) + OtherAgentsStates sensor, synthetic random-goal episodes (SURVEY.md 8(d) rule), auto-reset from a
pool of 8x4096 scenarios.  One "step" = one env.step() of all worlds of a rank.  Steps are issued through cagym_rollout
(up to --roll env steps per launch, agent records on chip, every step writes its full observation / reward / flag
tensors to HBM slice t of a trajectory buffer).

Timing (repeats, median): after --warmup untimed steps the block of EXACTLY --steps steps is timed
`repeats` times, each time bracketed by barrier + torch.cuda.synchronize() on both sides and max-reduced over ranks (the
clock of a rank stops when its own closing synchronize returns, the closing barrier follows: the collective's latency is
the bench's, not the K steps'); `ms_per_step` / `value` are the MEDIAN block (min / max beside it).  Short blocks are
repeated until the timed blocks cover at least 2 s whatever --steps is (the driver's 20-step block is 0.2 ms: ~9000
blocks).  ONE clock: `value`, `ms_per_step` and `roofline.achieved` / `frac` all come from the wall-clock median block;
HIP events ride in every SECOND block only (recording one costs the host 4 us) and give `roofline.launch_ms_hip_events` /
`frac_hip_events`; the wall clock is the median of the blocks without them.

Multi-GPU: `python bench.py --gpus N` with no RANK in the environment starts N one-GPU ranks itself
(torch.distributed.run, before anything touches the GPU in this process); under a launcher (RANK set) it is one rank.
Worlds are independent, each rank owns its own 4096 worlds (weak scaling); the only collective is the RCCL all-gather
of per-world episode statistics (24-byte records packed by one kernel, cagym_pack_episode_stats) on a side stream.  The
counters are cumulative, so it is issued once every G timed blocks, G chosen from its measured duration so that it costs
at most 1 % of the timed time; the main stream never waits for the side stream inside a block (the bracketing
synchronize at the block's end is the only join).

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


def profile_replay(name):
    """Counters of a secondary configuration, collected by the rocprofv3 --pmc passes of tools/final_profile_r4.sh and stored by
    tools/store_profiles_r4.py (profiles/r4/<name>): replayed into the line, labelled as such; None when the file is not there."""
    q = os.path.join(ROOT, "profiles", "r4", name)
    return json.load(open(q)) if os.path.exists(q) else None


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
    ap.add_argument("--halves", type=int, default=1, help="cfg4: split the rank's worlds into this many independent groups, each with its own handle and HIP stream, so that one group's GA3C forward (matrix cores) overlaps another group's env step (vector units); 1 = one group (measured, round 4: 0.147 ms with 1, 0.190 with 2 - the forward is one latency-bound workgroup per CU whatever the number of worlds; round 3: 0.302 / 0.383 / 0.500 with 1 / 2 / 4: not kept as default)")
    ap.add_argument("--cfg4-step", default="fused", choices=["fused", "split", "overlap"],
                    help="cfg4: fused = cagym_ga3c_act, then cagym_step_autoreset (default; round 4: time-shared LDS, four workgroups per CU); "
                         "split = cagym_step_begin, cagym_ga3c_act, cagym_step_finish on one stream; overlap = cagym_step_begin on a side stream BESIDE "
                         "cagym_ga3c_act, then cagym_step_finish.  Bit-identical results (tests/test_split_step.py); measured 0.194 / 0.213 / 0.251 ms per step: "
                         "side by side on a CU the two kernels halve each other's occupancy and each takes twice as long (profiles/r4/cfg4_overlap*.txt)")
    ap.add_argument("--max-obstacles", type=int, default=10, help="cfg4: rectangles per world are drawn from 2..this (BASELINE: 10); occupancy experiments only")
    ap.add_argument("--graph", action="store_true", help="cfg4: replay the step's launches from one captured HIP graph (measured: 0.311 vs 0.303 ms eager - the step is not launch-bound; profiles/r3/cfg4_graph_vs_eager.txt)")
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
    # The reference's own Python env.step() cannot run on the GPU box (its source never travels): its figure is the one measured in the
    # development container on the same scenario rule, reported beside the port with its source
    ref_py = {4: (1260.0, 11939.0), 10: (457.0, 3412.0)}.get(M)
    reference_python = None if ref_py is None else {
        "value_1_process": ref_py[0], "value_8_processes": ref_py[1], "unit": "env-steps/s", "agents": M,
        "policy": "NonCooperative (the rvo2 library the reference's RVOPolicy drives is absent: RVO could not be timed)",
        "host": "8 vCPU Intel Xeon @ 2.10 GHz (development container), Python 3.10, NumPy 2.2",
        "source": "BASELINE.md section 2: unmodified reference modules imported from the reference checkout, free-space random-goal episodes, auto-reset, 8 s per point; NOT measured in this run"}
    return {"value": worlds * steps / el, "unit": "env-steps/s", "cores": threads, "kind": "port", "reference_python": reference_python,
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

    stats_state = {"every": 1, "block": 0, "collective_ms": None, "probing": False}

    def gather_stats(env, force=False):
        """Episode-statistics all-gather (multi-GPU only): one pack kernel on the main stream, the collective on the side
        stream; issued in one of every stats_state["every"] timed blocks (cumulative counters: nothing is lost)."""
        nonlocal gathered
        if world_size > 1 and (force or (not stats_state["probing"] and stats_state["block"] % stats_state["every"] == 0)):
            local = env.packed_episode_stats()
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
            gather_stats(env)  # at most once per block; joined by the bracketing synchronize, never by the main stream
            return launches
    elif args.config == "cfg4":
        # agent 0 GA3C-CADRL (state kernel + fused forward kernel per step) + 9 RVO agents among 2-10 rectangles,
        # LaserScan on every agent (scanned inside the step launch), game over when agent 0 is done, auto-reset
        GA3C = importlib.import_module("gym-exploration-2d_amd.ga3c").GA3CCADRLPolicy
        K = args.max_obstacles
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
        split = args.cfg4_step in ("split", "overlap")
        if split:
            launch_mode = ("per step: cagym_step_begin (k_step_pre3: obstacle + agent half-planes and linear programs of the 9 RVO agents) %s "
                           "cagym_ga3c_act (device-side selection + state vectors + fused forward, no host sync), then cagym_step_finish (k_step_post3: action maps + "
                           "dynamics, collisions + wall test, rewards, LaserScan, observations, auto-reset)" % ("on a side stream BESIDE" if args.cfg4_step == "overlap" else "in front of"))
            wp = 4 if ", 4," in env.kernel_name(rollout=False, auto_reset=True) else 5  # worlds per workgroup of this handle's specialisation
            kernel_name = "k_step_post3<256, 10, %d, true, true> (+ k_step_pre3<256, 10, %d, true>)" % (wp, wp)
        else:
            launch_mode = "per step: cagym_ga3c_act (device-side selection + state vectors + fused forward, no host sync) + cagym_step_autoreset (laser scan inside)"
            kernel_name = env.kernel_name(rollout=False, auto_reset=True)

        # --halves H > 1: worlds are independent, so the rank's worlds are dealt to H handles, each on its own HIP stream: group A's
        # GA3C forward (matrix cores) runs beside group B's env kernel (VALU / LDS, two workgroups per CU) instead of after it
        groups = [(env, ga3c, ext, None)]
        if args.halves > 1:
            H = args.halves
            env.close()
            groups = []
            for h in range(H):
                lo, hi = h * N // H, (h + 1) * N // H
                sl = np.r_[lo:hi, N + lo:N + hi]  # both pool halves of this group's worlds
                e_h = BEnv(hi - lo, M, n_scenarios=2 * (hi - lo), max_obstacles=K, laserscan=True, game_over_mode="agent0", device=device)
                e_h.set_scenarios(a6[sl], pol4[sl], scen.DYN_UNICYCLE, coop=np.full((2 * (hi - lo), M), 0.5), obstacles=ob[sl], n_obst=nob[sl])
                e_h.reset()
                groups.append((e_h, GA3C(e_h), torch.zeros((hi - lo, M, 2), dtype=torch.float32, device=device), torch.cuda.Stream(device=device)))
            env, ga3c, ext = groups[0][0], groups[0][1], groups[0][2]  # statistics / per-kernel timings: the first group (all run the same kernels)
            launch_mode += "; worlds dealt to %d handles on %d HIP streams (GA3C forward of one group beside the env kernel of another)" % (H, H)
            torch.cuda.synchronize(device)

        def one_step():
            if len(groups) == 1:
                if args.cfg4_step == "overlap":
                    env.step_overlapped(ga3c.act, ext, auto_reset=True)
                elif split:
                    env.step_begin()
                    ga3c.act(ext)
                    env.step_finish(ext, auto_reset=True)
                else:
                    ga3c.act(ext)
                    env.step(ext, auto_reset=True)
                return
            main = torch.cuda.current_stream(device)
            for e_h, g_h, x_h, st_h in groups:
                st_h.wait_stream(main)
                with torch.cuda.stream(st_h):
                    g_h.act(x_h)
                    e_h.step(x_h, auto_reset=True)
            for _e, _g, _x, st_h in groups:
                main.wait_stream(st_h)

        # the step is a chain of 4 launches + 1 memset issued from Python: captured once into a HIP graph and replayed
        # (nothing in it allocates or synchronises; same device work, no per-launch host cost)
        step_graph = None
        if args.graph:
            one_step()  # allocates the policy's workspace outside the capture
            torch.cuda.synchronize(device)
            cs = torch.cuda.Stream(device=device)
            cs.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(cs):
                one_step()
            torch.cuda.current_stream(device).wait_stream(cs)
            step_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(step_graph):
                one_step()
            launch_mode += "; the step's launches replayed from one captured HIP graph"

        def run(n_steps):
            for _ in range(n_steps):
                if step_graph is not None:
                    step_graph.replay()
                else:
                    one_step()
            gather_stats(env)
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
            n0 = env.N  # = N, or one group's worlds with --halves
            r = {"ga3c_evals_per_s": n0 / t_nn, "ga3c_ms_per_step": 1e3 * t_nn, "ga3c_network_tflops": n0 * 0.67e6 / t_nn / 1e12,
                 "ga3c_note": "cagym_ga3c_act = ONE launch (agent list, state rows, LSTM-64 + 3 x FC-256 + logits per workgroup of 32 worlds); "
                              "ga3c_network_tflops counts the network's 0.67 MFLOP per evaluation once - the kernel issues three "
                              "v_mfma_f32_32x32x16_f16 per product (fp32 operands split into two f16 halves, fp32 accumulation: "
                              "csrc/cagym_ga3c16.h; max |p - p_fp64| 2.2e-6 against 1.7e-6 for the exact-fp32 matrix-core kernel, tests/test_ga3c.py)",
                 "fused_env_kernel_ms_per_step": 1e3 * t_env, "kernels_timed_alone_on_worlds": n0,
                 "fused_env_kernel_hbm_frac": balg * n0 * M / t_env / (HBM_PEAK_GBS * 1e9)}
            if split:
                def both():
                    env.step_begin()
                    env.step_finish(ext, auto_reset=True)
                t_pre = loop(env.step_begin)
                t_both = loop(both)
                # the env part of a split step, its two launches back to back on one stream with nothing beside them
                r.update({"pre_kernel_ms": 1e3 * t_pre, "post_kernel_ms": 1e3 * (t_both - t_pre), "env_kernel_ms_per_step": 1e3 * t_both,
                          "env_kernel_hbm_frac": balg * n0 * M / t_both / (HBM_PEAK_GBS * 1e9)})
            else:
                r.update({"env_kernel_ms_per_step": 1e3 * t_env, "env_kernel_hbm_frac": r["fused_env_kernel_hbm_frac"]})
            return r
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
                    "targets, 15 NonCooperative, rectangles, OtherAgentsStates [19, 10] (BASELINE configs[4]); `value` is the env part with the IG agents driven at a "
                    "CONSTANT (v, omega) = (2 m/s, 0) - a timing workload: they run into walls, 91 %% of its episodes end in collisions -, the whole "
                    "step with the belief update and the Dec-MCTS planning step at the experiment's budget (planned actions) is `cfg5.whole_loop_env_steps_per_s`" % (N, M, " on this rank" if world_size > 1 else ""))
        launch_mode = "cagym_step_autoreset per step"
        kernel_name = env.kernel_name(rollout=False, auto_reset=True)

        def run(n_steps):
            for _ in range(n_steps):
                env.step(ext, auto_reset=True)
            gather_stats(env)
            return n_steps

        def cfg5_extra():
            """The WHOLE cfg5 step as the reference runs it (env.py:342-379 _take_action_dmcts + experiments/src/dmcts.py:50-95):
            belief update from the IG robots' poses (targetMap.update), team MI reward, one Dec-MCTS planning step at the
            experiment's budget (Ntree 30, Nsims 10, horizon 4, xdt 5, Ncycles 5, 3 robots: cagym_dmcts_plan), then the env
            step with the planned (v, omega).  Timed with synchronize brackets around `loop_steps` whole steps; the parts
            are timed again in their own loops.  Planner primitives (bare kernels) beside it."""
            dmm = importlib.import_module("gym-exploration-2d_amd.dmcts")
            ig = IG(env)
            R = 3
            planner = dmm.DeviceDecMCTSPlanner(ig, R, radius=0.5, Ntree=30, Nsims=10, horizon=4, c_p=1.0, gamma=0.95, Ncycles=5, seed=1)
            wd = torch.arange(N, dtype=torch.int32, device=device)
            det = torch.zeros((N, R, 1, 2), dtype=torch.float64, device=device)
            nd = torch.zeros((N, R), dtype=torch.int32, device=device)
            cum = torch.zeros(N, dtype=torch.float64, device=device)
            st = env.state()

            def poses_now():
                return torch.stack([st["pos_x"][:, :R], st["pos_y"][:, :R], st["heading"][:, :R]], dim=2).contiguous()

            def whole_step():
                nonlocal cum
                p = poses_now()
                obs = ig.update_belief(p, det, nd)   # detections: none (static targets are farther than 5 m almost always; Q24)
                cum = cum + ig.mi_reward(obs, wd)
                actions, _ = planner.plan(p)
                ext[:, :R] = actions.float()
                env.step(ext, auto_reset=True)
                ig.reset_belief(env.game_over)       # a restarted world starts from the prior again (targetMap.__init__)

            def loop(fn, reps=10):
                fn()
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize(device)
                return (time.perf_counter() - t0) / reps
            loop_steps = 6
            t_whole = loop(whole_step, reps=loop_steps)
            p0 = poses_now()
            t_plan = loop(lambda: planner.plan(p0), reps=3)
            t_bel = loop(lambda: ig.mi_reward(ig.update_belief(p0, det, nd), wd), reps=10)
            t_env = loop(lambda: env.step(ext, auto_reset=True), reps=50)
            # planner primitives (bare kernels; every argument already on the device)
            rng = np.random.default_rng(0)
            Q = N * 32
            # query poses with the scenario generator's own clearance rule (>= 1 m from every rectangle): the agents' start positions of
            # the world the query belongs to, cycled, with random headings (uniform poses would start a share of the roll-outs inside walls)
            world = torch.arange(Q, device=device, dtype=torch.int32) % N
            wq = np.arange(Q) % N
            ep_now = env.state()["episode"].cpu().numpy().astype(np.int64)  # the scenario each world is on now (auto-reset walks the pool)
            sc_now = (np.arange(N) + ep_now * N) % S
            starts = a6[sc_now[wq], (np.arange(Q) // N) % M, 0:2]
            poses = torch.from_numpy(np.concatenate([starts, rng.uniform(-np.pi, np.pi, (Q, 1))], 1)).to(device)
            t_vis = loop(lambda: ig.visible_cells(poses, world))
            Qr, nsims, H = N * 3, 10, 4  # experiments/src/dmcts.py budget: Nsims 10, horizon 4, xdt 5
            zeros = torch.zeros((Qr, 60), dtype=torch.int64, device=device)
            nst = torch.full((Qr,), H, dtype=torch.int32, device=device)
            rad = torch.full((Qr,), 0.5, dtype=torch.float64, device=device)
            pr, wr = poses[:Qr].contiguous(), world[:Qr].contiguous()
            t_ro = loop(lambda: ig.rollouts(pr, zeros, zeros, wr, nst, rad, nsims, 7, max_steps=H))
            grows = R * 5 * 30
            return {"whole_step_ms": 1e3 * t_whole, "whole_loop_env_steps_per_s": N / t_whole, "loop_steps_timed": loop_steps,
                    "planner_ms_per_step": 1e3 * t_plan, "belief_update_and_reward_ms": 1e3 * t_bel, "env_step_ms": 1e3 * t_env,
                    "planner_budget": "Ntree 30, Nsims 10, horizon 4, xdt 5, Ncycles 5, 3 robots per world (experiments/src/dmcts.py:31-36,74-78)",
                    "planner_tree_grows_per_s": N * grows / t_plan, "planner_rollouts_per_s": N * grows * 10 / t_plan,
                    "planner_workspace_GB": planner.workspace.numel() / 1e9,
                    "visibility_queries_per_s": Q / t_vis, "rollouts_per_s_horizon4": Qr * nsims / t_ro,
                    "rollout_visibility_queries_per_s": Qr * nsims * H / t_ro,
                    **cfg5_counters()}

        def cfg5_counters():
            c = profile_replay("cfg5_planner_pmc.json")
            src = "replayed from profiles/r4/cfg5_planner_pmc.json (rocprofv3 --pmc SQ / TCC passes of k_dmcts_plan on this workload), not measured in this run"
            return {"planner_valu_busy": c and c.get("valu_busy"), "l2_hit_rate": c and c.get("l2_hit_rate"), "counters_source": src if c else None,
                    "bound": "the planner (99.9 % of the step) is LATENCY-bound, not HBM- or MFMA-bound: a world's planning step is one serial chain of "
                             "R x Ncycles x Ntree = 450 grows, each a chain of dependent gathers through the L2-resident distance field (sphere traces of "
                             "the roll-outs' visibility queries); 8 worlds per CU x 2 waves fill the CU's 16 wave slots at 128 VGPRs"}
        extra["cfg5"] = cfg5_extra

    def barrier():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    def timed_block():
        """EXACTLY --steps steps between two barrier + synchronize brackets; returns (wall s maxed over ranks, device ms or None,
        launches).  Every SECOND block also carries the HIP events of `launch_ms_hip_events`: recording one costs the host 4 us
        (tools/host_overhead.py) - 2 % of a 20-step block - and is instrumentation, not workload, so the wall-clock figure is the
        median of the blocks WITHOUT events and the device figure the median of the blocks with them."""
        with_events = stats_state["block"] % 2 == 1
        barrier()
        if with_events:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if with_events:
            ev0.record()
        launches = run(args.steps)
        if with_events:
            ev1.record()
        # closing bracket: synchronize, read the clock, THEN the barrier - the rank's K steps are done when its own synchronize returns,
        # and the MAX over ranks below is the whole job's time; the barrier collective's own latency (tens of microseconds against a
        # 212 us block of 20 steps) is the bench's, not the workload's.  (One rank: the bracket is the synchronize alone, as before.)
        torch.cuda.synchronize(device)
        el = time.perf_counter() - t0
        if world_size > 1:
            dist.barrier()
        stats_state["block"] += 1
        if world_size > 1:
            t = torch.tensor([el], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, (ev0.elapsed_time(ev1) if with_events else None), launches

    run(args.warmup)
    barrier()
    if world_size > 1:
        # duration of the statistics collective (pack kernel + all-gather, bracketed by synchronize), median of 5, on its own
        tc = []
        for _ in range(5):
            barrier()
            t0 = time.perf_counter()
            gather_stats(env, force=True)
            barrier()
            tc.append(time.perf_counter() - t0)
        stats_state["collective_ms"] = 1e3 * sorted(tc)[2]
        stats_state["probing"] = True  # no collective inside the probe blocks below
    blocks = [timed_block()]
    if world_size > 1:
        # one block in `every` carries the collective: at most 1 % of the timed time, priced on the median of five probe
        # blocks (the first block of a run is slower than the rest); same value on every rank
        while len(blocks) < 5:
            blocks.append(timed_block())
        med = sorted(b[0] for b in blocks)[2]
        ev = max(1, int(stats_state["collective_ms"] * 1e-3 / (0.008 * max(med, 1e-6))) + 1)  # 0.8 % of the probe median: the blocks that follow are a little shorter
        r = torch.tensor([ev], dtype=torch.int64, device=device)
        dist.all_reduce(r, op=dist.ReduceOp.MAX)
        stats_state["every"] = int(r.item())
        stats_state["probing"] = False

    def more_blocks(n):
        """rank 0 decides how many more blocks to time (0 = stop); every rank runs the same count"""
        if world_size > 1:
            r = torch.tensor([n], dtype=torch.int64, device=device)
            dist.broadcast(r, 0)
            n = int(r.item())
        for _ in range(n):
            blocks.append(timed_block())
        return n

    if args.repeats > 0:
        more_blocks(max(0, args.repeats - len(blocks)))  # (a multi-rank run has timed its 5 probe blocks already)
    else:  # at least 5 blocks, and short blocks are repeated until the timed blocks cover 2 s whatever --steps is
        while len(blocks) < 50000:
            covered = sum(b[0] for b in blocks if b[1] is None)  # the wall clock is taken from the blocks without HIP events
            med = sorted(b[0] for b in blocks)[len(blocks) // 2]
            n = max(5 - len(blocks), int((2.0 - covered) / max(med, 1e-6)) + 1 if covered < 2.0 else 0)
            if more_blocks(min(n, 50000 - len(blocks))) == 0:
                break
    if world_size > 1:
        gather_stats(env, force=True)  # the final counters
        barrier()
    walls = sorted(b[0] for b in blocks if b[1] is None)  # the blocks without HIP events
    elapsed = walls[len(walls) // 2]  # median block
    dev_sorted = sorted(b[1] for b in blocks if b[1] is not None)
    dev_ms = dev_sorted[len(dev_sorted) // 2] if dev_sorted else 1e3 * elapsed  # (--repeats 1: no block with events)
    launches = blocks[0][2]

    st = stats_mod.summarize(gathered if gathered is not None else env.packed_episode_stats())
    if rank == 0:
        total_worlds = N * world_size if scaling == "weak" else total
        value = total_worlds * args.steps / elapsed
        steps_per_launch = args.steps / launches
        launch_ms_dev = dev_ms / launches            # HIP events around the block, median
        launch_ms = 1e3 * elapsed / launches         # the wall clock `value` is computed from: ONE clock for the whole line
        achieved = balg * N * M * steps_per_launch / (launch_ms * 1e-3) / 1e9
        achieved_dev = balg * N * M * steps_per_launch / (launch_ms_dev * 1e-3) / 1e9
        # HBM bytes per launch from the PMC counters are NOT collected in this run (they need rocprofv3 --pmc passes);
        # the figure below replays profiles/ (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes) for the
        # headline shape only and is labelled as such; null for any other shape.
        traffic = None
        replay = None  # counters of the headline shape, collected by the rocprofv3 --pmc passes of tools/final_profile_r3.sh
        rp = next((q for q in (os.path.join(ROOT, "profiles", r, "headline_pmc.json") for r in ("r4", "r3")) if os.path.exists(q)), None)
        if (N, M, policy, args.per_step_launch) == (4096, 10, "rvo", False) and rp:
            replay = json.load(open(rp))
        rp_name = os.path.relpath(rp, ROOT) if rp else None
        traffic_per_agent_step = (replay["hbm_bytes_per_agent_step"] + replay.get("hbm_bytes_per_agent_per_launch", 0.0) / steps_per_launch) if replay else None
        short_launch = steps_per_launch <= 32  # the driver's command: one 20-step launch per block
        if replay:
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
            "repeats": len(blocks), "timing": "wall clock: median of the %d timed blocks of %d steps without HIP events; launch_ms_hip_events: median of the %d blocks in between that carry them" % (len(walls), args.steps, len(dev_sorted)),
            "ms_per_step_min": 1e3 * walls[0] / args.steps, "ms_per_step_max": 1e3 * walls[-1] / args.steps,
            "timed_region_s": sum(walls),
            "rccl_ranks": world_size if (world_size > 1 and backend == "nccl") else 0,
            "stats_collective": None if world_size == 1 else {
                "ms": stats_state["collective_ms"], "every_blocks": stats_state["every"],
                "share_of_timed_time": stats_state["collective_ms"] * 1e-3 / (stats_state["every"] * max(elapsed, 1e-9)),
                "what": "cagym_pack_episode_stats (one kernel) + all-gather of 24-byte records on a side stream, issued in one of every_blocks timed blocks; joined only by the block's closing synchronize"},
            "collective_backend": backend if world_size > 1 else None,
            "agent_steps_per_s": value * M,
            "roofline": {"bound": "hbm", "bound_note": "the HBM roofline is the one BASELINE.json's metric asks for; the counters (valu_busy below) show the kernel is bound by VALU issue, and its HBM traffic is below the algorithmic bytes",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": None if traffic is None else
                         "replayed from %s (rocprofv3 --pmc passes on %s-step launches of this shape: FETCH_SIZE x2 + WRITE_SIZE per agent-step, plus the per-launch load / store of the agent records spread over this run's %d steps per launch), not measured in this run" % (rp_name, replay.get("steps_per_launch", "512"), int(steps_per_launch)),
                         # what actually bounds the kernel (the HBM fraction above is the metric BASELINE.json asks for): VALU issue
                         # (a 20-step launch ends with its slowest workgroup: its own counters, where the profile has them)
                         "valu_busy": None if not replay else replay.get("valu_busy_20_step_launches", replay["valu_busy"]) if short_launch else replay["valu_busy"],
                         "valu_insts_per_agent_step": None if not replay else replay.get("valu_insts_per_agent_step_20_step_launches", replay["valu_insts_per_agent_step"]) if short_launch else replay["valu_insts_per_agent_step"],
                         "valu_source": None if not replay else "replayed from %s (SQ_ACTIVE_INST_VALU, SQ_INSTS_VALU, GRBM_GUI_ACTIVE of the %s-step launches), not measured in this run" % (rp_name, "20" if short_launch and "valu_busy_20_step_launches" in replay else replay.get("steps_per_launch", "512")),
                         "kernel": kernel_name,
                         "clock": "wall-clock median block (the clock of `value` and `ms_per_step`: frac = alg_bytes_per_agent_step x agents x worlds / ms_per_step / peak)",
                         "launch_ms": launch_ms, "launch_ms_min": 1e3 * walls[0] / launches, "launch_ms_max": 1e3 * walls[-1] / launches,
                         "launch_ms_hip_events": launch_ms_dev, "achieved_hip_events": achieved_dev, "frac_hip_events": achieved_dev / HBM_PEAK_GBS,
                         "alg_bytes_per_agent_step": balg,
                         "measured_d2d_copy_GBs": copy_gbs,
                         "alg_bytes_per_launch": balg * N * M * steps_per_launch,
                         "traffic_bytes_per_launch": None if traffic is None else traffic_per_agent_step * N * M * steps_per_launch},
            "episodes": st,
        }
        for name, fn in extra.items():
            line[name] = fn()
        if args.config == "cfg4":  # the timed step holds three launches: the roofline object prices the env kernel alone
            c4 = profile_replay("cfg4_pmc.json")
            if c4:
                line["roofline"].update({"valu_busy": c4.get("valu_busy"), "valu_insts_per_agent_step": c4.get("valu_insts_per_agent_step"),
                                         "waves_per_simd": c4.get("waves_per_simd"),
                                         "valu_source": "replayed from profiles/r4/cfg4_pmc.json (rocprofv3 --pmc SQ pass of the env kernel on this workload), not measured in this run"})
            line["roofline"].update({"achieved": line["cfg4"]["env_kernel_hbm_frac"] * HBM_PEAK_GBS, "frac": line["cfg4"]["env_kernel_hbm_frac"],
                                     "launch_ms": line["cfg4"]["env_kernel_ms_per_step"], "launch_ms_min": None, "launch_ms_max": None,
                                     "note": ("env part alone: cagym_step_begin + cagym_step_finish back to back on one stream (both launches incl. laser scan), timed in its own loop"
                                              if args.cfg4_step != "fused" else "env kernel alone (cagym_step_autoreset incl. laser scan), timed in its own loop")})
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(M, pol, config=args.config)
        print(json.dumps(line), flush=True)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
