#!/bin/bash
# GPU-box job: VALU utilisation counters + host cost of the per-launch episode-stats packing
set -e
mkdir -p gpurun_out/r1b
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PMC="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU"
rocprofv3 --kernel-trace --output-format csv --pmc $PMC -d gpurun_out/r1b/pmc_valu_4096 -o run -- python3 bench.py --steps 128 --warmup 64 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc $PMC -d gpurun_out/r1b/pmc_valu_65536 -o run -- python3 bench.py --steps 128 --warmup 64 --no-cpu-baseline --worlds 65536 --pool-factor 2 > /dev/null 2>&1
python tools/pmc_summary.py gpurun_out/r1b/pmc_valu_4096 k_rollout2 > gpurun_out/r1b/pmc_valu_4096.txt
python tools/pmc_summary.py gpurun_out/r1b/pmc_valu_65536 k_rollout2 > gpurun_out/r1b/pmc_valu_65536.txt
cat gpurun_out/r1b/pmc_valu_4096.txt gpurun_out/r1b/pmc_valu_65536.txt
python - <<'PY'
import importlib, time, numpy as np, torch
scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
stats = importlib.import_module("gym-exploration-2d_amd.stats")
B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
env = B(4096, 10, n_scenarios=8192, game_over_mode="all")
env.set_scenarios(scen.random_worlds_fast(8192, 10, seed=1), scen.POLICY_RVO, scen.DYN_UNICYCLE)
env.reset()
traj = env.alloc_rollout(64)
for _ in range(3):
    env.rollout(64, auto_reset=True, out=traj)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    p = stats.pack_episode_stats(env.episode_stats())
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("pack_episode_stats host %.1f us/call, with device drain %.1f us/call" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
t0 = time.perf_counter()
for _ in range(50):
    env.rollout(64, auto_reset=True, out=traj)
    p = stats.pack_episode_stats(env.episode_stats())
torch.cuda.synchronize()
print("rollout+pack %.3f ms/launch" % ((time.perf_counter() - t0) / 50 * 1e3))
PY
