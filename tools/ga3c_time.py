"""HIP-event timing of cagym_ga3c_act (selection + state vectors + forward) on the library CAGYM_LIB names (default: in-tree).
    python tools/ga3c_time.py [tag]
Cases: cfg4's composition (8192 worlds x 10, agent 0 GA3C), every slot GA3C (81 920 evaluations)."""
import importlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    tag = sys.argv[1] if len(sys.argv) > 1 else "default"
    scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
    B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
    GA3C = importlib.import_module("gym-exploration-2d_amd.ga3c").GA3CCADRLPolicy
    out = {"tag": tag}
    for name, N, M in (("agent0_8192", 8192, 10), ("all_agents_81920", 8192, 10)):
        pol = np.full((N, M), scen.POLICY_GA3C if "all" in name else scen.POLICY_RVO, dtype=np.int32)
        pol[:, 0] = scen.POLICY_GA3C
        env = B(N, M, game_over_mode="agent0")
        env.set_scenarios(scen.random_worlds_fast(N, M, seed=5), pol, scen.DYN_UNICYCLE, coop=np.full((N, M), 0.5))
        env.reset()
        p = GA3C(env)
        ext = torch.zeros((N, M, 2), dtype=torch.float32, device=env.device)
        for _ in range(3):
            p.act(ext)
            env.step(ext)
        torch.cuda.synchronize()
        ts = []
        for rep in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                p.act(ext)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 50 * 1e3)
        out[name + "_us"] = round(float(np.median(ts)), 2)
        out[name + "_checksum"] = float(ext.double().sum().item())
        env.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
