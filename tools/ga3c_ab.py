"""A/B of the GA3C-CADRL forward kernels: CAGYM_GA3C=valu (round 1's vector kernel) vs the default matrix-core kernel.
Runs each variant in a child process (the choice is read once per process), compares probabilities / actions, prints rates.
    python tools/ga3c_ab.py            # parent
"""
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FLOP_PER_EVAL = 2.0 * (10 * 71 * 256 + 68 * 256 + 2 * 256 * 256 + 256 * 11)  # full 10-step sequence


def child(tag):
    import torch
    scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
    B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
    GA3C = importlib.import_module("gym-exploration-2d_amd.ga3c").GA3CCADRLPolicy
    out = {}
    for name, N, M in (("all_agents_81920", 8192, 10), ("agent0_8192", 8192, 10), ("all_agents_20x2048", 2048, 20)):
        pol = np.full((N, M), scen.POLICY_GA3C if "all" in name else scen.POLICY_RVO, dtype=np.int32)
        pol[:, 0] = scen.POLICY_GA3C
        env = B(N, M, game_over_mode="agent0")
        env.set_scenarios(scen.random_worlds_fast(N, M, seed=5), pol, scen.DYN_UNICYCLE, coop=np.full((N, M), 0.5))
        env.reset()
        p = GA3C(env)
        ext = torch.zeros((N, M, 2), dtype=torch.float32, device=env.device)
        for _ in range(3):  # a few steps so that velocities / headings are not all zero
            p.act(ext)
            env.step(ext)
        p.states()
        act, probs = p.forward(want_probs=True)
        torch.cuda.synchronize()
        np.savez(f"/tmp/ga3c_ab_{tag}_{name}.npz", act=act.cpu().numpy(), probs=probs.cpu().numpy())
        Bn = int(act.numel())
        nseq = p.state.reshape(-1, 76)[p.agent_index().long(), 1].float().mean().item()
        idx = p.agent_index()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def launch():  # the bare C-ABI call: no allocation, no index refresh
            rc = p.L.cagym_ga3c_forward(env.h, p.blob.data_ptr(), p.state.data_ptr(), idx.data_ptr(), Bn, None, act.data_ptr(), None, env._stream())
            assert rc == 0

        for _ in range(5):
            launch()
        reps = 100
        e0.record()
        for _ in range(reps):
            launch()
        e1.record()
        torch.cuda.synchronize()
        dt = e0.elapsed_time(e1) * 1e-3 / reps
        out[name] = {"evals": Bn, "us": dt * 1e6, "evals_per_s": Bn / dt, "mean_seq_len": nseq,
                     "tflops_full_seq_equiv": Bn * FLOP_PER_EVAL / dt / 1e12}
        env.close()
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
        sys.exit(0)
    res = {}
    variants = {"valu": {"CAGYM_GA3C": "valu"}, "mfma": {}}
    for extra in os.environ.get("GA3C_AB_LIBS", "").split(","):  # tag=path of diagnostic builds (tools/build_alt.sh)
        if "=" in extra:
            variants[extra.split("=")[0]] = {"CAGYM_LIB": os.path.abspath(extra.split("=")[1])}
    for tag, extra in variants.items():
        env = dict(os.environ)
        env.pop("CAGYM_GA3C", None)
        env.update(extra)
        r = subprocess.run([sys.executable, __file__, tag], env=env, capture_output=True, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print(r.stdout[-2000:], r.stderr[-4000:])
            sys.exit(1)
        res[tag] = json.loads(line[0][7:])
    for tag in variants:
        if tag == "valu":
            continue
        for name in res["valu"]:
            a, b = np.load(f"/tmp/ga3c_ab_valu_{name}.npz"), np.load(f"/tmp/ga3c_ab_{tag}_{name}.npz")
            res.setdefault("compare_" + tag, {})[name] = {"max_abs_prob_diff": float(np.abs(a["probs"] - b["probs"]).max()),
                                                          "action_mismatches": int((a["act"] != b["act"]).sum()), "n": int(a["act"].size)}
    for tag in variants:
        print(tag, {k: (round(v["us"], 1), round(v["evals_per_s"] / 1e6, 1), round(v["tflops_full_seq_equiv"], 1)) for k, v in res[tag].items()})
    print(json.dumps({k: v for k, v in res.items() if k.startswith("compare")}))
