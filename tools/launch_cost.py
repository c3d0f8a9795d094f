#!/usr/bin/env python3
"""Diagnostic: where does the fixed cost of a cagym_rollout launch go?  (never the shipped library)

Part A (shipped library): HIP-event time of R back-to-back launches of n steps each, n = 1 .. 128, and of ONE isolated
launch (sync before, events around it): per-launch cost without and with the host gap.
Part B (-DCAGYM_WGTRACE build): every workgroup stamps the 100 MHz s_memrealtime clock at entry, after the prologue, after
each step and at exit: dispatch ramp, per-step time of the first steps vs later ones, spread of the workgroups' end times.
"""
import ctypes
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("gym-exploration-2d_amd.build")
import torch

N, M = int(os.environ.get("LC_WORLDS", 4096)), int(os.environ.get("LC_AGENTS", 10))


def make_env():
    scen = importlib.import_module("gym-exploration-2d_amd.scenarios")
    B = importlib.import_module("gym-exploration-2d_amd.batched_env").BatchedCollisionAvoidanceEnv
    env = B(N, M, n_scenarios=8 * N, game_over_mode="all")
    env.set_scenarios(scen.random_worlds_fast(8 * N, M, seed=1234), scen.POLICY_RVO, scen.DYN_UNICYCLE, coop=np.full((8 * N, M), 0.5))
    env.reset()
    return env


def part_a():
    env = make_env()
    traj = env.alloc_rollout(128)
    env.rollout(128, out=traj)
    env.rollout(128, out=traj)
    torch.cuda.synchronize()
    print("Part A: %d worlds x %d agents, shipped library" % (N, M))
    print("  n_steps   back-to-back us/launch   us/step    isolated us/launch (median of 9)")
    for n in (1, 2, 5, 10, 20, 40, 64, 128):
        R = 40
        env.rollout(n, out=traj)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(R):
            env.rollout(n, out=traj)
        e1.record()
        torch.cuda.synchronize()
        bb = e0.elapsed_time(e1) * 1e3 / R
        iso = []
        for _ in range(9):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            env.rollout(n, out=traj)
            e1.record()
            torch.cuda.synchronize()
            iso.append(e0.elapsed_time(e1) * 1e3)
        print("  %6d   %12.1f            %7.2f    %10.1f" % (n, bb, bb / n, float(np.median(iso))))
    del env, traj


WG_LIB = os.path.join(b.CSRC, "libcagym_hip_wgtrace.so")


def build_wgtrace():
    assert b.build_variant("wgtrace", ["-DCAGYM_WGTRACE"]) == WG_LIB  # one monolithic diagnostic unit, cached (build.py)


def part_b():
    lib = WG_LIB
    assert os.environ.get("CAGYM_LIB") == lib, "part B runs in its own process with CAGYM_LIB set before the package loads"
    env = make_env()
    traj = env.alloc_rollout(64)
    for _ in range(4):
        env.rollout(64, out=traj)
    torch.cuda.synchronize()
    W = 48
    n_wg = (N + 3) // 4 if M == 10 and (N + 3) // 4 <= 5 * 256 else (N + 4) // 5 if M == 10 else 0
    if not n_wg:
        print("Part B handles M = 10 only")
        return
    n_wg = min(n_wg, 4096)
    buf = (ctypes.c_ulonglong * (W * n_wg))()
    print("Part B: per-workgroup s_memrealtime trace (10 ns ticks), %d workgroups" % n_wg)
    for n, warm in ((20, True), (20, False), (36, True), (5, True)):
        if not warm:  # warm: the traced launch is queued right behind another one (no idle gap in front of it)
            torch.cuda.synchronize()
        ep0 = env.state()["episode"].clone()
        if warm:
            env.rollout(64, out=traj)
            ep0 = None  # (episode counters of the launch in front are not separable; the sync'd variant reports resets)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.rollout(n, out=traj)
        e1.record()
        torch.cuda.synchronize()
        ev_us = e0.elapsed_time(e1) * 1e3
        env.L.cagym_debug_wgtrace(buf, n_wg)
        resets = None
        if ep0 is not None:
            wpw = (N + n_wg - 1) // n_wg
            d = (env.state()["episode"] - ep0).cpu().numpy().astype(np.float64)
            d = np.concatenate([d, np.zeros(n_wg * wpw - d.size)])
            resets = d.reshape(n_wg, wpw).sum(1)
        T = np.frombuffer(buf, dtype=np.uint64).reshape(n_wg, W).astype(np.int64)
        t0 = T[:, 0].min()
        start, pro, end = (T[:, 0] - t0) * 0.01, (T[:, 1] - T[:, 0]) * 0.01, (T[:, 38] - t0) * 0.01
        steps = np.diff(T[:, 1:2 + n], axis=1) * 0.01  # [n_wg, n] us
        xcc = T[:, 39] & 15
        busy = (T[:, 39] >> 8).astype(np.float64)  # busy egos (LP groups) summed over the launch's steps
        print("  n_steps %d, %s: event %.1f us; kernel span (first entry -> last exit) %.1f us"
              % (n, "behind a running launch" if warm else "after a sync (idle GPU)", ev_us, end.max()))
        print("    workgroup entry: min %.1f  median %.1f  p99 %.1f  max %.1f us;  prologue median %.2f max %.2f us"
              % (start.min(), np.median(start), np.percentile(start, 99), start.max(), np.median(pro), pro.max()))
        print("    workgroup exit : min %.1f  median %.1f  p99 %.1f  max %.1f us;  own duration median %.1f max %.1f us"
              % (end.min(), np.median(end), np.percentile(end, 99), end.max(), np.median(end - start), (end - start).max()))
        print("    step time, median over workgroups: " + " ".join("%.1f" % v for v in np.median(steps, axis=0)))
        print("    step time, max over workgroups   : " + " ".join("%.1f" % v for v in steps.max(axis=0)))
        tot = steps.sum(axis=1)
        h = n // 2
        print("    persistence: corr(first-half time, second-half time) over workgroups %.3f; corr(total time, busy-ego count) %.3f; "
              "busy per step: min %.1f median %.1f max %.1f; least-squares time = %.2f + %.3f x busy us per step"
              % (np.corrcoef(steps[:, :h].sum(1), steps[:, h:].sum(1))[0, 1], np.corrcoef(tot, busy)[0, 1], busy.min() / n,
                 np.median(busy) / n, busy.max() / n, *np.polyfit(busy / n, tot / n, 1)[::-1]))
        if resets is not None:
            A = np.stack([np.ones(n_wg), busy / n, resets], 1)
            coef = np.linalg.lstsq(A, tot, rcond=None)[0]
            print("    auto-resets per workgroup in this launch: mean %.2f max %d; least squares: workgroup time = %.1f + %.2f x (busy egos per step) + %.2f x resets us; "
                  "workgroups by resets (count: median time): %s" % (resets.mean(), resets.max(), coef[0], coef[1], coef[2],
                  "  ".join("%d (%d): %.1f" % (r, (resets == r).sum(), np.median(tot[resets == r])) for r in sorted(set(resets.astype(int).tolist())))))
        if n == 20 and warm:
            np.save(os.path.join(ROOT, "gpurun_out", "wgtrace_20.npy"), T)
        print("    sum of steps per workgroup: min %.1f median %.1f p99 %.1f max %.1f us; XCC ids seen %s, entry median per XCC %s"
              % (tot.min(), np.median(tot), np.percentile(tot, 99), tot.max(), sorted(set(xcc.tolist())),
                 " ".join("%.1f" % np.median(start[xcc == x]) for x in sorted(set(xcc.tolist())))))


if __name__ == "__main__":
    if "--part-b" in sys.argv:  # child process: CAGYM_LIB was set by the parent before this interpreter started
        part_b()
    else:
        if "--b-only" not in sys.argv:
            part_a()
        if "--a-only" not in sys.argv:
            build_wgtrace()
            sys.stdout.flush()
            sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), "--part-b"], env=dict(os.environ, CAGYM_LIB=WG_LIB)))
