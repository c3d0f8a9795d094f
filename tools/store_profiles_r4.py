#!/usr/bin/env python3
"""Copy the outputs of tools/final_profile_r4.sh (gpurun_out/final_r4) into profiles/r4/ with their summaries, and write the
counter files bench.py replays (headline_pmc.json, cfg4_pmc.json, cfg5_planner_pmc.json).  Parts that have not been run yet are skipped."""
import csv, glob, json, os, shutil
O, P = "gpurun_out/final_r4/", "profiles/r4/"
os.makedirs(P, exist_ok=True)


def have(*names):
    return all(os.path.exists(O + n) for n in names)


def cp(src, dst=None):
    if os.path.exists(O + src):
        shutil.copy(O + src, P + (dst or os.path.basename(src)))
        return True
    return False


def first(pattern):
    g = glob.glob(O + pattern, recursive=True)
    return g[0][len(O):] if g else None


def avg(path, key):
    for l in open(path):
        if l.strip().startswith(key):
            return float(l.split()[-1])


def dur(path):
    for l in open(path):
        if "avg_ns=" in l:
            return float(l.split("avg_ns=")[1])


def kstat(path, name):
    for r in csv.reader(open(path)):
        if r and name in r[0]:
            return r
    return None


def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])


# ---- part a: tests, headline lines, kernel stats, counters -------------------------------------------------------------------------
for a in ("gputests.log", "smoke.log", "bench_default.json", "bench_driver_cmd.json", "bench_under_rocprof.json", "bench20_under_rocprof.json"):
    cp(a)
ks512 = first("stats/**/*kernel_stats.csv")
ks20 = first("stats20/**/*kernel_stats.csv")
if ks512:
    cp(ks512, "bench_4096x10_rvo_kernel_stats.csv")
if ks20:
    cp(ks20, "bench_driver_cmd_20steps_kernel_stats.csv")
if have("pmc_fetch.txt", "pmc_write.txt", "pmc_sq_4096.txt", "pmc_sq_20.txt") and ks512 and ks20:
    A = 4096 * 10
    f, w = avg(O + "pmc_fetch.txt", "FETCH_SIZE"), avg(O + "pmc_write.txt", "WRITE_SIZE")
    tot512 = (2 * f + w) * 1024  # FETCH_SIZE counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md): x2; KB -> B
    per_launch = None
    if have("pmc_fetch20.txt", "pmc_write20.txt"):
        f20, w20 = avg(O + "pmc_fetch20.txt", "FETCH_SIZE"), avg(O + "pmc_write20.txt", "WRITE_SIZE")
        tot20 = (2 * f20 + w20) * 1024
        # bytes(T) = per_step * A * T + per_launch * A  from the 512-step and the 20-step passes
        per_step = (tot512 - tot20) / (A * (512 - 20))
        per_launch = max(0.0, (tot20 - per_step * A * 20) / A)
    else:
        per_step = tot512 / (A * 512)
    k512, k20 = kstat(O + ks512, "k_rollout3"), kstat(O + ks20, "k_rollout3")
    d0, d20 = json.load(open(O + "bench_default.json")), json.load(open(O + "bench_driver_cmd.json"))
    open(P + "bench_4096x10_rvo_pmc_hbm.txt", "w").write(
        "kernel %s (round 4), 4096 worlds x 10 agents; rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE | WRITE_SIZE (own passes) -- python3 bench.py --steps 1024 --warmup 512 --repeats 5 (512-step launches)\n"
        "and the same two passes on the driver's command (--steps 20 --warmup 5 --repeats 200: 20-step launches); per-dispatch averages by tools/pmc_summary.py\n"
        "512-step launches: FETCH_SIZE %.0f KB (x2: gfx950 counts 64 B per 128-B request) + WRITE_SIZE %.4g KB = %.1f MB per dispatch = %.1f B per agent-step\n"
        "%s"
        "algorithmic bytes: 517 B per agent-step = %.1f MB per 512-step dispatch\n"
        "kernel duration, --kernel-trace --stats: %.4f ms average over %s launches of 512 steps (bench_4096x10_rvo_kernel_stats.csv); bench.py HIP events %.4f ms (bench_default.json)\n"
        "driver command: %.4f ms average over %s launches of 20 steps (bench_driver_cmd_20steps_kernel_stats.csv); bench.py HIP events %.4f ms, wall %.4f ms (bench_driver_cmd.json)\n" % (
            d0["roofline"]["kernel"], f, w, tot512 / 1e6, tot512 / (A * 512),
            "" if per_launch is None else "20-step launches: %.1f MB per dispatch; two-point fit bytes(T) = %.1f B x agent-steps + %.1f B x agents per launch (the load and store of the agent records)\n" % (tot20 / 1e6, per_step, per_launch),
            517.0 * A * 512 / 1e6, float(k512[3]) / 1e6, k512[1], d0["roofline"]["launch_ms_hip_events"],
            float(k20[3]) / 1e6, k20[1], d20["roofline"]["launch_ms_hip_events"], d20["roofline"]["launch_ms"]))
    txt = ["rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS -- python3 bench.py --steps 1024 --warmup 512 --repeats 5 | --steps 20 --warmup 5 --repeats 200",
           "per-dispatch averages; SQ_ACTIVE_INST_VALU in quad-cycles summed over SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs"]
    pmc = {}
    for tag, steps in (("4096", 512), ("20", 20)):
        p = O + "pmc_sq_%s.txt" % tag
        txt += ["--- 4096 worlds, %d steps per launch" % steps, open(p).read().rstrip()]
        a, g = avg(p, "SQ_ACTIVE_INST_VALU"), avg(p, "GRBM_GUI_ACTIVE")
        pmc[steps] = (4 * a / (1024 * g / 8), avg(p, "SQ_INSTS_VALU") / (A * steps))
        txt.append("VALU busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) = %.0f %%; wave VALU instructions per agent-step = %.1f; VALU cycles per wave instruction = %.2f"
                   % (100 * pmc[steps][0], pmc[steps][1], 4 * a / avg(p, "SQ_INSTS_VALU")))
    open(P + "bench_rvo_pmc_sq.txt", "w").write("\n".join(txt) + "\n")
    json.dump({"hbm_bytes_per_agent_step": per_step, "hbm_bytes_per_agent_per_launch": per_launch or 0.0, "valu_busy": pmc[512][0], "valu_insts_per_agent_step": pmc[512][1],
               "valu_busy_20_step_launches": pmc[20][0], "valu_insts_per_agent_step_20_step_launches": pmc[20][1], "steps_per_launch": 512,
               "source": "tools/final_profile_r4.sh a: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (own passes, FETCH x2 gfx950 correction) on 512-step and 20-step launches of 4096 x 10 (two-point fit: per agent-step + per agent per launch), and the SQ passes"},
              open(P + "headline_pmc.json", "w"), indent=1)
    print("headline: %.1f B/agent-step + %s B/agent/launch, VALU busy %.3f / %.3f" % (per_step, per_launch, pmc[512][0], pmc[20][0]))

# ---- part b: sweeps, rehearsal, launch cost -----------------------------------------------------------------------------------------
if have("sweep.txt"):
    out = ["N-sweep and secondary configurations, round-4 kernels (tools/final_profile_r4.sh b; bench.py flags as listed)",
           "columns: env-steps/s (median timed block), agent-steps/s, roofline.frac (alg bytes per agent-step / 8 TB/s), launch ms, kernel"]
    for l in open(O + "sweep.txt"):
        l = l.strip()
        if l.startswith("{"):
            d = json.loads(l)
            out.append("   %.1f M env-steps/s  %.2f G agent-steps/s  frac %.4f  launch %.3f ms  %s" % (
                d["value"] / 1e6, d["agent_steps_per_s"] / 1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], d["roofline"]["kernel"]))
        elif l:
            out.append(l)
    out.append("flags: N-sweep = --worlds N --roll R --steps 512 --warmup 128 --repeats 5 --pool-factor 2 --scenarios device; roll sweep = --roll R --repeats 5;")
    out.append("       per-step = --per-step-launch --steps 1024 --warmup 128; cfg2 = --config cfg2; 2048x20 = --worlds 2048 --agents 20 --roll 256")
    open(P + "sweep.txt", "w").write("\n".join(out) + "\n")
if have("rehearsal_2ranks_gloo_driver_cmd.json"):
    open(P + "rehearsal_2ranks_one_gpu_gloo_driver_cmd.json", "w").write(json.dumps(last_json(O + "rehearsal_2ranks_gloo_driver_cmd.json")) + "\n")
cp("launch_cost.txt")
cp("valu_breakdown.txt")

# ---- part c: cfg4 / cfg5 ----------------------------------------------------------------------------------------------------------------
for a in ("bench_cfg4.json", "bench_cfg4_split.json", "bench_cfg4_overlap.json", "bench_cfg5.json", "cfg4_timeline.txt", "cfg4_pmc.txt", "dmcts_phases.txt"):
    cp(a)
k4 = first("stats_cfg4/**/*kernel_stats.csv")
if k4:
    cp(k4, "bench_cfg4_kernel_stats.csv")
if have("cfg4_pmc/p1.txt"):
    p = O + "cfg4_pmc/p1.txt"
    a, g, n = avg(p, "SQ_ACTIVE_INST_VALU"), avg(p, "GRBM_GUI_ACTIVE"), avg(p, "SQ_INSTS_VALU")
    wc, bc = avg(p, "SQ_WAVE_CYCLES"), avg(p, "SQ_BUSY_CU_CYCLES")
    json.dump({"valu_busy": 4 * a / (1024 * g / 8), "valu_insts_per_agent_step": n / (8192 * 10), "waves_per_simd": (4 * wc / (1024 * g / 8)) if wc else None,  # SQ_WAVE_CYCLES counts quad-cycles per resident wave
               "kernel_avg_us": dur(p) / 1e3, "source": "tools/cfg4_pmc.sh (rocprofv3 --pmc SQ pass of bench.py --config cfg4), env kernel k_step3<256, 10, 4, true, true>"},
              open(P + "cfg4_pmc.json", "w"), indent=1)
if have("cfg5_prof/pmc_summary.txt"):
    cp("cfg5_prof/pmc_summary.txt", "cfg5_planner_pmc.txt")
    k5 = first("cfg5_prof/stats/**/*kernel_stats.csv")
    if k5:
        cp(k5, "bench_cfg5_kernel_stats.csv")
    vals = {}
    sec = None
    for l in open(O + "cfg5_prof/pmc_summary.txt"):
        if l.startswith("== "):
            sec = l[3:].strip()
        elif sec == "k_dmcts_plan" and l.startswith("  "):
            k, v = l.split()
            vals[k] = float(v)
    if "SQ_ACTIVE_INST_VALU" in vals and "GRBM_GUI_ACTIVE" in vals:
        hit, miss = vals.get("TCC_HIT_sum"), vals.get("TCC_MISS_sum")
        json.dump({"valu_busy": 4 * vals["SQ_ACTIVE_INST_VALU"] / (1024 * vals["GRBM_GUI_ACTIVE"] / 8),
                   "l2_hit_rate": (hit / (hit + miss)) if hit is not None and miss is not None and hit + miss > 0 else None,
                   "valu_insts_per_vmem_read": vals["SQ_INSTS_VALU"] / vals["SQ_INSTS_VMEM_RD"] if vals.get("SQ_INSTS_VMEM_RD") else None,
                   "source": "tools/cfg5_profile.sh (rocprofv3 --pmc SQ and TCC passes of bench.py --config cfg5), kernel k_dmcts_plan"},
                  open(P + "cfg5_planner_pmc.json", "w"), indent=1)
print("stored:", sorted(os.listdir(P)))
