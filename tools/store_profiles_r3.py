#!/usr/bin/env python3
"""Copy the outputs of tools/final_profile_r3.sh (gpurun_out/final_r3) into profiles/r3/ with their summaries."""
import csv, json, os, shutil
O, P = "gpurun_out/final_r3/", "profiles/r3/"
os.makedirs(P, exist_ok=True)
shutil.copy(O + "stats/run_kernel_stats.csv", P + "bench_4096x10_rvo_kernel_stats.csv")
shutil.copy(O + "stats20/run_kernel_stats.csv", P + "bench_driver_cmd_20steps_kernel_stats.csv")
shutil.copy(O + "stats_cfg4/run_kernel_stats.csv", P + "bench_cfg4_kernel_stats.csv")
for a in ("bench_default.json", "bench_driver_cmd.json", "bench_under_rocprof.json", "bench20_under_rocprof.json", "bench_cfg4.json",
          "bench_cfg5.json", "wave_trace_4096.txt", "launch_cost.txt", "slow_workgroup_wave_trace.txt", "smoke.log", "cfg4_outcomes.txt", "valu_breakdown.txt", "cfg4_timeline.txt", "cfg4_pmc.txt"):
    shutil.copy(O + a, P + a)
# the 2-rank rehearsal prints gloo warnings on stderr only; keep the JSON line alone
open(P + "rehearsal_2ranks_one_gpu_gloo.json", "w").write(
    [l for l in open(O + "rehearsal_2ranks_gloo.json").read().splitlines() if l.startswith("{")][-1] + "\n")  # gloo greets on stdout
out = ["N-sweep and secondary configurations, final round-3 kernels (tools/final_profile_r3.sh; bench.py flags as listed)",
       "columns: env-steps/s (median of 5 timed blocks), agent-steps/s, roofline.frac (alg bytes per agent-step / 8 TB/s), launch ms, kernel"]
for l in open(O + "sweep.txt"):
    l = l.strip()
    if l.startswith("{"):
        d = json.loads(l)
        out.append("   %.1f M env-steps/s  %.2f G agent-steps/s  frac %.4f  launch %.3f ms  %s" % (
            d["value"] / 1e6, d["agent_steps_per_s"] / 1e9, d["roofline"]["frac"], d["roofline"]["launch_ms"], d["roofline"]["kernel"]))
    else:
        out.append(l)
out.append("flags: N-sweep = --worlds N --roll R --steps 512 --warmup 128 --repeats 5 --pool-factor 2 --scenarios device; roll sweep = --roll R --repeats 5;")
out.append("       per-step = --per-step-launch --steps 1024 --warmup 128; cfg2 = --config cfg2; 2048x20 = --worlds 2048 --agents 20 --roll 256")
d0 = json.load(open(O + "bench_default.json"))
out.append("measured device-to-device copy ceiling (1 GiB torch copy, read + write counted): %.1f TB/s" % (d0["roofline"]["measured_d2d_copy_GBs"] / 1e3))
open(P + "sweep.txt", "w").write("\n".join(out) + "\n")


def avg(path, key):
    for l in open(path):
        if key in l:
            return float(l.split()[-1])


def dur(path):
    for l in open(path):
        if "avg_ns=" in l:
            return float(l.split("avg_ns=")[1])


def kstat(path, name):
    for r in csv.reader(open(path)):
        if r and r[0].startswith(name):
            return r
    return None


f, w = avg(O + "pmc_fetch.txt", "FETCH_SIZE"), avg(O + "pmc_write.txt", "WRITE_SIZE")
units = 512 * 4096 * 10
tot = (2 * f + w) * 1024
ks = kstat(O + "stats/run_kernel_stats.csv", "void k_rollout3")
ks20 = kstat(O + "stats20/run_kernel_stats.csv", "void k_rollout3")
d20 = json.load(open(O + "bench_driver_cmd.json"))
open(P + "bench_4096x10_rvo_pmc_hbm.txt", "w").write(
    "kernel %s (round-3 final), 512 env steps x 4096 worlds x 10 agents per dispatch (%d agent-steps)\n"
    "rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -- python3 bench.py --steps 1024 --warmup 512 --repeats 5 --no-cpu-baseline   (own pass)\n"
    "rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -- python3 bench.py --steps 1024 --warmup 512 --repeats 5 --no-cpu-baseline   (own pass)\n"
    "(tools/final_profile_r3.sh; per-dispatch averages by tools/pmc_summary.py)\n"
    "FETCH_SIZE avg/dispatch = %.0f KB -> x2 (gfx950 counts 64 B per 128-B request) = %.2f MB\n"
    "WRITE_SIZE avg/dispatch = %.4g KB = %.1f MB\n"
    "traffic = %.1f MB/dispatch = %.1f B per agent-step; algorithmic bytes (517 B/agent-step) = %.2f MB/dispatch\n"
    "kernel duration in the two passes: %.3f / %.3f ms; --kernel-trace --stats run: %.4f ms average over %s launches of 512 steps\n"
    "(bench_4096x10_rvo_kernel_stats.csv); bench.py's own HIP-event figure in the same session: %.3f ms per launch (bench_default.json)\n"
    "driver command (--steps 20 --warmup 5): --kernel-trace --stats average %.4f ms over %s launches of 20 steps\n"
    "(bench_driver_cmd_20steps_kernel_stats.csv); bench.py's HIP-event median %.4f ms (bench_driver_cmd.json)\n" % (
        d0["roofline"]["kernel"], units, f, 2 * f * 1024 / 1e6, w, w * 1024 / 1e6, tot / 1e6, tot / units, 517.0 * units / 1e6,
        dur(O + "pmc_fetch.txt") / 1e6, dur(O + "pmc_write.txt") / 1e6, float(ks[3]) / 1e6, ks[1], d0["roofline"]["launch_ms_hip_events"],
        float(ks20[3]) / 1e6, ks20[1], d20["roofline"]["launch_ms_hip_events"]))
print("traffic B/agent-step %.1f" % (tot / units))
txt = ["rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS -- python3 bench.py --steps 1024 --warmup 512 --repeats 5 --no-cpu-baseline | ... --steps 20 --warmup 5 --repeats 200",
       "per-dispatch averages (512 / 20 env steps per dispatch at 4096 worlds); SQ_ACTIVE_INST_VALU in quad-cycles summed over SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs"]
pmc = {}
for tag, steps in (("4096", 512), ("20", 20)):
    p = O + "pmc_sq_%s.txt" % tag
    txt.append("--- 4096 worlds, %d steps per launch" % steps)
    txt.append(open(p).read().rstrip())
    a, g = avg(p, "SQ_ACTIVE_INST_VALU"), avg(p, "GRBM_GUI_ACTIVE")
    txt.append("VALU busy = 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) = %.0f %%" % (100 * 4 * a / (1024 * g / 8)))
    txt.append("wave VALU instructions per agent-step = %.1f" % (avg(p, "SQ_INSTS_VALU") / (4096 * 10 * steps)))
    pmc[steps] = (4 * a / (1024 * g / 8), avg(p, "SQ_INSTS_VALU") / (4096 * 10 * steps))
open(P + "bench_rvo_pmc_sq.txt", "w").write("\n".join(txt) + "\n")

# the counters bench.py replays in its roofline object (headline shape only)
per_step = tot / units
json.dump({"hbm_bytes_per_agent_step": per_step, "hbm_bytes_per_agent_per_launch": 0.0, "valu_busy": pmc[512][0], "valu_insts_per_agent_step": pmc[512][1],
           "valu_busy_20_step_launches": pmc[20][0], "valu_insts_per_agent_step_20_step_launches": pmc[20][1], "steps_per_launch": 512,
           "source": "tools/final_profile_r3.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (own passes, FETCH x2 gfx950 correction) and the SQ pass, 512-step launches of 4096 x 10"},
          open(P + "headline_pmc.json", "w"), indent=1)
# cfg5 planner counters, 2-rank rehearsal with the driver's command
shutil.copy(O + "cfg5_prof/pmc_summary.txt", P + "cfg5_planner_pmc.txt")
shutil.copy(O + "cfg5_prof/stats/run_kernel_stats.csv", P + "bench_cfg5_kernel_stats.csv")
open(P + "rehearsal_2ranks_one_gpu_gloo_driver_cmd.json", "w").write(
    [l for l in open(O + "rehearsal_2ranks_gloo_driver_cmd.json").read().splitlines() if l.startswith("{")][-1] + "\n")
