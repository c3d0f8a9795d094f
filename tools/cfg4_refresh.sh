set -e
O=gpurun_out/s7
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; tail -2 $O/gputests.log
python bench.py --config cfg4 --steps 200 --warmup 50 > $O/bench_cfg4.json 2> $O/bench_cfg4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfg4 -o run -- python3 bench.py --config cfg4 --steps 200 --warmup 50 --repeats 5 --no-cpu-baseline > $O/bench_cfg4_under_rocprof.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
head -6 $O/stats_cfg4/run_kernel_stats.csv | cut -c1-160
python - <<PY
import json
for f in ("bench_cfg4","bench_driver_cmd"):
    d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("cfg4"))
PY
tail -1 $O/smoke.log
