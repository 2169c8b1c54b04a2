#!/bin/bash
# Scratch: wall time of the default bench.py run (what the driver launches) and of --steps 1 --warmup 0.
mkdir -p gpurun_out
t0=$(date +%s.%N)
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || exit 1
t1=$(date +%s.%N)
timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bench_s1.json 2> gpurun_out/bench_s1.err || exit 1
t2=$(date +%s.%N)
python - <<PY
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print("default run: wall %.1f s" % ($t1 - $t0), {k: d[k] for k in ("value", "ms_per_step", "steps", "warmup", "world_size", "collectives_per_step")})
print("  frac", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["value"], "clock", d["calibration"].get("k1_core_clock_mhz"), "hard eig share", d["hard_spectrum"]["eig_share"])
d = json.loads(open("gpurun_out/bench_s1.json").read().strip().splitlines()[-1])
print("steps 1 warmup 0: wall %.1f s" % ($t2 - $t1), {k: d[k] for k in ("value", "ms_per_step", "steps", "warmup")}, d["roofline"]["frac"])
PY
