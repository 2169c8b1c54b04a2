#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel stats + PMC passes of the default
# bench.py workload.  Raw output goes to gpurun_out/prof/, scripts/summarize_profiles.py turns
# it into the small JSON / CSV files committed under profiles/.
#   gpurun --timeout 1100 -- 'bash scripts/collect_profiles.sh r2'
set -eo pipefail
TAG=${1:-r2}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
# (the full-width CPU sample is a ~200 s gesdd: off here, the profile run has to fit one gpurun call)
export DMDX_BENCH_CPU_FULL=0
BENCH="python3 bench.py --steps 3 --warmup 1"
echo "[prof] plain bench"; $BENCH > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "[prof] kernel stats"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" --output-format csv -- $BENCH --no-cpu-baseline --no-calibrate --no-hard-spectrum > "$OUT/stats.log" 2>&1
# counters in their own runs, with --kernel-trace only (separate passes: FETCH_SIZE and
# WRITE_SIZE do not fit in one; the SQ set is one pass)
PB="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-calibrate --no-hard-spectrum"
RP="timeout -k 10 150 rocprofv3"   # a profiler pass that stops making progress must not eat the GPU budget
# (round 2: the FETCH_SIZE pass once sat for its whole limit before the program had printed anything,
# and took 6 s when repeated: a pass that fails is retried once and never aborts the collection)
pass() { "$@" || { echo "[prof] pass failed, retrying once: $*"; "$@" || echo "[prof] pass failed twice: $*"; }; }
echo "[prof] pmc fetch"
pass $RP --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" --output-format csv -- $PB > "$OUT/pmc_fetch.log" 2>&1
echo "[prof] pmc write"
pass $RP --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmc_write" --output-format csv -- $PB > "$OUT/pmc_write.log" 2>&1
echo "[prof] pmc sq"
pass $RP --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d "$OUT/pmc_sq" --output-format csv -- $PB > "$OUT/pmc_sq.log" 2>&1
echo "[prof] randomized path (K2 / K3 tall-skinny GEMMs), events + kernel stats"
python3 scripts/bench_randomized.py > "$OUT/bench_randomized.json" 2> "$OUT/bench_randomized.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_rand" --output-format csv -- python3 scripts/bench_randomized.py --steps 1 > "$OUT/stats_rand.log" 2>&1
echo "[prof] gap-free (power-law) spectrum: the same step, kernel stats of the eigen stage"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_powerlaw" --output-format csv -- python3 bench.py --steps 3 --warmup 1 --spectrum powerlaw --no-cpu-baseline --no-calibrate > "$OUT/stats_powerlaw.log" 2>&1
echo "[prof] cfg4 (227.6 GB resident, randomized, 2 power iterations): kernel stats at rank 50 / 200, HBM traffic at rank 50"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_cfg4_k50" --output-format csv -- python3 scripts/bench_cfg4.py --k 50 --quick > "$OUT/stats_cfg4_k50.log" 2>&1 || echo "[prof] cfg4 k50 stats failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats_cfg4_k200" --output-format csv -- python3 scripts/bench_cfg4.py --k 200 --quick > "$OUT/stats_cfg4_k200.log" 2>&1 || echo "[prof] cfg4 k200 stats failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_cfg4_k50" --output-format csv -- python3 scripts/bench_cfg4.py --k 50 --quick > "$OUT/pmc_cfg4_k50.log" 2>&1 || echo "[prof] cfg4 k50 pmc failed"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d "$OUT/pmc_cfg4_sq" --output-format csv -- python3 scripts/bench_cfg4.py --k 50 --quick > "$OUT/pmc_cfg4_sq.log" 2>&1 || echo "[prof] cfg4 k50 sq pmc failed"
echo "[prof] summarize"
python3 scripts/summarize_profiles.py "$OUT" "$ROOT/gpurun_out/profiles_$TAG" "$TAG"
# the raw traces are large: keep only the small summaries + logs in gpurun_out/
find "$OUT" -name "*.csv" -size +20M -delete
echo "[prof] done"
