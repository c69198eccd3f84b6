#!/bin/bash
# HBM-side traffic per kernel symbol of the bench step: FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# (counters only with --kernel-trace, as the pool requires) over the SAME bench command, autotune results cached.
# usage: tools/gpu_pmc_bench.sh [sd15|sdxl|svd|bs8] [steps]   -> gpurun_out/pmcb/traffic_by_symbol[_<config>].json
# (bs8 = SD1.5 at 8 images per GPU, the per-GPU shape of BASELINE configs[3]: `bench.py --config sd15 --images 8`)
# The file carries a `_meta` block (round, commit = $SFAST_COMMIT as passed by the caller -- the GPU box has no .git --, sha256 of the
# packaged tune cache, the command): bench.py quotes `traffic` from it only while the kernel choices are the ones it was taken with.
cd "$(dirname "$0")/.."
CFG=${1:-sd15}
STEPS=${2:-6}
SUF=""; [ "$CFG" != "sd15" ] && SUF="_$CFG"
BARGS="--config $CFG"; [ "$CFG" = "bs8" ] && BARGS="--config sd15 --images 8"
BARGS="$BARGS --preheat-seconds 0"   # (bench.py replays the step for 3 s before its warm-up by default: thousands of dispatches under counters)
# SVD-XT: 930 launches per step and two graph shapes calibrated with 26 replays each = 53 k dispatches per pass, ~15 ms each under counters
[ "$CFG" = "svd" ] && export SFAST_GRAPH_CALIBRATE=0
export BARGS
rm -rf gpurun_out/pmcb; mkdir -p gpurun_out/pmcb
export TMPDIR=/tmp
R=$PWD
# round 5: the passes run on the PACKAGED kernel choices (sfast/engine/tune_gfx950.json), exactly like the driver's `python bench.py`, and
# WITHOUT the variants: the literal B = 1 step is not in the packaged cache, its autotuning launches (thousands, under counters) flooded
# the first round-5 pass and the "last 40 %" window (profiles/r05_pmc_traffic_contaminated_run6.log); the SDXL pass timed out on them
# pass 0: --kernel-trace ONLY (no counters): the per-dispatch durations `rocprofv3 --stats` shows. Kernels run ~10-15 % slower while counters
# are collected, so the durations of the --pmc passes (`avg_us`) are not the ones a roofline fraction may be priced with: `avg_us_trace` is.
( cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/pmcb -o trace -- python $R/bench.py $BARGS --steps $STEPS --warmup 2 --no-cpu-baseline --no-roofline --no-end-to-end --no-variants > $R/gpurun_out/pmcb/trace.log 2>&1 )
echo "trace pass exit=$?"
for db in $(find gpurun_out/pmcb -name "*.db"); do python tools/rocpd_summary.py $db --csv gpurun_out/pmcb/trace.csv --top 200 --step-marker cfg_ddim --steps $(( STEPS - 1 )) > gpurun_out/pmcb/trace.txt; rm -f $db; done
pass() { # name, counters...
  local name=$1; shift
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmcb -o $name -- python $R/bench.py $BARGS --steps $STEPS --warmup 2 --no-cpu-baseline --no-roofline --no-end-to-end --no-variants > $R/gpurun_out/pmcb/$name.log 2>&1 )
  echo "pmc $name exit=$? $(tail -n 1 $R/gpurun_out/pmcb/$name.log | cut -c1-100)"
  for db in $(find $R/gpurun_out/pmcb -name "*.db"); do python $R/tools/pmc_extract.py $db $R/gpurun_out/pmcb/$name.json --by-symbol --last-frac 0.4; rm -f $db; done
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
SUF=$SUF CFG=$CFG STEPS=$STEPS python - <<'PY'
import hashlib, json, os
f = {r["kernel"]: r for r in json.load(open("gpurun_out/pmcb/fetch.json"))["rows"]}
w = {r["kernel"]: r for r in json.load(open("gpurun_out/pmcb/write.json"))["rows"]}
import csv
trace = {}
try:
    with open("gpurun_out/pmcb/trace.csv") as fh:
        for row in csv.DictReader(l for l in fh if not l.startswith("#")):
            trace[row["kernel"]] = (float(row["avg_us"]), int(row["calls"]))
except OSError:
    pass
out = {}
for k, r in f.items():
    fetch_kb = r["counters"].get("FETCH_SIZE", 0.0)
    write_kb = w.get(k, {}).get("counters", {}).get("WRITE_SIZE", 0.0)
    # gfx950: FETCH_SIZE counts 128-B fabric requests as 64 B (MI355X_MICROARCH.md, HBM section) -> x2; unit KB
    out[k] = dict(bytes_per_launch=(2.0 * fetch_kb + write_kb) * 1024.0, fetch_kb_raw=fetch_kb, write_kb_raw=write_kb,
                  launches=r["dispatches"], avg_us=r.get("avg_us"))
    t = trace.get(k[:110])   # tools/rocpd_summary.py cuts names at 110 characters
    if t:
        out[k]["avg_us_trace"], out[k]["launches_trace"] = t
sha = hashlib.sha256(open("stable-fast_amd/sfast/engine/tune_gfx950.json", "rb").read()).hexdigest()[:16]
out["_meta"] = dict(round=6, commit=os.environ.get("SFAST_COMMIT", "unknown"), tune_cache_sha256=sha,
                    command=f"python bench.py {os.environ['BARGS']} --steps {os.environ['STEPS']} --warmup 2 --no-cpu-baseline --no-roofline --no-end-to-end --no-variants",
                    method="separate rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE; WRITE_SIZE); last 40 % of the dispatches; bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB; "
                           "avg_us = per-dispatch average UNDER counter collection, avg_us_trace = the same from a --kernel-trace-only pass (steady window of graph replays)")
json.dump(out, open("gpurun_out/pmcb/traffic_by_symbol" + os.environ.get("SUF", "") + ".json", "w"), indent=1)
del out["_meta"]
for k, v in sorted(out.items(), key=lambda kv: -(kv[1]["avg_us"] or 0) * kv[1]["launches"])[:12]:
    print(f"{k[:90]:90s} n={v['launches']:5d} {v['avg_us'] or 0:7.1f} us  {v['bytes_per_launch'] / 1e6:8.2f} MB/launch")
PY
