#!/bin/bash
# Fresh measured kernel choices for every configuration (packaged cache ignored), TWICE; a choice replaces the packaged one only where both
# fresh runs agree on it (per-problem timing noise flips near-ties); then a same-box A/B of the step: packaged choices vs the merged file.
# usage: tools/retune_full.sh   -> gpurun_out/tune_full/{a,b}_{sd15,bs8,sdxl,svd}.json, merged_tune.json, ab.log
O=gpurun_out/tune_full; rm -rf $O; mkdir -p $O
COMMON="--no-cpu-baseline --no-end-to-end --no-roofline --no-sdxl-variant"
args() { case $1 in sd15) echo "--config sd15";; bs8) echo "--config sd15 --images 8 --no-variants";; sdxl) echo "--config sdxl --no-variants";; svd) echo "--config svd --no-variants";; esac; }
for pass in a b; do
  for tag in sd15 bs8 sdxl svd; do
    SFAST_TUNE_PACKAGED=0 SFAST_TUNE_CACHE=$O/${pass}_$tag.json timeout -k 10 1500 python bench.py $(args $tag) --steps 10 --warmup 3 $COMMON > $O/tune_${pass}_$tag.log 2>&1
  done
done
python - <<'PY'
import json, glob
pkg = json.load(open("stable-fast_amd/sfast/engine/tune_gfx950.json"))
out = dict(pkg); n = new = 0
for tag in ("sd15", "bs8", "sdxl", "svd"):
    a = json.load(open(f"gpurun_out/tune_full/a_{tag}.json")); b = json.load(open(f"gpurun_out/tune_full/b_{tag}.json"))
    for k, v in a.items():
        if b.get(k) == v and v[0] > 0:
            if k not in pkg: new += 1
            elif pkg[k] != v: n += 1; print(f"  {k}: {pkg[k]} -> {v}")
            out[k] = v
json.dump({k: out[k] for k in sorted(out)}, open("gpurun_out/tune_full/merged_tune.json", "w"), indent=0)
print(f"{n} packaged choices replaced, {new} problems added, {len(out)} problems")
PY
val() { python -c "import json,sys
for l in sys.stdin:
    if l.startswith('{'): print(round(json.loads(l)['value'],3))"; }
for rep in 1 2; do
  for tag in sd15 bs8 sdxl svd; do
    A="$(args $tag) --no-variants --no-cpu-baseline --no-end-to-end --no-roofline"
    [ $tag = svd ] && A="$A --steps 30 --warmup 5"
    p=$(timeout -k 10 600 python bench.py $A 2>/dev/null | val)
    cp $O/merged_tune.json /tmp/merged_ro.json
    f=$(SFAST_TUNE_PACKAGED=0 SFAST_TUNE_CACHE=/tmp/merged_ro.json timeout -k 10 600 python bench.py $A 2>/dev/null | val)
    echo "rep $rep $tag packaged $p merged $f" | tee -a $O/ab.log
  done
done
