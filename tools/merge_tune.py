#!/usr/bin/env python3
"""Fold a JSON of measured (variant, split-K) choices (tools/retune_pp.py --out, or a SFAST_TUNE_CACHE file) into the packaged cache
stable-fast_amd/sfast/engine/tune_gfx950.json:   python tools/merge_tune.py gpurun_out/tune_r06_pp.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "stable-fast_amd", "sfast", "engine", "tune_gfx950.json")
pkg = json.load(open(PKG))
n = 0
for path in sys.argv[1:]:
    for k, v in json.load(open(path)).items():
        if pkg.get(k) != list(v):
            pkg[k] = list(v)
            n += 1
json.dump({k: pkg[k] for k in sorted(pkg)}, open(PKG, "w"), indent=0)
print(f"merged {n} entries -> {PKG} ({len(pkg)} problems)")
