#!/usr/bin/env python3
"""Re-time the packaged kernel choices (sfast/engine/tune_gfx950.json) against the 256-row ping-pong tiles of round 6 (igemm_pp.h).

    SFAST_TUNE_EXTEND=1 is set here: every cached problem of the plans below whose 256-row tiling puts tiles on >= 96 CUs is timed once
    more -- its cached (variant, split-K) against variants 51 - 56 -- and the winner is written to --out (a JSON of ONLY the
    problems this run touched, same format as the packaged file). tools/merge_tune.py folds it into the packaged cache.

    python tools/retune_pp.py --out gpurun_out/tune_r06_pp.json [--configs sd15:2,sd15:16,sdxl:2,svd:2,vae:1]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-fast_amd"))
os.environ["SFAST_TUNE_EXTEND"] = "1"
import torch  # noqa: E402

from sfast.engine import autotune  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--configs", default="sd15:2,sd15:16,sdxl:2,svd:2,vae:1,vae:8")
    a = ap.parse_args()
    dev = torch.device("cuda")
    from sfast.engine import SVDUNetEngine, UNet2DEngine, VaeDecoderEngine
    from sfast.engine.unet_spec import (SD15_CONFIG, SD_VAE_DECODER_CONFIG, SDXL_CONFIG, SVD_CONFIG, random_params, random_svd_params,
                                        random_vae_decoder_params)
    before = autotune.export_cache()
    for item in a.configs.split(","):
        name, b = item.split(":")
        b = int(b)
        if name in ("sd15", "sdxl"):
            cfg = SD15_CONFIG if name == "sd15" else SDXL_CONFIG
            eng = UNet2DEngine(cfg, random_params(cfg, seed=0, dtype=torch.float16, device=dev))
            hw = cfg["sample_size"]
            eng.get_plan(b, hw, hw, 77)
        elif name == "svd":
            eng = SVDUNetEngine(SVD_CONFIG, random_svd_params(SVD_CONFIG, seed=0, dtype=torch.float16, device=dev))
            eng.get_plan(b, 25, 72, 128)
        elif name == "vae":
            eng = VaeDecoderEngine(SD_VAE_DECODER_CONFIG, random_vae_decoder_params(SD_VAE_DECODER_CONFIG, seed=0, dtype=torch.float16, device=dev))
            eng.get_plan(b, 64, 64)
        torch.cuda.synchronize()
        now = autotune.export_cache()
        changed = {k: v for k, v in now.items() if before.get(k) != v}
        print(f"[retune_pp] {item}: {len(autotune._extended)} problems re-timed so far, {len(changed)} choices changed", flush=True)
        del eng
        torch.cuda.empty_cache()
    now = autotune.export_cache()
    changed = {k: v for k, v in now.items() if before.get(k) != v}
    with open(a.out, "w") as f:
        json.dump({k: list(v) for k, v in sorted(changed.items())}, f, indent=0)
    for k, v in sorted(changed.items()):
        print(f"  {k}: {before.get(k)} -> {v}")


if __name__ == "__main__":
    main()
