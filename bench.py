#!/usr/bin/env python3
"""Benchmark of the hot path: SD1.5 512x512 bs=1 fp16 denoise iterations per second.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one denoise iteration exactly as the reference counts "it/s" (BASELINE.md section 1): one
classifier-free-guidance batch-2 UNet forward + guidance combine + DDIM update, replayed as ONE
hipGraph. Synthetic latents / text embeddings, seeded random-init weights of the SD1.5 architecture
(no checkpoints or network here). Inputs are resident in HBM when the timed region starts.
With N > 1 every rank owns a full replica (weights broadcast once from rank 0 over RCCL/xGMI) and
runs its own independent loop on its own image: weak scaling, value = N * K / max-over-ranks time.

Rank 0 prints ONE JSON line. Besides the contract keys it carries
  "roofline"     for the dominant kernel of the step (measured live with HIP events on the launch stream)
  "cpu_baseline" the fp32 oracle UNet (restatement of diffusers; diffusers itself is not installable
                 here) timed on the host cores for the same CFG batch-2 forward, rank 0 at N=1 only.
"""
import argparse
import json
import re
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "stable-fast_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

# (round 6: no `_KEEP_GRAPHS` list any more -- every graph the product hands out is an OwnedGraph whose teardown is deferred to a
# synchronised point, sfast/engine/unet2d.py; the raw torch.cuda.CUDAGraph objects this script captures itself are retired the same way)

MFMA_PEAK_TFLOPS = 2500.0  # dense f16/bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=None,
                    help="untimed steps before the timed ones. Default 150 (SVD-XT, 0.2 s per step: 20) = three 50-step images, the warm-up of the reference's own benchmark "
                         "protocol (3 warm-up pipeline calls: SURVEY.md section 8d; /root/reference/src/sfast/cuda/graphs.py:87-92 warms a graph "
                         "up with 3 calls too). With 20 (rounds 1 - 6) the SD1.5 line read 1.2 %% below its steady state (191.1 vs 193.4 it/s, "
                         "twice each in one session; 200 / 1000 / 2000 timed steps behind 200 / 1000 warm-up steps all read 193.4 - 193.6): "
                         "the chip needs about a second of sustained load to reach its steady clocks")
    ap.add_argument("--preheat-seconds", type=float, default=None,
                    help="sd15 / sdxl: replay the step untimed for this long before the warm-up steps (default 3.0; 0 = off): the chip takes "
                         "seconds of sustained load to reach its steady clocks, reported as `preheat_seconds` in the line")
    ap.add_argument("--config", default="sd15", choices=["sd15", "sdxl", "vae", "svd"],
                    help="sd15 (the BASELINE metric) | sdxl | vae (SD VAE decode 64x64 latent -> 512x512, SURVEY 8f rank 1) | "
                         "svd (SVD-XT 576x1024, 25 frames, BASELINE configs[4])")
    ap.add_argument("--frames", type=int, default=25, help="--config svd: frames per video")
    ap.add_argument("--images", type=int, default=1, help="images per GPU (UNet batch is 2x this: CFG)")
    ap.add_argument("--weights", default=os.environ.get("SFAST_SD15_DIR") or None,
                    help="diffusers model directory (or unet .safetensors file) to load REAL UNet weights from (sd15 / sdxl configs; default: "
                         "$SFAST_SD15_DIR, else seeded random-init weights -- there is no checkpoint on the driver's box)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--through-compile", action="store_true",
                    help="also time the same step through sfast.compilers.compile() + a pipeline-shaped loop (N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the end-to-end ms/image leg (text encoder + 50 steps + VAE decode + post-process; sd15 only)")
    ap.add_argument("--e2e-images", type=int, default=3, help="images timed by the end-to-end leg (after one warm-up image)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the extra lines of the same run: text-K/V hoisted out of the step graph, the literal B = 1 (no-CFG) step, "
                         "and (N = 8 or --bs64-sharded) BASELINE configs[3]: bs = 64 sharded over the ranks")
    ap.add_argument("--no-sdxl-variant", action="store_true",
                    help="skip `variants.sdxl` of the default N = 1 SD1.5 run: the 1x4x128x128-latent step (BASELINE configs[2]) timed by a child "
                         "process of this script (bench.py --config sdxl --steps 200), so that the driver-run record carries both latent sizes")
    ap.add_argument("--bs64-sharded", action="store_true",
                    help="run the configs[3] leg (64 images split over the N ranks, 64 / N per GPU) for any N, not only N = 8")
    ap.add_argument("--dump-kernels", default=None, help="write the per-op timing table to this JSON file")
    a = ap.parse_args()
    if a.preheat_seconds is None:
        a.preheat_seconds = 3.0
    if a.warmup is None:
        a.warmup = 20 if a.config == "svd" else 150
    return a


def per_op_timing(loop, reps=2, burst=4):  # noqa: C901
    """Eager replay of the step with HIP events on the launch stream. Every op is launched `burst` times back to
    back between one event pair (the pair's own ~2-3 us of record overhead would otherwise be charged to every
    5-15 us kernel), `reps` rounds; seconds = mean per launch. Runs after the timed region: repeating in-place
    residual ops leaves garbage in the activation pool, which nothing reads afterwards."""
    from sfast.hip import lib as L
    plan = loop.plan
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream
    n = len(plan.ops)
    tot = [0.0] * n
    names = [None] * n
    for _ in range(reps):
        evs = []
        for i, op in enumerate(plan.ops):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(burst):
                op.launch(sp)
            b.record(stream)
            names[i] = L.last_kernel()
            evs.append((a, b))
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(evs):
            tot[i] += a.elapsed_time(b) * 1e-3 / burst
    rows = []
    for i, op in enumerate(plan.ops):
        rows.append(dict(kind=op.kind, name=op.name, kernel=names[i], seconds=tot[i] / reps, flops=op.flops, bytes=op.bytes))
    return rows


def in_situ_timing(plan, idxs, reps=3):
    """Duration of the launches `idxs` of the plan INSIDE the step: the whole plan runs eagerly in program order, one launch per op,
    and only the chosen launches are bracketed by HIP events on the launch stream. This is what the roofline object quotes: the
    launch sees the caches its predecessor left behind, as in the graph replay and in the rocprofv3 trace of this command (a burst
    of identical launches re-reads its own inputs from L2 and reads ~20 % low). The interval of an EMPTY event pair is measured and
    reported, not subtracted. Against the trace of the same session the raw intervals read ~10-15 % low on the long launches (self-
    attention 77-80 us vs 83-87 us) and high on the 10 us ones; the trace shows the same per-dispatch durations for graph and eager
    replays (profiles/r02_kernel_stats_{graph,eager}_replay_run15.txt), so the difference is how a marker-packet interval and a
    dispatch's begin/end timestamps delimit a kernel, not a different execution."""
    stream = torch.cuda.current_stream()
    sp = stream.cuda_stream
    want = set(idxs)
    cal = []
    for _ in range(32):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        b.record(stream)
        cal.append((a, b))
    torch.cuda.synchronize()
    overhead = sorted(a.elapsed_time(b) for a, b in cal)[len(cal) // 2] * 1e-3
    tot = {i: 0.0 for i in want}
    for _ in range(reps):
        evs = {}
        for i, op in enumerate(plan.ops):
            if i in want:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                op.launch(sp)
                b.record(stream)
                evs[i] = (a, b)
            else:
                op.launch(sp)
        torch.cuda.synchronize()
        for i, (a, b) in evs.items():
            tot[i] += a.elapsed_time(b) * 1e-3
    return {i: t / reps for i, t in tot.items()}, overhead


def differential_graph_timing(plan, idxs, replays=30, rounds=3):
    """What the replayed hipGraph loses when the launches `idxs` are left out: the step is captured twice on one stream -- complete,
    and without those launches -- and HIP events around `replays` back-to-back replays of each give (t_full - t_without) per
    step. ROCm refuses timing events inside a captured graph ("External events are disallowed in rocm",
    tools/graph_event_probe.py), so this is the only event-based view from inside the graph; it is an EXCLUSIVE time (a kernel's
    ramp-up and drain overlap its neighbours' and do not come back when it is removed) and reads ~20 % below the rocprofv3
    per-dispatch durations. Values computed by the graph without the launches are garbage; nothing reads them."""
    s = torch.cuda.Stream()
    skip = set(idxs)
    graphs = []
    with torch.cuda.stream(s):
        for leave_out in (False, True):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                sp = torch.cuda.current_stream().cuda_stream
                for i, op in enumerate(plan.ops):
                    if leave_out and i in skip:
                        continue
                    op.launch(sp)
            graphs.append(g)
        for g in graphs:
            for _ in range(3):
                g.replay()
        torch.cuda.synchronize()
        best = [None, None]
        for _ in range(rounds):
            for k, g in enumerate(graphs):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(s)
                for _ in range(replays):
                    g.replay()
                b.record(s)
                b.synchronize()
                t = a.elapsed_time(b) * 1e-3 / replays
                best[k] = t if best[k] is None or t < best[k] else best[k]
    torch.cuda.current_stream().wait_stream(s)
    from sfast.engine.unet2d import retire_graph
    for g in graphs:  # never destroyed in the same breath as the last replay
        retire_graph(g, None)
    return max(best[0] - best[1], 0.0), best[0], best[1]


_IGEMM_WAVES = {("128x128", False): (2, 2), ("128x160", False): (4, 1), ("64x64", False): (2, 2), ("64x160", False): (2, 1),
                ("128x64", False): (2, 2), ("64x128", False): (2, 2),
                ("256x128", False): (4, 2), ("128x128", True): (2, 2), ("64x128", True): (2, 2)}


def kernel_symbol(variant):
    """Device symbol (as rocprofv3 / profiles/*.csv print it) of a library kernel-variant string; split-K launches
    of one tile shape share a symbol, so the split factor is dropped. A trailing "+staged" / "+gnstats" marks the
    instantiation with LDS-staged stores (last template argument)."""
    import re
    mp = re.match(r"igemm_(conv|lin)_(f16|bf16)(_geglu)?\[256x(\d+),split=(\d+),pp([wl]?)(\d)\](\+staged|\+gnstats)?", variant)
    if mp:
        # igemm_pp_kernel<T, BN, NS, KSP, PW, MODE, GEGLU, STAGED, EXP = 0> (csrc/igemm_pp.h): "ppw" = four producer waves (KSP = 2), "ppl" =
        # producers + lockstep consumers (KSP = 0), "pp" = the 8-wave ping-pong (KSP = 4; 2 for the 256-wide tile)
        bn, pw = int(mp.group(4)), 4 if mp.group(6) else 0
        ksp = 0 if mp.group(6) == "l" else 2 if (pw or bn >= 256) else 4
        t = "DF16_" if mp.group(2) == "f16" else "DF16b"
        staged = int(mp.group(8) is not None and int(mp.group(5)) == 1)
        return (f"_ZN5sfast15igemm_pp_kernelI{t}Li{bn}ELi{mp.group(7)}ELi{ksp}ELi{pw}ELi{1 if mp.group(1) == 'conv' else 0}ELb{int(mp.group(3) is not None)}"
                f"ELb{staged}ELi0EEEvNS_9IgemmArgsE")
    m = re.match(r"igemm_(conv|lin)_(f16|bf16)(_geglu)?\[(\d+x\d+),split=(\d+),(reg|dma(\d)|ws(\d)|pk(\d))\](\+staged|\+gnstats)?(\+join)?", variant)
    if not m:
        ma = re.match(r"attn_fwd\[D=(\d+),BQ=(\d+)\](\+bias)?", variant)
        if ma:  # attn_fwd_kernel<T, D, NW, TRACE = 0, BIAS = false>: one wave per 32 queries
            return f"_ZN5sfast15attn_fwd_kernelIDF16_Li{ma.group(1)}ELi{int(ma.group(2)) // 32}ELi0ELb{int(ma.group(3) is not None)}EEEvNS_8AttnArgsE"
        return variant.split("[")[0]
    mode = 1 if m.group(1) == "conv" else 0
    t = "DF16_" if m.group(2) == "f16" else "DF16b"
    geglu = m.group(3) is not None
    bm, bn = m.group(4).split("x")
    # staged stores (and statistics from the tile flush) only when the GEMM kernel itself finishes the tile: unsplit, or split-K joined
    # inside the kernel ("+join"); otherwise a split GEMM writes fp32 slabs and splitk_reduce[_rows]_kernel finishes (igemm.hip igemm_run)
    staged = int(m.group(10) is not None and (int(m.group(5)) == 1 or m.group(11) is not None))
    if m.group(6).startswith("pk"):
        # igemm_pk_kernel<T, BM, BN, WN, NS, PD, MODE, STAGED> (csrc/igemm_pk.h): WN = 5 for the 160- / 320-row tiles, PD = 2 when a
        # wave holds two 32-row weight blocks
        wn_pk = 5 if int(bn) in (160, 320) else 4
        pd = 2 if int(bn) // (wn_pk * 32) >= 2 else 3
        return f"_ZN5sfast15igemm_pk_kernelI{t}Li{bm}ELi{bn}ELi{wn_pk}ELi{m.group(9)}ELi{pd}ELi{mode}ELb{staged}EEEvNS_9IgemmArgsE"
    wm, wn = _IGEMM_WAVES[(m.group(4), geglu)]
    if m.group(6) == "reg":
        return f"_ZN5sfast12igemm_kernelI{t}Li{bm}ELi{bn}ELi{wm}ELi{wn}ELi{mode}ELb{int(geglu)}ELb{staged}EEEvNS_9IgemmArgsE"
    if m.group(6).startswith("ws"):
        return (f"_ZN5sfast20igemm_glds_ws_kernelI{t}Li{bm}ELi{bn}ELi{wm}ELi{wn}ELi4ELi{m.group(8)}ELi{mode}ELb{int(geglu)}ELb{staged}ELi0EEE"
                "vNS_9IgemmArgsE")
    return f"_ZN5sfast17igemm_glds_kernelI{t}Li{bm}ELi{bn}ELi{wm}ELi{wn}ELi{m.group(7)}ELi{mode}ELb{int(geglu)}ELi0ELb0EEEvNS_9IgemmArgsE"


def roofline_from(rows, plan=None, config="sd15"):
    by_kernel = {}
    for i, r in enumerate(rows):
        sym = kernel_symbol(r["kernel"])
        k = by_kernel.setdefault(sym, dict(seconds=0.0, flops=0.0, bytes=0.0, launches=0, kinds={}, variants=set(), idx=[]))
        k["idx"].append(i)
        k["seconds"] += r["seconds"]
        k["flops"] += r["flops"]
        k["bytes"] += r["bytes"]
        k["launches"] += 1
        k["kinds"][r["kind"]] = k["kinds"].get(r["kind"], 0) + 1
        k["variants"].add(r["kernel"])
    total = sum(v["seconds"] for v in by_kernel.values())
    # dominant kernel = the device symbol with the largest share of the step (what `rocprofv3 --stats` ranks first)
    dom_name, dom = max(by_kernel.items(), key=lambda kv: kv[1]["seconds"])
    share = dom["seconds"] / total
    timing = "HIP events around bursts of 4 identical launches (eager replay of the plan)"
    extra = {}
    if plan is not None:
        situ, overhead = in_situ_timing(plan, dom["idx"])
        dom = dict(dom, seconds=sum(situ.values()))
        timing = (f"HIP events around each of this symbol's launches inside an eager in-order replay of the whole step, 3 rounds; raw "
                  f"intervals (an empty event pair measures {overhead * 1e6:.1f} us, not subtracted)")
        try:  # second opinion from inside the replayed graph: what the step loses when these launches are left out
            dt, t_full, t_wo = differential_graph_timing(plan, dom["idx"])
            extra = dict(in_graph_exclusive_us=dt / len(dom["idx"]) * 1e6,
                         in_graph_method=(f"difference of 30 back-to-back replays of the step as a serial hipGraph ({t_full * 1e3:.3f} ms) and of the "
                                          f"same graph without this symbol's {len(dom['idx'])} launches ({t_wo * 1e3:.3f} ms), best of 3 rounds"))
        except RuntimeError:
            pass
    split_note = {}
    if any(re.search(r"split=([2-9]|\d\d)", v) for v in dom["variants"]):
        split_note = dict(interval_includes="the split-K reduce launch (splitk_reduce[_rows]_kernel, 5-10 us) that follows each launch of this "
                                            "symbol: one C-ABI call, one event interval; the rocprofv3 per-symbol average excludes it")
    mfma = dom["flops"] > 0 and (dom["flops"] / MFMA_PEAK_TFLOPS / 1e12) > (dom["bytes"] / HBM_PEAK_GBS / 1e9)
    if mfma:
        achieved = dom["flops"] / dom["seconds"] / 1e12
        roof = dict(bound="mfma", achieved=achieved, peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=achieved / MFMA_PEAK_TFLOPS)
    else:
        achieved = dom["bytes"] / dom["seconds"] / 1e9
        roof = dict(bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS)
    roof.update(kernel=dom_name, variants=sorted(dom["variants"])[:6], op_kinds=dom["kinds"],
                launches_per_step=dom["launches"], avg_launch_us=dom["seconds"] / dom["launches"] * 1e6,
                share_of_step=share, timing=timing, **extra, **split_note, algorithmic_gflop_per_launch=dom["flops"] / dom["launches"] / 1e9,
                algorithmic_mbytes_per_launch=dom["bytes"] / dom["launches"] / 1e6, traffic=None)
    # Two named fractions so that the line is reproducible from profiles/ alone:
    #  * frac_interval    -- live, this run: algorithmic work / HIP-event interval around the C-ABI call. For a split-K problem the call is
    #                        TWO launches (main kernel + splitk_reduce*), and the interval also holds the event pair's own overhead.
    #  * frac_kernel_only -- algorithmic work / the rocprofv3 per-dispatch average of THIS symbol, from the committed counter / trace passes
    #                        over this same command (profiles/rNN_pmc_traffic_by_symbol*.json: `avg_us`), i.e. what `rocprofv3 --stats` shows.
    # `frac` / `achieved` stay the live interval figures (the conservative ones).
    roof["frac_interval"] = roof["frac"]
    roof["frac_kernel_only"] = None
    # HBM-side bytes per launch from the PMC counters: collected in their own rocprofv3 --pmc passes over this same command
    # (tools/gpu_pmc_bench.sh: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KB units) and committed under profiles/. A file is used only when its
    # `_meta` block says it was taken with the kernel choices in use now (sha256 of the packaged tune cache); otherwise `traffic` is null
    # and `traffic_note` says why (round 4 silently quoted a round-3 file).
    try:
        import glob
        import hashlib
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.environ.get("SFAST_TRAFFIC_PROFILE")
        if not path:
            # counters belong to ONE bench command: the SD1.5 default has the plain name, other configs their own suffix
            suffix = "" if config == "sd15" else f"_{config}"
            cands = sorted(glob.glob(os.path.join(here, "profiles", f"r[0-9][0-9]_pmc_traffic_by_symbol{suffix}.json")))
            path = cands[-1] if cands else None
        if path:
            with open(path) as f:
                doc = json.load(f)
            meta = doc.get("_meta") or {}
            with open(os.path.join(here, "stable-fast_amd", "sfast", "engine", "tune_gfx950.json"), "rb") as f:
                tune_sha = hashlib.sha256(f.read()).hexdigest()[:16]
            rel = os.path.relpath(path, here)
            t = doc.get(dom_name)
            if not meta:
                roof["traffic_note"] = f"{rel} carries no _meta block (taken before round 5): not quoted"
            elif meta.get("tune_cache_sha256") != tune_sha:
                roof["traffic_note"] = (f"{rel} (round {meta.get('round')}, commit {meta.get('commit')}) was taken with other kernel choices "
                                        f"(tune cache {meta.get('tune_cache_sha256')} != {tune_sha}): not quoted")
            elif not t:
                roof["traffic_note"] = f"{rel} has no row for this symbol"
            else:
                roof["traffic"] = t["bytes_per_launch"]
                roof["traffic_over_algorithmic"] = t["bytes_per_launch"] / max(dom["bytes"] / dom["launches"], 1.0)
                roof["traffic_source"] = (f"{rel} (round {meta.get('round')}, commit {meta.get('commit')}, {meta.get('command')}; counters from "
                                          "separate rocprofv3 --pmc passes over this command, not re-measured in this run; FETCH_SIZE x2 + "
                                          "WRITE_SIZE, includes Infinity-Cache hits)")
                # the --kernel-trace-only pass of the same session (kernels run ~10-15 % slower while counters are collected: `avg_us` of the
                # --pmc passes is not a duration to price a roofline fraction with)
                if t.get("avg_us_trace"):
                    per_launch = (dom["flops"] if mfma else dom["bytes"]) / dom["launches"]
                    rate = per_launch / (t["avg_us_trace"] * 1e-6) / (1e12 if mfma else 1e9)
                    roof["frac_kernel_only"] = rate / (MFMA_PEAK_TFLOPS if mfma else HBM_PEAK_GBS)
                    roof["kernel_only_avg_us"] = t["avg_us_trace"]
                    roof["kernel_only_source"] = (f"{rel}: rocprofv3 --kernel-trace (no counters) per-dispatch average of this symbol over "
                                                  f"{t.get('launches_trace')} dispatches of the steady window of graph replays, "
                                                  + (f"durations from {meta['avg_us_trace_from']}" if meta.get("avg_us_trace_from")
                                                     else "same session as the counters"))
    except (OSError, ValueError, KeyError) as e:
        roof["traffic_note"] = f"traffic file unreadable: {type(e).__name__}: {e}"
    fam = {}
    for r in rows:
        f = fam.setdefault(r["kind"], dict(seconds=0.0, flops=0.0, bytes=0.0, launches=0))
        f["seconds"] += r["seconds"]
        f["flops"] += r["flops"]
        f["bytes"] += r["bytes"]
        f["launches"] += 1
    # per family: absolute rates and the fraction of the roofline SURVEY section 8(d) assigns to it (MFMA for
    # attention / conv / GEMM main loops, HBM for the normalisations and the weight-streaming GEMVs)
    mfma_fams = ("conv3x3", "conv1x1", "linear", "geglu", "attn_self", "attn_cross", "conv_temporal", "attn_temporal")
    families = {k: dict(ms=v["seconds"] * 1e3, launches=v["launches"],
                        tflops=(v["flops"] / v["seconds"] / 1e12 if v["flops"] else None),
                        gbs=v["bytes"] / v["seconds"] / 1e9,
                        roofline=("mfma" if k in mfma_fams else "hbm"),
                        frac=(v["flops"] / v["seconds"] / 1e12 / MFMA_PEAK_TFLOPS if k in mfma_fams
                              else v["bytes"] / v["seconds"] / 1e9 / HBM_PEAK_GBS))
                for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["seconds"])}
    return roof, families, total


def cpu_baseline(cfg_name, images):
    """fp32 oracle UNet on the host cores, same CFG batch-2 forward. Bounded: 1 warm-up + 3 timed."""
    sys.path.insert(0, ROOT)
    from oracle import unet_ref as U
    torch.manual_seed(0)
    m = U.build(cfg_name, seed=0)
    cfg = U.SD15_CONFIG if cfg_name == "sd15" else U.SDXL_CONFIG
    B = 2 * images
    hw = cfg["sample_size"]
    x = torch.randn(B, 4, hw, hw)
    e = torch.randn(B, 77, cfg["cross_attention_dim"])
    added = None
    if cfg_name == "sdxl":
        added = dict(text_embeds=torch.randn(B, 1280), time_ids=torch.tensor([[1024., 1024, 0, 0, 1024, 1024]] * B))
    with torch.inference_mode():
        m(x, 981, e, added_cond_kwargs=added)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            m(x, 981, e, added_cond_kwargs=added)
            ts.append(time.perf_counter() - t0)
    med = sorted(ts)[len(ts) // 2]
    return dict(value=1.0 / med, unit="it/s", cores=torch.get_num_threads(), kind="port",
                sample=f"3 timed CFG batch-{B} fp32 UNet forwards of the oracle restatement (oracle/unet_ref.py) after 1 warm-up; "
                       f"median {med:.2f} s/forward; diffusers itself is not installable here")


class _Config(dict):
    __getattr__ = dict.get


class _DDIMSchedulerLike:
    """Harness stand-in with diffusers' DDIMScheduler call surface (diffusers is not installable here): SD1.5's scaled-linear
    betas, leading spacing with steps_offset 1, eta = 0, set_alpha_to_one False. Its `step` is the eager torch arithmetic a
    pipeline would run without trace_scheduler; compile(..., trace_scheduler=True) replaces it."""

    init_noise_sigma = 1.0
    _sfast_ddim_like = True  # declared look-alike: patch_scheduler recognises DDIM by class name or this opt-in, never by attributes

    def __init__(self):
        self.config = _Config(num_train_timesteps=1000, prediction_type="epsilon", clip_sample=False, thresholding=False,
                              steps_offset=1, timestep_spacing="leading")
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, n, device=None):
        self.num_inference_steps = n
        ratio = 1000 // n
        self.timesteps = ((torch.arange(0, n) * ratio).flip(0) + 1).to(device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None, variance_noise=None,
             return_dict=True):
        t = int(timestep)
        prev = t - 1000 // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        x0 = (sample - (1 - a_t) ** 0.5 * model_output) / a_t ** 0.5
        out = a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * model_output
        return (out,) if not return_dict else _Config(prev_sample=out)


class _EulerSchedulerLike:
    """Harness stand-in with the call surface and arithmetic of diffusers' EulerDiscreteScheduler (the default of its SDXL pipelines;
    s_churn = 0): sigma schedule from SD's scaled-linear betas, `scale_model_input`, first-order step, host-side step index."""
    _sfast_euler_like = True

    def __init__(self):
        self.config = _Config(num_train_timesteps=1000, prediction_type="epsilon")
        betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float64) ** 2
        acp = torch.cumprod(1.0 - betas, dim=0)
        self._sig_all = ((1 - acp) / acp) ** 0.5
        self.sigmas, self.timesteps, self._step_index, self.is_scale_input_called = None, None, None, False
        self.init_noise_sigma = 1.0

    @property
    def step_index(self):
        return self._step_index

    def set_timesteps(self, n, device=None):
        ts = torch.linspace(999, 0, n).round().long()
        self.timesteps = ts.to(device)
        self.sigmas = torch.cat([self._sig_all[ts], torch.zeros(1, dtype=torch.float64)]).to(torch.float32).to(device)
        self.init_noise_sigma = float((self.sigmas.max() ** 2 + 1) ** 0.5)
        self._step_index = None

    def _init_step_index(self, timestep):
        self._step_index = int((self.timesteps == int(timestep)).nonzero()[0])

    def scale_model_input(self, sample, timestep):
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma = self.sigmas[self._step_index]
        self.is_scale_input_called = True
        return sample / ((sigma ** 2 + 1) ** 0.5)

    def step(self, model_output, timestep, sample, s_churn=0.0, s_tmin=0.0, s_tmax=float("inf"), s_noise=1.0, generator=None, return_dict=True):
        if self._step_index is None:
            self._init_step_index(timestep)
        sigma = self.sigmas[self._step_index]
        x = sample.to(torch.float32)
        x0 = x - sigma * model_output
        prev = (x + (x - x0) / sigma * (self.sigmas[self._step_index + 1] - sigma)).to(model_output.dtype)
        self._step_index += 1
        return (prev,) if not return_dict else _Config(prev_sample=prev)


class _PipelineLike:
    """The denoise loop of diffusers' StableDiffusion(XL)Pipeline.__call__ (classifier-free guidance), prompt embeddings given."""

    def __init__(self, unet, scheduler, device):
        self.unet, self.scheduler, self.vae, self.device = unet, scheduler, None, device

    @torch.no_grad()
    def denoise(self, latents, prompt_embeds, guidance_scale, timesteps, added_cond_kwargs=None):
        kw = {} if added_cond_kwargs is None else {"added_cond_kwargs": added_cond_kwargs}
        for t in timesteps:
            latent_model_input = torch.cat([latents] * 2)
            latent_model_input = self.scheduler.scale_model_input(latent_model_input, t)
            noise_pred = self.unet(latent_model_input, t, encoder_hidden_states=prompt_embeds, return_dict=False, **kw)[0]
            noise_pred_uncond, noise_pred_text = noise_pred.chunk(2)
            noise_pred = noise_pred_uncond + guidance_scale * (noise_pred_text - noise_pred_uncond)
            latents = self.scheduler.step(noise_pred, t, latents, return_dict=False)[0]
        return latents


def through_compile(args, cfg, params, dev, latents, ehs, engine_ms, added=None):
    """The same step measured through the DROP-IN surface: a module with diffusers' parameter layout + a pipeline-shaped denoise
    loop + `sfast.compilers.compile(pipe, config)` with enable_cuda_graph and trace_scheduler -- i.e. what a stable-fast user
    runs (reference examples/optimize_stable_diffusion_pipeline.py:127-151), including the per-step input copies, the output
    clone and the pipeline's own guidance arithmetic that the fused DenoiseLoop does not have."""
    from sfast.compilers.diffusion_pipeline_compiler import CompilationConfig, compile
    from sfast.engine.unet_spec import module_from_params
    unet = module_from_params(cfg, params)
    euler = added is not None  # SDXL: its pipelines' default scheduler
    sched = _EulerSchedulerLike() if euler else _DDIMSchedulerLike()
    pipe = _PipelineLike(unet, sched, dev)
    config = CompilationConfig.Default()
    config.enable_cuda_graph = True
    config.trace_scheduler = True
    compile(pipe, config)
    sched.set_timesteps(50, device=dev)
    ts = list(sched.timesteps)
    lat = latents.clone() * (sched.init_noise_sigma if euler else 1.0)
    pipe.denoise(lat, ehs, 7.5, ts[:3], added)  # builds + captures the plan
    torch.cuda.synchronize()
    steps = min(args.steps, 50 if euler else 100)  # the Euler step index walks the 50-entry sigma table once
    seq = [ts[i % 50] for i in range(steps)]
    if euler:
        sched._step_index = None
    t0 = time.perf_counter()
    out = pipe.denoise(lat, ehs, 7.5, seq, added)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms = el / steps * 1e3
    return {"value": steps / el, "unit": "it/s", "ms_per_step": ms, "steps": steps, "gap_vs_fused_loop": ms / engine_ms - 1.0,
            "scheduler": type(sched).__name__, "native_scheduler_steps": getattr(sched.step, "native_calls", 0),
            "outputs_finite": bool(torch.isfinite(out).all()),
            "path": "module_from_params -> sfast.compilers.compile(enable_cuda_graph, trace_scheduler) -> pipeline-shaped CFG loop"}


def _time_steps(step, steps, warmup, dev, preheat=0.0):
    if preheat > 0:   # sustained load first (see --preheat-seconds): a capture or a plan build in between lets the clocks fall back
        t_end, k = time.perf_counter() + preheat, 0
        while time.perf_counter() < t_end:
            for _ in range(8):
                step(k)
                k += 1
            torch.cuda.synchronize(dev)
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps


def step_variants(args, engine, cfg, hw, dev, latents, ehs, headline_ms):
    """Two more lines from the same process, same weights (N = 1): (a) the headline step with the 16 text K/V projections hoisted
    out of the graph (run once per prompt -- a loop-invariant hoist a pipeline may do, `compile()` cannot; the headline keeps them
    in, so its step is literally one UNet forward + guidance + scheduler update); (b) the LITERAL batch-1 step SURVEY section 8d
    lists beside the CFG one: UNet forward at B = 1 (no classifier-free guidance) + the DDIM update, one hipGraph."""
    from sfast.engine.denoise import DenoiseLoop
    from sfast.engine.unet2d import capture_plan_graph
    from sfast.hip import lib as L
    out = {}
    steps, warm = min(args.steps, 100), min(args.warmup, 10)
    loop = DenoiseLoop(engine, images=args.images, height=hw, width=hw, ctx_len=77, guidance=7.5, num_steps=50,
                       use_graph=not args.no_graph, hoist_text_kv=True)
    if args.config == "sdxl":
        si = loop.plan.static_in
        si["text_embeds"].normal_()
        si["time_ids"].copy_(torch.tensor([1024., 1024, 0, 0, 1024, 1024], device=dev).repeat(2 * args.images))
    loop.set_inputs(latents, ehs)
    loop.capture(warmups=2)
    ms = _time_steps(loop.step, steps, warm, dev, preheat=args.preheat_seconds / 2) * 1e3
    out["text_kv_hoisted"] = {"value": 1e3 / ms, "unit": "it/s", "ms_per_step": ms, "kernel_launches_per_step": len(loop._step_ops) + 2,
                              "gain_vs_headline": headline_ms / ms,
                              "note": "cross-attention K/V projections of the text context run once per prompt (DenoiseLoop(hoist_text_kv=True)), not per step"}
    del loop   # its graph retires through its OwnedGraph handle
    if args.images == 1:
        plan = engine.get_plan(1, hw, hw, 77)
        kw = {}
        if args.config == "sdxl":
            kw["added_cond_kwargs"] = {"text_embeds": torch.randn(1, 1280, device=dev).half(),
                                       "time_ids": torch.tensor([[1024., 1024, 0, 0, 1024, 1024]], device=dev).half()}
        engine.load_inputs(plan, latents[:1], 981, ehs[1:2], **kw)
        coef = torch.tensor([[1.02, -0.05]], dtype=torch.float32, device=dev)   # x_prev = A x + B eps: one DDIM row in linear form
        lib = L.init_device(dev)
        lat = latents[:1].clone()
        n = lat.numel()

        def tail(stream):   # scheduler update written straight into the plan's sample input (eps-prediction, eta = 0)
            L.check(lib.sfast_hip_linear_step(plan.static_out.data_ptr(), lat.data_ptr(), plan.static_in["sample"].data_ptr(), coef.data_ptr(),
                                              None, 0, 1, n, engine.dt, stream), "sfast_hip_linear_step")
        stream = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(stream):
            plan.run(stream.cuda_stream)
        torch.cuda.synchronize(dev)
        graph, _ = capture_plan_graph(plan, stream, tail=tail)
        with torch.cuda.stream(stream):
            ms1 = _time_steps(lambda i: graph.replay(), steps, warm, dev, preheat=args.preheat_seconds / 2) * 1e3
        out["literal_b1"] = {"value": 1e3 / ms1, "unit": "it/s", "ms_per_step": ms1, "unet_batch": 1, "kernel_launches_per_step": len(plan.ops) + 1,
                             "note": "UNet forward at batch 1 (no classifier-free guidance) + DDIM update, one hipGraph -- SURVEY 8d config 2, B = 1"}
    return out


def child_variant(extra, steps=200, warmup=150, timeout=150, env_extra=None):
    """north_star asks for it/s on BOTH latent sizes and BASELINE configs[3] runs 8 images per GPU; the driver times `python bench.py`
    only. So the default SD1.5 run ends by timing (a) the 1x4x128x128-latent step (BASELINE configs[2]: SDXL 1024x1024 bs=1 fp16) and
    (b) the per-GPU shape of configs[3] (SD1.5, 8 images = UNet batch 16) -- CFG UNet + guidance + DDIM update as one hipGraph,
    packaged kernel choices -- each in a CHILD process of this same script (its own weights, plan and `roofline` block) and embeds the
    children's JSON lines. A failure is reported, never raised, and a hung child costs at most `timeout` seconds (240 s
    before): the contract line must survive. The children time 200 steps behind 150 warm-up steps like the headline: with 20 / 5 (rounds
    4 - 6) they read 3 - 5 % below the standalone runs of the same configurations (8 images: 51.2 vs 53.7 steps/s, SDXL 42.5 vs 43.8 it/s in
    one session, `--steps 20` / `60` / `200` twice each) -- the chip takes seconds of sustained load to reach its steady clocks."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + list(extra) + ["--gpus", "1", "--steps", str(steps), "--warmup", str(warmup),
                                                                       "--no-cpu-baseline", "--no-end-to-end", "--no-variants"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
        line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")), None)
        if r.returncode != 0 or line is None:
            return {"error": f"child exited {r.returncode}", "stderr_tail": r.stderr[-400:], "seconds": time.perf_counter() - t0}
        child = json.loads(line)
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}", "seconds": time.perf_counter() - t0}
    keep = ("metric", "value", "unit", "steps", "warmup", "preheat_seconds", "ms_per_step", "dtype", "config", "gpu_ms_per_step_events", "outputs_finite",
            "kernel_launches_per_step", "packed_weight_launches", "roofline", "kernel_families", "sum_of_kernel_ms_eager")
    out = {k: child[k] for k in keep if k in child}
    out["wall_seconds_of_child_process"] = time.perf_counter() - t0
    out["command"] = " ".join(["python", "bench.py"] + cmd[2:])
    return out


def sdxl_variant(steps=200, warmup=150, timeout=150):
    return child_variant(["--config", "sdxl"], steps, warmup, timeout)


def bs8_variant(steps=200, warmup=150, timeout=150):
    return child_variant(["--config", "sd15", "--images", "8"], steps, warmup, timeout)


def batch_invariant_variant(bs8, steps=200, warmup=150, timeout=150):
    """the cost of SFAST_BATCH_INVARIANT=1 (every kernel choice and statistics partition follows the per-sample
    problem at the reference batch 2, so a sample's latents are bit-equal at any batch: tests/test_unet_gpu.py
    test_batch_invariant_mode_is_bit_exact_across_batch_sizes) where it costs most -- 8 images per GPU (UNet batch 16), whose
    measured choices are the 256-row tiles the mode gives up. At batch 2 (the headline) the mode changes nothing."""
    out = child_variant(["--config", "sd15", "--images", "8"], steps, warmup, timeout, env_extra={"SFAST_BATCH_INVARIANT": "1"})
    if "value" in out:
        out["env"] = "SFAST_BATCH_INVARIANT=1"
        if isinstance(out.get("config"), dict) and "workload" in out["config"]:
            out["config"]["workload"] += " [SFAST_BATCH_INVARIANT=1]"
        if isinstance(bs8, dict) and bs8.get("value"):
            out["relative_to_variants_bs8"] = out["value"] / bs8["value"]
    return out


def bs64_sharded(args, engine, cfg, hw, dev, rank, world, use_dist):
    """BASELINE.json configs[3]: SD1.5 512x512 bs = 64 fp16 sharded over the node -- 64 / N images per GPU (UNet batch 128 / N with
    CFG), independent per-GPU denoise loops, no per-step collective. Reported as image-steps/s over all ranks (max-over-ranks time)
    and as ms per image-step; N = 8 gives the per-GPU shape the packaged kernel choices cover (8 images, UNet batch 16)."""
    from sfast.engine.denoise import DenoiseLoop

    def agree(value):
        if not use_dist:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        return float(t.item())

    images = max(1, 64 // world)
    err, loop = None, None
    try:
        loop = DenoiseLoop(engine, images=images, height=hw, width=hw, ctx_len=77, guidance=7.5, num_steps=50, use_graph=not args.no_graph)
        g = torch.Generator(device=dev).manual_seed(4321 + rank)
        loop.set_inputs(torch.randn(images, 4, hw, hw, generator=g, device=dev).half(),
                        torch.randn(2 * images, 77, cfg["cross_attention_dim"], generator=g, device=dev).half())
        loop.capture(warmups=2)
        for i in range(3):
            loop.step(i)
        torch.cuda.synchronize(dev)
    except Exception as e:
        err = f"{type(e).__name__}: {e}"
    if agree(1.0 if err else 0.0) > 0:      # also the start barrier
        return {"error": err or "set-up failed on another rank", "n_gpus": world}
    steps = max(5, min(args.steps, 30))
    try:
        t0 = time.perf_counter()
        for i in range(steps):
            loop.step(i)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
    except Exception as e:
        err, el = f"{type(e).__name__}: {e}", float("inf")
    el = agree(el)
    if el == float("inf"):
        return {"error": err or "the timed steps failed on another rank", "n_gpus": world}
    return {"metric": "SD1.5 512x512 bs=64 fp16 sharded over the node (image-steps/s)", "value": world * images * steps / el, "unit": "image-steps/s",
            "n_gpus": world, "images_per_gpu": images, "unet_batch_per_gpu": 2 * images, "global_batch": world * images, "steps": steps,
            "ms_per_step": el / steps * 1e3, "ms_per_image_step": el / steps / images * 1e3, "scaling": "strong" if world * images == 64 else "weak",
            "outputs_finite": bool(torch.isfinite(loop.latents).all()),
            "note": "BASELINE.json configs[3]; batch-parallel shards, weights broadcast once (RCCL), no per-step collective"}


def end_to_end(args, loop, dev, rank, world, use_dist):
    """Second half of BASELINE.json's metric, "end-to-end ms/image at 1/2/4/8 GPU": ONE timed region per image holding everything the
    reference's protocol times around `pipe(**kwargs)` (/root/reference/examples/optimize_stable_diffusion_pipeline.py:127-151) --
    CLIP text encoding of the (uncond, cond) prompt pair, 50 CFG denoise iterations, VAE decode, image post-process, and the copy of
    the uint8 image to the host. Pieces: a random-init `transformers.CLIPTextModel` of the SD1.5 text-encoder architecture
    (CLIP ViT-L/14 text tower, 123,060,480 parameters), hipGraph-captured exactly as compile() captures `pipe.text_encoder`
    (compilers/diffusion_pipeline_compiler.py: `_graphed_with_fallback`, reference :93-118); the fused DenoiseLoop graph (the bench's
    own step); `post_quant_conv` as the eager 1x1 conv compile() leaves it; the native VAE decoder plan as a hipGraph
    (compile_vae); `sfast_hip_image_postprocess`. Each GPU renders its own image (weak scaling, like the it/s line)."""
    import torch.nn.functional as TF
    from sfast.compilers.diffusion_pipeline_compiler import _graphed_with_fallback
    from sfast.engine import VaeDecoderEngine, capture_plan_graph
    from sfast.engine.unet_spec import SD_VAE_DECODER_CONFIG, random_vae_decoder_params
    from sfast.hip import functional as Fn
    from transformers import CLIPTextConfig, CLIPTextModel
    # N > 1: no rank may be left alone in a collective. Everything that can fail runs between two all-reduces that every rank
    # reaches: the first carries "my set-up / warm-up failed" (and is the start barrier), the second the elapsed time (inf = failed).
    def agree(value):
        if not use_dist:
            return value
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        return float(t.item())

    err = None
    try:
        images = args.images
        torch.manual_seed(0)
        tcfg = CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                              max_position_embeddings=77, hidden_act="quick_gelu", projection_dim=768)
        text = CLIPTextModel(tcfg).to(dev, torch.float16).eval()
        n_text = sum(p.numel() for p in text.parameters())
        text.forward = _graphed_with_fallback(text.forward)
        vcfg = SD_VAE_DECODER_CONFIG
        vae = VaeDecoderEngine(vcfg, random_vae_decoder_params(vcfg, seed=0, dtype=torch.float16, device=dev))
        hw = loop.latents.shape[-1]
        vplan = vae.get_plan(images, hw, hw)
        g = torch.Generator(device=dev).manual_seed(77 + rank)
        pq_w = (torch.randn(4, 4, 1, 1, generator=g, device=dev) * 0.5).half()
        pq_b = torch.zeros(4, device=dev, dtype=torch.float16)
        lat0 = torch.randn(images, 4, hw, hw, generator=g, device=dev).half()
        ids = torch.randint(0, 49408, (2 * images, 77), generator=g, device=dev)
        ids[:, 0], ids[:, -1] = 49406, 49407
        host = torch.empty((images, 8 * hw, 8 * hw, 3), dtype=torch.uint8).pin_memory()
        vstream = torch.cuda.Stream(device=dev)
        vae.load_inputs(vplan, lat0)
        with torch.cuda.stream(vstream):
            vplan.run(vstream.cuda_stream)
        torch.cuda.synchronize()
        vgraph, _ = capture_plan_graph(vplan, vstream)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]

        def one_image(mark=False):
            with torch.no_grad():
                if mark:
                    ev[0].record()
                ehs = text(ids)[0]                                   # [2 * images, 77, 768]: rows [uncond..., cond...]
                if mark:
                    ev[1].record()
                loop.set_inputs(lat0, ehs)                            # also runs the text-side K/V projections of the UNet, once
                loop.set_step(0)
                for i in range(50):
                    loop.step(i)
                if mark:
                    ev[2].record()
                z = TF.conv2d(loop.latents * (1.0 / 0.18215), pq_w, pq_b)   # latents / scaling_factor -> post_quant_conv
                vae.load_inputs(vplan, z)
                vgraph.replay()
                if mark:
                    ev[3].record()
                img = Fn.image_postprocess(vplan.static_out)          # [-1, 1] NCHW f16 -> uint8 NHWC
                host.copy_(img, non_blocking=True)
                if mark:
                    ev[4].record()
                torch.cuda.synchronize()


        one_image()   # warm-up: captures the text-encoder graph
        one_image()
        torch.cuda.synchronize()
    except Exception as e:  # reported below, after every rank has been told
        err = f"{type(e).__name__}: {e}"
    if agree(1.0 if err else 0.0) > 0:
        return {"error": err or "set-up failed on another rank", "n_gpus": world}
    n = max(1, args.e2e_images)
    try:
        t0 = time.perf_counter()
        for _ in range(n):
            one_image()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    except Exception as e:
        err, el = f"{type(e).__name__}: {e}", float("inf")
    el = agree(el)   # max over ranks
    if el == float("inf"):
        return {"error": err or "the timed images failed on another rank", "n_gpus": world}
    one_image(mark=True)
    parts = [ev[i].elapsed_time(ev[i + 1]) for i in range(4)]
    ok = bool(host.float().std() > 0) and bool(torch.isfinite(loop.latents).all())
    return {"ms_per_image_end_to_end": el / n / images * 1e3, "images_per_s_all_gpus": world * n * images / el, "images_timed_per_gpu": n * images,
            "n_gpus": world, "breakdown_ms": {"text_encoder": parts[0], "denoise_50_steps": parts[1], "vae_decode": parts[2],
                                              "postprocess_and_host_copy": parts[3]},
            "text_encoder": {"arch": "CLIP ViT-L/14 text (transformers.CLIPTextModel, random init)", "params": n_text,
                             "hipgraph": bool(getattr(text.forward, "_cached", None))},
            "image": [8 * hw, 8 * hw, 3], "outputs_ok": ok,
            "protocol": "one timed region per image: token ids on the device -> uint8 image on the host (text encode, 50-step CFG DDIM, "
                        "post_quant_conv, VAE decode, post-process, D2H copy); reference examples/optimize_stable_diffusion_pipeline.py:127-151"}


def bench_vae(args, dev, rank, world, use_dist):
    """`--config vae`: one step = one VAE decode (64x64 latent -> 512x512 image, bs = --images, fp16) replayed as a hipGraph."""
    from sfast.engine import VaeDecoderEngine, capture_plan_graph
    from sfast.engine.unet_spec import SD_VAE_DECODER_CONFIG, random_vae_decoder_params
    cfg = SD_VAE_DECODER_CONFIG
    params = random_vae_decoder_params(cfg, seed=0, dtype=torch.float16, device=dev)
    eng = VaeDecoderEngine(cfg, params)
    hw = 64
    z = torch.randn(args.images, 4, hw, hw, generator=torch.Generator(device=dev).manual_seed(1 + rank), device=dev).half()
    plan = eng.get_plan(args.images, hw, hw)
    eng.load_inputs(plan, z)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        plan.run(stream.cuda_stream)
    torch.cuda.synchronize()
    graph, _ = capture_plan_graph(plan, stream)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            graph.replay()
    sync_all()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(args.steps):
            graph.replay()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    if rank != 0:
        return
    ms = elapsed / args.steps * 1e3
    out = {"metric": "SD VAE decode 64x64 latent -> 512x512 image fp16 (images/s)", "value": world * args.images * args.steps / elapsed,
           "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
           "config": {"workload": f"AutoencoderKL decoder (SD1.x VAE, 49.5 M parameters), {args.images} x [4,64,64] latent -> [3,512,512] per GPU, "
                                  "hipGraph replay, seeded random-init weights", "images_per_gpu": args.images, "parallelism": f"replicas x{world}"},
           "outputs_finite": bool(torch.isfinite(plan.static_out).all()), "kernel_launches_per_step": len(plan.ops)}
    if not args.no_roofline and world == 1:
        class _L:  # per_op_timing expects an object with .plan
            pass
        holder = _L()
        holder.plan = plan
        rows = per_op_timing(holder)
        roof, families, total = roofline_from(rows, plan, "vae")
        out["roofline"], out["kernel_families"], out["sum_of_kernel_ms_eager"] = roof, families, total * 1e3
    if not args.no_cpu_baseline and world == 1:
        # baselines beside it: the oracle restatement of the same decoder (a) eagerly on this GPU through PyTorch-ROCm,
        # (b) in fp32 on the host cores (one decode, no warm-up: bounded)
        sys.path.insert(0, ROOT)
        from oracle import vae_ref as V
        ref16 = V.Decoder(**cfg).to(dev, torch.float16).eval().to(memory_format=torch.channels_last)
        ref16.load_state_dict({k: v for k, v in params.items()})
        with torch.no_grad():
            y_ref = ref16(z)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                ref16(z)
            torch.cuda.synchronize()
            t_eager = (time.perf_counter() - t0) / 5
        eng.load_inputs(plan, z)
        graph.replay()
        torch.cuda.synchronize()
        out["pytorch_rocm_eager_baseline"] = {"ms": t_eager * 1e3, "speedup": t_eager * 1e3 / ms,
                                              "rel_l2_engine_vs_eager_fp16": float((plan.static_out.float() - y_ref.float()).norm() / y_ref.float().norm())}
        ref32 = V.Decoder(**cfg).eval()
        ref32.load_state_dict({k: v.float().cpu() for k, v in params.items()})
        torch.set_num_threads(os.cpu_count() or 1)
        with torch.no_grad():
            t0 = time.perf_counter()
            y32 = ref32(z[:1].float().cpu())
            t_cpu = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / t_cpu, "unit": "images/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": "one fp32 decode of the oracle restatement (oracle/vae_ref.py) on the host cores, no warm-up",
                               "rel_l2_engine_vs_fp32": float((plan.static_out[:1].float().cpu() - y32).norm() / y32.norm())}
    print(json.dumps(out))


def bench_svd(args, dev, rank, world, use_dist):
    """`--config svd`: one step = one CFG batch-2 forward of the SVD-XT spatio-temporal UNet (576x1024 -> 72x128 latent, 25 frames),
    replayed as a hipGraph. Seeded random-init weights of the published architecture (1.52 B parameters)."""
    from sfast.engine import SVDUNetEngine, capture_plan_graph
    from sfast.engine.unet_spec import SVD_CONFIG, random_svd_params
    cfg = SVD_CONFIG
    params = random_svd_params(cfg, seed=0, dtype=torch.float16, device=dev)
    eng = SVDUNetEngine(cfg, params)
    B, Fr, H, W = 2 * args.images, args.frames, 72, 128
    g = torch.Generator(device=dev).manual_seed(1 + rank)
    sample = torch.randn(B, Fr, 8, H, W, generator=g, device=dev).half()
    ehs = torch.randn(B, 1, 1024, generator=g, device=dev).half()
    tids = torch.tensor([[6.0, 127.0, 0.02]] * B, device=dev)
    plan = eng.get_plan(B, Fr, H, W)
    eng.load_inputs(plan, sample, 500.0, ehs, tids)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        plan.run(stream.cuda_stream)
    torch.cuda.synchronize()
    graph, _ = capture_plan_graph(plan, stream)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            graph.replay()
    sync_all()
    t0 = time.perf_counter()
    with torch.cuda.stream(stream):
        for _ in range(args.steps):
            graph.replay()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    if rank != 0:
        return
    summ = plan.summary()
    tflop = sum(v["gflop"] for v in summ.values()) / 1e3
    ms = elapsed / args.steps * 1e3
    out = {"metric": "UNet iters/sec SVD-XT 576x1024 25 frames fp16", "value": world * args.steps / elapsed, "unit": "it/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f16", "data": "synthetic",
           "config": {"workload": f"UNetSpatioTemporalConditionModel (SVD-XT, 1.52 B parameters), CFG batch {B} x {Fr} frames x [8,{H},{W}] latent, "
                                  "hipGraph replay, seeded random-init weights", "videos_per_gpu": args.images, "frames": Fr,
                      "parallelism": f"replicas x{world}"},
           "outputs_finite": bool(torch.isfinite(plan.static_out).all()), "kernel_launches_per_step": len(plan.ops),
           "algorithmic_tflop_per_step": tflop, "achieved_tflops": tflop / (ms * 1e-3), "gn_fused": plan.gn_fused,
           "activation_pool_gb": plan.pool.total_bytes() / 1e9}
    if not args.no_roofline and world == 1:
        class _H:
            pass
        holder = _H()
        holder.plan = plan
        rows = per_op_timing(holder, reps=1, burst=2)
        roof, families, total = roofline_from(rows, plan, "svd")
        out["roofline"], out["kernel_families"], out["sum_of_kernel_ms_eager"] = roof, families, total * 1e3
    if not args.no_cpu_baseline and world == 1:
        sys.path.insert(0, ROOT)
        from oracle import svd_ref as SR
        ref = SR.build("svd", seed=0)
        torch.set_num_threads(os.cpu_count() or 1)
        with torch.no_grad():
            t0 = time.perf_counter()
            ref(sample[:1, :2].float().cpu(), 500.0, ehs[:1].float().cpu(), tids[:1].cpu())
            t_cpu = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / (t_cpu * B * Fr / 2.0), "unit": "it/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": f"one fp32 forward of the oracle restatement (oracle/svd_ref.py) on 1 video x 2 frames at 72x128 "
                                         f"({t_cpu:.1f} s), scaled linearly to {B} x {Fr} frames"}
    print(json.dumps(out), flush=True)


def torchrun_argv(n, argv, port=None):
    """Command that re-launches this script as `n` ranks of one node (what the driver does itself for N > 1)."""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` typed by hand: one process per GPU needs the launcher; become it
        import subprocess
        raise SystemExit(subprocess.call(torchrun_argv(args.gpus, sys.argv[1:])))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU: the HIP path has no CPU fallback")
    from sfast.hip import lib as _L
    if not os.path.exists(_L.LIB_PATH) and local == 0 and world == 1:
        # a checkout without the in-tree .so: build the product library first (hipcc, ~1 min); never a fallback
        import importlib.util
        spec = importlib.util.spec_from_file_location("sfast_build", os.path.join(ROOT, "stable-fast_amd", "build.py"))
        _build = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_build)
        _build.build(verbose=False)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ  # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.config == "vae":
        bench_vae(args, dev, rank, world, use_dist)
        if use_dist:
            dist.destroy_process_group()
        return
    if args.config == "svd":
        bench_svd(args, dev, rank, world, use_dist)
        if use_dist:
            dist.destroy_process_group()
        return

    from sfast.engine import UNet2DEngine
    from sfast.engine.denoise import DenoiseLoop
    from sfast.engine.replicas import broadcast_parameters
    from sfast.engine.unet_spec import SD15_CONFIG, SDXL_CONFIG, random_params

    cfg = SD15_CONFIG if args.config == "sd15" else SDXL_CONFIG
    # rank 0 owns the weights; replicas receive them with one bucketed RCCL broadcast over xGMI
    if args.weights and rank == 0:
        from sfast.engine.unet_spec import load_params
        params = load_params(args.weights, cfg, dtype=torch.float16, device=dev)   # the other ranks receive them by the broadcast below
    else:
        params = random_params(cfg, seed=0 if rank == 0 else 1000 + rank, dtype=torch.float16, device=dev)
    t0 = time.perf_counter()
    bytes_bcast = broadcast_parameters(params, src=0) if world > 1 else 0
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t0

    engine = UNet2DEngine(cfg, params)
    hw = cfg["sample_size"]
    if world > 1:
        # rank 0 builds (and, for shapes outside the packaged cache, times) its plan first and shares the kernel choices:
        # every replica then runs identical kernels and nobody else spends start-up time tuning
        from sfast.engine.replicas import share_tune_cache
        if rank == 0:
            engine.get_plan(2 * args.images, hw, hw, 77)
        share_tune_cache(src=0)
    loop = DenoiseLoop(engine, images=args.images, height=hw, width=hw, ctx_len=77, guidance=7.5, num_steps=50,
                       use_graph=not args.no_graph)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    latents = torch.randn(args.images, 4, hw, hw, generator=g, device=dev).half()
    ehs = torch.randn(2 * args.images, 77, cfg["cross_attention_dim"], generator=g, device=dev).half()
    loop.set_inputs(latents, ehs)
    if args.config == "sdxl":
        si = loop.plan.static_in
        si["text_embeds"].copy_(torch.randn(si["text_embeds"].shape, generator=g, device=dev).half())
        si["time_ids"].copy_(torch.tensor([1024., 1024, 0, 0, 1024, 1024], device=dev).repeat(2 * args.images))
    loop.capture(warmups=3)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # Pre-heat: the chip reaches its steady clocks only after seconds of sustained load (8 images per GPU: 54.1 steps/s behind 20 warm-up
    # steps, 58.2 behind 150, 58.7 - 58.8 behind 500, timed over 200 or 1000 steps alike; SDXL 43.8 / 45.8 / 46.0 -- one session,
    # profiles/r06_warmup_ramp_run43.log). The step is replayed untimed for --preheat-seconds before the W warm-up steps of the contract.
    if args.preheat_seconds > 0:
        t_end, k = time.perf_counter() + args.preheat_seconds, 0
        while time.perf_counter() < t_end:
            for _ in range(8):
                loop.step(k)
                k += 1
            torch.cuda.synchronize()
    for i in range(args.warmup):
        loop.step(i)
    sync_all()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(args.steps):
        loop.step(i)
    ev1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    gpu_ms = ev0.elapsed_time(ev1)
    finite = bool(torch.isfinite(loop.latents).all())

    if rank == 0:
        value = world * args.steps / elapsed
        out = {
            "metric": (f"UNet iters/sec SD1.5 512x512 bs={args.images} fp16" if args.config == "sd15"
                       else f"UNet iters/sec SDXL 1024x1024 bs={args.images} fp16"),
            "value": value, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "preheat_seconds": args.preheat_seconds,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_image_50_steps_unet_and_scheduler": elapsed / args.steps * 1e3 * 50 / max(1, args.images), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": ("synthetic latents / text embeddings, weights from " + args.weights) if args.weights else "synthetic",
            "config": {"workload": f"{'SD1.5 512x512' if args.config == 'sd15' else 'SDXL 1024x1024'} bs={args.images} fp16, 50-step DDIM "
                                   f"schedule, one step = CFG batch-{2 * args.images} UNet forward + guidance combine + DDIM update, "
                                   f"hipGraph {'off' if args.no_graph else 'on'}, seeded random-init weights, {world} replica(s) "
                                   f"(one image per GPU, weights broadcast once over RCCL)",
                       "images_per_gpu": args.images, "unet_batch": 2 * args.images, "latent": [4, hw, hw], "parallelism": f"replicas x{world}"},
            "gpu_ms_per_step_events": gpu_ms / args.steps, "outputs_finite": finite,
            "reference_published_other_hw": {"H100": 104.6, "A100": 61.8, "RTX4080": 51.6, "source": "BASELINE.md section 1 (stable-fast README)"},
            "kernel_launches_per_step": len(loop._step_ops) + 2, "text_kv_in_step_graph": not loop.hoist_text_kv,
            "packed_weight_launches": int(getattr(loop.plan, "packed_ops", 0)),  # GEMM / conv launches on the pipe-4 kernels (SFAST_PACKED_WEIGHTS=0: none)
            "graph_calibration_ms": getattr(loop.plan, "graph_calibration_ms", None),
            "activation_pool_mb": loop.plan.pool.total_bytes() / 1e6,
        }
        if world > 1:
            out["weight_broadcast"] = {"bytes": bytes_bcast, "seconds": t_bcast, "gb_per_s": bytes_bcast / max(t_bcast, 1e-9) / 1e9}
        if not args.no_roofline and world == 1:
            rows = per_op_timing(loop)
            roof, families, eager_total = roofline_from(rows, loop.plan, args.config if args.images == 1 else (f"bs{args.images}" if args.config == "sd15" else f"{args.config}_bs{args.images}"))  # (tools/gpu_pmc_bench.sh bs8 -> ..._bs8.json)
            out["roofline"] = roof
            out["kernel_families"] = families
            out["sum_of_kernel_ms_eager"] = eager_total * 1e3
            if args.dump_kernels:
                os.makedirs(os.path.dirname(os.path.abspath(args.dump_kernels)), exist_ok=True)
                with open(args.dump_kernels, "w") as f:
                    json.dump(rows, f, indent=1)
        if args.through_compile and world == 1 and args.images == 1:
            added = None
            if args.config == "sdxl":
                si = loop.plan.static_in
                added = {"text_embeds": si["text_embeds"].reshape(2, -1).clone(), "time_ids": si["time_ids"].reshape(2, -1).clone()}
            out["through_compile"] = through_compile(args, cfg, params, dev, latents, ehs, elapsed / args.steps * 1e3, added)
        if world == 1 and not args.no_variants:
            try:
                out["variants"] = step_variants(args, engine, cfg, hw, dev, latents, ehs, elapsed / args.steps * 1e3)
            except Exception as e:   # extra lines must never take the contract line down
                out["variants"] = {"error": f"{type(e).__name__}: {e}"}
        if (world == 1 and args.config == "sd15" and args.images == 1 and not args.no_variants and not args.no_sdxl_variant
                and not args.no_graph and isinstance(out.get("variants"), dict)):
            out["variants"]["sdxl"] = sdxl_variant()
            out["variants"]["bs8"] = bs8_variant()
            out["variants"]["batch_invariant"] = batch_invariant_variant(out["variants"]["bs8"])
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.config, args.images)
    # BASELINE configs[3] (bs = 64 sharded over the ranks): every rank runs its shard; a leg of the N = 8 run (or --bs64-sharded)
    c3 = None
    if args.config == "sd15" and not args.no_variants and args.images == 1 and (world == 8 or args.bs64_sharded):
        try:
            c3 = bs64_sharded(args, engine, cfg, hw, dev, rank, world, use_dist)
        except Exception as e:
            c3 = {"error": f"{type(e).__name__}: {e}"}
    # end-to-end ms/image (every rank renders its own image; rank 0 reports the max-over-ranks time)
    e2e = None
    if args.config == "sd15" and not args.no_end_to_end and not args.no_graph:
        try:
            e2e = end_to_end(args, loop, dev, rank, world, use_dist)
        except Exception as e:  # the it/s line is the contract; this leg must never take it down (end_to_end() itself keeps the ranks
            e2e = {"error": f"{type(e).__name__}: {e}"}   # of an N > 1 run in step when one of them fails; this catches the rest)
    if rank == 0:
        if c3 is not None:
            out["configs3_bs64_sharded"] = c3
        if e2e is not None:
            out["end_to_end"] = e2e
            if "ms_per_image_end_to_end" in e2e:
                out["ms_per_image_end_to_end"] = e2e["ms_per_image_end_to_end"]
        # headline values of the variant legs at the FRONT of the line: the driver keeps a truncated tail of long lines
        v = out.get("variants") if isinstance(out.get("variants"), dict) else {}
        summ = {k: {"value": v[k]["value"], "unit": v[k].get("unit"), "ms_per_step": v[k].get("ms_per_step"),
                    "workload": (v[k].get("config") or {}).get("workload", k)}
                for k in v if isinstance(v[k], dict) and "value" in v[k]}
        if summ:
            head = {k: out[k] for k in ("metric", "value", "unit") if k in out}
            head["variants_summary"] = summ
            out = {**head, **{k: val for k, val in out.items() if k not in head}}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
