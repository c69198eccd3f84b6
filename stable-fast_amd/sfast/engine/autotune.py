"""Measured kernel-variant selection for the MFMA implicit-GEMM ops of a plan.

The reference picks its convolution algorithm by benchmarking the candidates once per problem and
caching the winner (cuDNN `cudnnFind...Ex` behind `benchmark_cache`,
/root/reference/src/sfast/csrc/operators/cudnn/cudnn_convolution_impl.cc:344-370, :492-541). This is the
same idea for libsfast_hip.so: every distinct (problem shape, epilogue form) of a plan is timed on the
device over the library's tile shapes x main-loop structures x split-K factors, and the fastest
(variant, split_k) is written into the op's params struct before the plan is captured. Results are
cached per process (and optionally in a JSON file, `SFAST_TUNE_CACHE=path`), so later plans and
replicas pay nothing. `SFAST_AUTOTUNE=0` disables it; the library's analytic planner is used then.
"""
import ctypes as C
import json
import os
import threading

import torch

from ..hip import lib as L

_cache = {}
_cache_lock = threading.Lock()
_loaded_files = set()

SPLITS = (1, 2, 3, 4, 5, 6, 8, 10, 12, 15, 16, 20, 24)
VARIANTS = (1, 2, 3, 5, 11, 12, 13, 15, 16, 17, 18, 21, 22, 23, 24, 25, 26)
GEGLU_VARIANTS = (1, 3, 11, 13, 16, 18, 21, 23)
PK_VARIANTS = (41, 42, 43, 44, 45, 46)  # igemm_pk.h: packed weights global -> VGPR; only for ops that were handed packed copies
CONV_PATCH_VARIANTS = (31, 32, 34)   # conv_patch.hip: 3x3 / stride 1 / pad 1 convs only (sfast_hip_conv2d_plan says whether a problem fits)
# igemm_pp.h (round 6): 256-row ping-pong tiles -- 256 x 128 / 160 / 256 with the consumer groups issuing the LDS-DMA requests (51 - 53)
# or four producer waves (55, 56: ping-pong consumers; 57, 58: lockstep consumers, one barrier per K-tile); candidates only where the tiles alone put work on at least PP_MIN_TILES of the 256 CUs
PP_VARIANTS = (51, 52, 53, 55, 56, 57, 58)
PP_GEGLU_VARIANTS = (53, 57)
PP_BN = {51: 128, 52: 160, 53: 256, 55: 128, 56: 160, 57: 128, 58: 160}
PP_MIN_TILES = 32   # tiles alone; with split-K the workgroups (tiles x splits) must reach PP_MIN_WGS
PP_MIN_WGS = 96
MAX_SLAB_BYTES = 192 << 20


PACKAGED_CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tune_gfx950.json")


def export_cache():
    """Snapshot of the measured (variant, split_k) choices of this process -- what `share_tune_cache` sends to replicas."""
    with _cache_lock:
        return {k: list(v) for k, v in _cache.items()}


def import_cache(entries, overwrite=False):
    """Merge choices measured elsewhere (another rank, a file): replicas then build identical plans without timing anything."""
    n = 0
    with _cache_lock:
        for k, v in (entries or {}).items():
            if overwrite or k not in _cache:
                _cache[k] = (int(v[0]), int(v[1]))
                n += 1
    return n


def enabled():
    return os.environ.get("SFAST_AUTOTUNE", "1") not in ("0", "false", "off", "")


def _load_file(path):
    if not path or path in _loaded_files or not os.path.exists(path):
        return
    try:
        with open(path) as f:
            for k, v in json.load(f).items():
                _cache.setdefault(k, tuple(v))
    except (OSError, ValueError):
        pass
    _loaded_files.add(path)


def _save_file(path):
    if not path:
        return
    try:
        with open(path, "w") as f:
            json.dump({k: list(v) for k, v in sorted(_cache.items())}, f, indent=0)
    except OSError:
        pass


def _pp_candidates(M, N, K, geglu):
    if K % 64 != 0 or M < 1024:
        return ()
    return tuple(v for v in (PP_GEGLU_VARIANTS if geglu else PP_VARIANTS)
                 if -(-M // 256) * -(-N // (PP_BN[v] // 2 if geglu else PP_BN[v])) >= PP_MIN_TILES)


_extended = set()  # SFAST_TUNE_EXTEND=1: cached problems already re-timed against the pipe-5 candidates in this process


def problem_key(p, dtype_tag, device_name):
    if isinstance(p, L.GemmParams):
        epi = (int(p.geglu), int(p.act), int(p.n_wseg))
        return f"{device_name}|{dtype_tag}|gemm|{p.M}x{p.N}x{p.K}|{epi}"
    Hin = 2 * p.H if p.upsample2x else p.H
    Win = 2 * p.W if p.upsample2x else p.W
    Ho = (Hin + 2 * p.pad_h - p.dil_h * (p.KH - 1) - 1) // p.stride_h + 1
    Wo = (Win + 2 * p.pad_w - p.dil_w * (p.KW - 1) - 1) // p.stride_w + 1
    geo = (p.KH, p.KW, p.stride_h, int(p.upsample2x), int(p.C1 != p.Cin))
    return f"{device_name}|{dtype_tag}|conv|{p.B * Ho * Wo}x{p.Cout}x{p.KH * p.KW * p.Cin}|{geo}"


def _mnk(p):
    if isinstance(p, L.GemmParams):
        return p.M, p.N, p.K, bool(p.geglu)
    Hin = 2 * p.H if p.upsample2x else p.H
    Win = 2 * p.W if p.upsample2x else p.W
    Ho = (Hin + 2 * p.pad_h - p.dil_h * (p.KH - 1) - 1) // p.stride_h + 1
    Wo = (Win + 2 * p.pad_w - p.dil_w * (p.KW - 1) - 1) // p.stride_w + 1
    return p.B * Ho * Wo, p.Cout, p.KH * p.KW * p.Cin, False


def _time(fn, stream, inner=4, groups=3):
    fn()
    best = None
    for _ in range(groups):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(inner):
            fn()
        b.record(stream)
        b.synchronize()
        t = a.elapsed_time(b) / inner
        best = t if best is None or t < best else best
    return best


BATCH_INVARIANT = os.environ.get("SFAST_BATCH_INVARIANT", "0") not in ("0", "false", "off", "")
BATCH_REF = 2   # the reference batch: the CFG pair of one image (csrc/common.h g_batch_ref uses the same number)


def _ref_problem(p, batch):
    """Copy of a GEMM / conv params struct with the batch replaced by BATCH_REF, or None when the rows do not divide by the batch."""
    q = type(p).from_buffer_copy(p)
    if isinstance(p, L.GemmParams):
        if batch <= 0 or p.M % batch:
            return None
        q.M = p.M // batch * BATCH_REF
    else:
        q.B = BATCH_REF
    return q


def pin_plan_to_reference_batch(plan, device, dtype_tag="f16"):
    """SFAST_BATCH_INVARIANT=1 (opt-in): every GEMM / conv of the plan runs the (variant, split-K) that the
    SAME per-sample problem gets at the reference batch (BATCH_REF = 2) -- the packaged / measured choice when there is one, else the
    library's analytic plan for that shape -- whatever the plan's own batch is. Tile shapes of pipes 0 - 4 accumulate K in the same order
    (64-wide K-tiles, 16-wide MFMA steps, fp32), so what must not follow the batch is (a) the split-K factor (partial sums are added in
    split order) and (b) anything that changes the partition of a sample's GroupNorm statistics (tile rows of a statistics-emitting
    conv): pinning both to the reference batch makes row i of a batch-B run bit-equal to the same sample run at any other batch.
    Pipe 5 visits a conv's K-tiles channel-slice-major, i.e. in another order: a reference choice that names it is replaced by the
    analytic (pipes 0 - 2) variant of the reference shape. Costs throughput at large batches (reported by bench.py as
    `variants.batch_invariant`). Returns the number of ops pinned."""
    lib = L.load()
    with _cache_lock:
        _load_file(os.environ.get("SFAST_TUNE_CACHE"))
        if os.environ.get("SFAST_TUNE_PACKAGED", "1") not in ("0", "false", "off", ""):
            _load_file(PACKAGED_CACHE)
    devname = getattr(torch.cuda.get_device_properties(device), "gcnArchName", "gpu").split(":")[0]
    o = (C.c_int32 * 5)()
    n = 0
    for op in plan.ops:
        if op.tune is None:
            continue
        p, _ = op.tune
        M, N, K, geglu = _mnk(p)
        if M <= 16 or (not isinstance(p, L.GemmParams) and (p.Cout < 16 or p.Cin % 8)):
            continue
        q = _ref_problem(p, plan.B)
        if q is None:
            continue
        is_conv = not isinstance(p, L.GemmParams)
        key = problem_key(q, dtype_tag, devname) + ("|pk" if getattr(op, "packed", None) else "")
        hit = _cache.get(key) or _cache.get(key[:-3] if key.endswith("|pk") else key)
        v, s = (int(hit[0]), int(hit[1])) if hit is not None else (0, 0)
        Mq = _mnk(q)[0]
        if v <= 0 or s <= 0 or (v >= 50 and is_conv) or (40 <= v < 50 and not getattr(op, "packed", None)):
            # no usable reference choice: the analytic plan of the REFERENCE shape (variant and split), never of this plan's own shape
            q.variant, q.split_k = 0, (s if s > 0 else 0)
            if is_conv:
                lib.sfast_hip_conv2d_plan(C.byref(q), 0, q.split_k, o)
            else:
                lib.sfast_hip_igemm_plan(Mq, N, K, int(geglu), 0, q.split_k, C.byref(o))
            v, s = int(o[4]), int(o[2])
        p.variant, p.split_k = v, max(s, 1)
        n += 1
    return n


def tune_plan(plan, device, dtype_tag="f16", verbose=False):
    """Choose (variant, split_k) for every tunable op of `plan` in place. Returns #problems measured."""
    if BATCH_INVARIANT:
        pin_plan_to_reference_batch(plan, device, dtype_tag)
        return 0
    lib = L.load()
    cache_path = os.environ.get("SFAST_TUNE_CACHE")
    with _cache_lock:
        _load_file(cache_path)
        if os.environ.get("SFAST_TUNE_PACKAGED", "1") not in ("0", "false", "off", ""):
            _load_file(PACKAGED_CACHE)  # choices measured on an MI355X for the SD1.5 / SDXL shapes; anything else is timed here
    devname = getattr(torch.cuda.get_device_properties(device), "gcnArchName", "gpu").split(":")[0]
    todo = {}
    ext_base = {}
    extend = os.environ.get("SFAST_TUNE_EXTEND", "0") not in ("0", "false", "off", "")
    for op in plan.ops:
        if op.tune is None:
            continue
        p, _ = op.tune
        M, N, K, geglu = _mnk(p)
        if M <= 16 or (not isinstance(p, L.GemmParams) and (p.Cout < 16 or p.Cin % 8)):
            continue
        key = problem_key(p, dtype_tag, devname) + ("|pk" if getattr(op, "packed", None) else "")
        hit = _cache.get(key)
        if hit is not None:
            p.variant, p.split_k = int(hit[0]), int(hit[1])
            # SFAST_TUNE_EXTEND=1 (tools/retune_pp.py): a cached choice made before the 256-row tiles existed is timed once more against
            # them -- how sfast/engine/tune_gfx950.json was brought up to date in round 6 without re-timing 360 problems x 17 variants
            if extend and int(hit[0]) > 0 and key not in _extended and _pp_candidates(M, N, K, geglu):
                ext_base[key] = (int(hit[0]), int(hit[1]))
                todo.setdefault(key, []).append(op)
        else:
            todo.setdefault(key, []).append(op)
    if not todo:
        return 0
    ws = torch.empty(MAX_SLAB_BYTES, dtype=torch.uint8, device=device)
    L.check(lib.sfast_hip_workspace_init(ws.data_ptr(), ws.numel(), torch.cuda.current_stream(device).cuda_stream), "sfast_hip_workspace_init")
    for t in plan.pool.all:
        t.zero_()
    stream = torch.cuda.current_stream(device)
    sp = stream.cuda_stream
    o = (C.c_int32 * 5)()
    measured = 0
    for key, ops in todo.items():
        p, launch_with = ops[0].tune
        M, N, K, geglu = _mnk(p)
        ktiles = (K + 63) // 64
        wrows = 2 * N if geglu else N
        best = None
        is_conv = not isinstance(p, L.GemmParams)
        cands = GEGLU_VARIANTS if geglu else (VARIANTS + CONV_PATCH_VARIANTS if is_conv else VARIANTS)
        cands = cands + _pp_candidates(M, N, K, geglu)
        only = None
        if key in ext_base:
            _extended.add(key)
            cands, only = (ext_base[key][0],) + tuple(v for v in _pp_candidates(M, N, K, geglu) if v != ext_base[key][0]), ext_base[key]
        elif key.endswith("|pk"):
            base = _cache.get(key[:-3])
            if base is not None and int(base[0]) > 0:
                cands, only = (int(base[0]),) + PK_VARIANTS, (int(base[0]), int(base[1]))  # the known best kernel against the packed pipe
            else:
                cands = cands + PK_VARIANTS
        for v in cands:
            for s in SPLITS:
                if only is not None and v == only[0] and s != only[1] and only[1] > 0:
                    continue
                if s > 1 and ktiles // s < 2:
                    continue
                if v >= 50 and (s > 4 or -(-M // 256) * -(-N // (PP_BN[v] // 2 if geglu else PP_BN[v])) * s < PP_MIN_WGS):
                    continue  # the 256-row tiles are candidates where tiles x (a small) split-K put work on a good part of the chip
                if s > 1 and s * M * wrows * 4 > MAX_SLAB_BYTES:
                    continue
                if is_conv:
                    lib.sfast_hip_conv2d_plan(C.byref(p), v, s, o)
                else:
                    lib.sfast_hip_igemm_plan(M, N, K, int(geglu), v, s, C.byref(o))
                bm, bn, splits, _, vid = list(o)
                if splits != s or vid != v:
                    continue
                bno = bn // 2 if geglu else bn
                wgs = -(-M // bm) * -(-N // bno) * s
                if (s > 1 and wgs > 2048) or (s == 1 and wgs > 8192 and bm * bn < 128 * 128):
                    continue  # no split-K once the tiles alone fill the chip; no 64-wide tiles on problems of > 8192 of them
                p.variant, p.split_k = v, s
                if (lib.sfast_hip_conv2d_workspace_bytes if is_conv else lib.sfast_hip_gemm_workspace_bytes)(C.byref(p)) > ws.numel():
                    continue
                if launch_with(sp, ws.data_ptr(), ws.numel()) != 0:
                    continue
                k = L.last_kernel()
                if ("pp" if v >= 50 else "pk" if v >= 40 else "patch" if v >= 30 else "ws" if v >= 20 else "dma" if v >= 10 else "reg") not in k.split(",")[-1]:
                    continue  # pipe not applicable to this problem: the library substituted another
                t = _time(lambda: launch_with(sp, ws.data_ptr(), ws.numel()), stream)
                if best is None or t < best[0]:
                    best = (t, v, s)
        if best is None:
            choice = (0, 0)
        else:
            choice = (best[1], best[2])
        measured += 1
        with _cache_lock:
            _cache[key] = choice
        for op in ops:
            op.tune[0].variant, op.tune[0].split_k = choice
        if verbose:
            print(f"[sfast autotune] {key}: variant {choice[0]} split {choice[1]}" + (f" {best[0] * 1e3:.1f} us" if best else ""))
    torch.cuda.synchronize(device)
    with _cache_lock:
        _save_file(cache_path)
    return measured
