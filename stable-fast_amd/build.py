#!/usr/bin/env python3
"""Build libsfast_hip.so (gfx950) in-tree with hipcc.

    python stable-fast_amd/build.py [--force] [--jobs N] [--probes]

Every csrc/*.hip is compiled to an object (in parallel) and linked into
stable-fast_amd/sfast/_lib/libsfast_hip.so. hipcc cross-compiles without a GPU.
Objects are cached by (source + headers + flags) hash so rebuilds are incremental.

--probes builds a SECOND library, sfast/_lib/libsfast_hip_probes.so, with -DSFAST_PROBES: the timing-only experiment / ablation
instantiations (results are garbage), the measured-and-never-selected LDS-patch conv pipe and the in-kernel split-K join. Only
tools/ and the tests of those candidates load it (SFAST_HIP_PROBES=1); the product library has none of that code
(sfast_hip_has_probes() == 0) and is the only one __graft_entry__.build() produces.
"""
import argparse
import concurrent.futures as cf
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
OUT_DIR = os.path.join(HERE, "sfast", "_lib")
OBJ_DIR = os.path.join(HERE, "build", "obj")
LIB = os.path.join(OUT_DIR, "libsfast_hip.so")
LIB_PROBES = os.path.join(OUT_DIR, "libsfast_hip_probes.so")

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
         "-Wno-unused-result", "-I", INCLUDE]
# Per-file additions. attention.hip: the softmax reads every S accumulator and rescales O on the VALU each tile, so MFMA
# results allocated to AGPRs cost ~150 v_accvgpr moves per 64-key tile (a quarter of the loop); the VGPR form of the MFMA
# removes them and lowers the register total (D=40: 134+32 -> 138).
EXTRA_FLAGS = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
# attention_q64.hip (one wave per SIMD, ~400 registers) is built WITHOUT that flag: the O accumulators and the MFMA operand fragments
# belong in the AGPR half of the unified file, only the S^T tiles the softmax reads must be arch VGPRs.


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def _digest(paths, extra):
    h = hashlib.sha256()
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(extra).encode())
    return h.hexdigest()[:16]


def build(force=False, jobs=None, verbose=True, probes=False):
    lib_path = LIB_PROBES if probes else LIB
    os.makedirs(OUT_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(INCLUDE, "sfast_hip.h"))
    cc = hipcc()
    tasks = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        flags = FLAGS + EXTRA_FLAGS.get(s, []) + (["-DSFAST_PROBES"] if probes else []) + os.environ.get("SFAST_EXTRA_CFLAGS", "").split()
        tag = _digest([src] + headers, flags)
        obj = os.path.join(OBJ_DIR, f"{os.path.splitext(s)[0]}.{'p.' if probes else ''}{tag}.o")
        objs.append(obj)
        if force or not os.path.exists(obj):
            tasks.append((src, obj, flags))

    def compile_one(t):
        src, obj, flags = t
        cmd = [cc] + flags + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        return src

    if tasks:
        with cf.ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 4)) as ex:
            for done in ex.map(compile_one, tasks):
                if verbose:
                    print(f"[sfast build] compiled {os.path.basename(done)}", flush=True)
    stamp = os.path.join(OBJ_DIR, "link_probes.stamp" if probes else "link.stamp")
    link_tag = _digest(objs, ["link"])
    old = open(stamp).read() if os.path.exists(stamp) else ""
    if force or tasks or not os.path.exists(lib_path) or old != link_tag:
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        with open(stamp, "w") as f:
            f.write(link_tag)
        if verbose:
            print(f"[sfast build] linked {lib_path}", flush=True)
    # drop stale objects of THIS flavour (product objects are name.<tag>.o, probe objects name.p.<tag>.o)
    keep = set(objs)
    for f in os.listdir(OBJ_DIR):
        p = os.path.join(OBJ_DIR, f)
        if f.endswith(".o") and p not in keep and ((".p." in f) == probes):
            os.remove(p)
    return lib_path


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    ap.add_argument("--probes", action="store_true", help="build libsfast_hip_probes.so (-DSFAST_PROBES) instead")
    a = ap.parse_args()
    print(build(force=a.force, jobs=a.jobs, probes=a.probes))
