"""Tools that time experiment / ablation instantiations or the never-selected candidates (patch conv pipe, split-K join) run
against the PROBE build of the library: `use_probe_build()` before the first `sfast.hip` import builds
stable-fast_amd/sfast/_lib/libsfast_hip_probes.so (-DSFAST_PROBES) if it is missing and makes `sfast.hip.lib` load it.
The product library (libsfast_hip.so) contains none of that code."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def use_probe_build(build_if_missing=True):
    os.environ["SFAST_HIP_PROBES"] = "1"
    lib = os.path.join(ROOT, "stable-fast_amd", "sfast", "_lib", "libsfast_hip_probes.so")
    if build_if_missing and not os.path.exists(lib):
        spec = importlib.util.spec_from_file_location("sfast_build", os.path.join(ROOT, "stable-fast_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(verbose=True, probes=True)
    return lib
