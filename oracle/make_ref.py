"""ORACLE -- test infrastructure only. Recipe for `oracle/_ref/`: the reference's OWN Triton kernels, runnable.

The reference's GroupNorm / GroupNorm+SiLU / LayerNorm / strided-copy / convolution fast paths are pure-Python Triton
kernels (no `sfast._C`), and Triton-ROCm is installed in this image, so they can run on the MI355X and pin SURVEY
section-8 rows a6-a9 / a15 against THE REFERENCE ITSELF instead of against an ATen restatement:

    /root/reference/src/sfast/triton/ops/group_norm.py   (group_norm_forward, group_norm_silu_forward :352-479)
    /root/reference/src/sfast/triton/ops/layer_norm.py   (LayerNorm.apply :273-322)
    /root/reference/src/sfast/triton/ops/copy.py         (copy :184-270)
    /root/reference/src/sfast/triton/ops/conv.py         (conv_forward :751-1046)
    /root/reference/src/sfast/triton/ops/{activation,utils}.py, /root/reference/src/sfast/utils/copy_func.py

This script packs those files, byte for byte, from where they lie under /root/reference into ONE archive,
`oracle/_ref/sfast_ref_triton.zip`, laid out as a stub package `sfast/` (empty `__init__.py`s, so neither `sfast._C` nor
the reference's `torch_ops` registration is imported); Python imports it through zipimport. `oracle/_ref/` is git-ignored
-- no reference source enters the history or the working tree as a source file -- but not gpurun-ignored, so the archive
travels to the GPU box like a built `.so`. It is executed only by `oracle/ref_triton_run.py`, in a SUBPROCESS whose
`sys.path` holds the archive and not this repository's own `sfast` package (the two share the top-level name).

Run here (needs /root/reference); on the GPU box the staged tree is used as it arrived. `__graft_entry__.build()` calls
`stage()` when /root/reference exists. Nothing outside tests/ (and the generator `tests/golden/make_golden_ref_triton.py`)
uses the staged tree.
"""
import hashlib
import json
import os
import zipfile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src/sfast"
OUT = os.path.join(HERE, "_ref")
ARCHIVE = os.path.join(OUT, "sfast_ref_triton.zip")

FILES = [
    ("triton/ops/group_norm.py", "sfast/triton/ops/group_norm.py"),
    ("triton/ops/layer_norm.py", "sfast/triton/ops/layer_norm.py"),
    ("triton/ops/copy.py", "sfast/triton/ops/copy.py"),
    ("triton/ops/conv.py", "sfast/triton/ops/conv.py"),
    ("triton/ops/activation.py", "sfast/triton/ops/activation.py"),
    ("triton/ops/utils.py", "sfast/triton/ops/utils.py"),
    ("utils/copy_func.py", "sfast/utils/copy_func.py"),
]
STUB_INITS = ["sfast/__init__.py", "sfast/triton/__init__.py", "sfast/triton/ops/__init__.py", "sfast/utils/__init__.py"]


def available() -> bool:
    if not os.path.exists(ARCHIVE):
        return False
    with zipfile.ZipFile(ARCHIVE) as z:
        names = set(z.namelist())
    return all(dst in names for _, dst in FILES)


def stage(verbose: bool = True) -> bool:
    """Pack the reference's Triton sources into oracle/_ref/sfast_ref_triton.zip. Returns False (and leaves any earlier
    archive alone) when /root/reference is absent -- the GPU box."""
    if not os.path.isdir(REF_SRC):
        if verbose:
            print(f"[oracle/_ref] {REF_SRC} absent; using the archive as it is ({'present' if available() else 'MISSING'})")
        return available()
    os.makedirs(OUT, exist_ok=True)
    manifest = {}
    with zipfile.ZipFile(ARCHIVE, "w", zipfile.ZIP_DEFLATED) as z:
        for rel in STUB_INITS:
            z.writestr(zipfile.ZipInfo(rel, date_time=(2024, 1, 1, 0, 0, 0)), "")  # stub: no sfast._C, no op registration
        for src, dst in FILES:
            with open(os.path.join(REF_SRC, src), "rb") as f:
                data = f.read()
            z.writestr(zipfile.ZipInfo(dst, date_time=(2024, 1, 1, 0, 0, 0)), data)
            manifest[dst] = {"from": os.path.join(REF_SRC, src), "sha256": hashlib.sha256(data).hexdigest()}
    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    if verbose:
        print(f"[oracle/_ref] packed {len(FILES)} reference Triton sources into {ARCHIVE}")
    return True


if __name__ == "__main__":
    sys.exit(0 if stage() else 1)
