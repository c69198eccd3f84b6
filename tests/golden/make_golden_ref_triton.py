"""Generates tests/golden/ref_triton_small.pt: OUTPUTS OF THE REFERENCE'S OWN TRITON KERNELS (GroupNorm, GroupNorm+SiLU,
LayerNorm, strided copy and -- when Triton-ROCm compiles it -- convolution) for the `small=True` rows of
`oracle/ref_cases.py`, run on an MI355X through `oracle/ref_triton_run.py` (the reference sources staged by
`oracle/make_ref.py`). Unlike the other fixtures under tests/golden/, these are NOT minted from this repository's oracle:
they are the reference's kernels' answers, so the CPU suite (`tests/test_oracle.py::test_oracle_vs_reference_triton_fixture`)
detects drift of `oracle.ops_ref` against the reference itself.

Needs a GPU and the staged archive; run on the GPU box:
    python tests/golden/make_golden_ref_triton.py --out gpurun_out/ref_triton_small.pt
then copy the file to tests/golden/ref_triton_small.pt and commit it (inputs are regenerated from seeds, only outputs are stored).
"""
import argparse
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "ref_triton_small.pt"))
    a = ap.parse_args()
    tmp = a.out + ".tmp"
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_triton_run.py"), "--out", tmp, "--small-only"],
                   check=True, env=env, cwd=ROOT)
    res = torch.load(tmp, weights_only=False)
    os.remove(tmp)
    keep = {"triton": res["triton"], "torch": res["torch"], "device": res["device"], "status": res["status"], "out": {}}
    for name, rec in res["out"].items():
        if isinstance(rec, dict):
            keep["out"][name] = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in rec.items()}
        else:
            keep["out"][name] = rec
    torch.save(keep, a.out)
    print("wrote", a.out, os.path.getsize(a.out), "bytes;", keep["status"])


if __name__ == "__main__":
    main()
