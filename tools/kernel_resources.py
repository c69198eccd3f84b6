#!/usr/bin/env python3
"""Static resource table of the library's kernels from hipcc's assembly listings (`hipcc -S --offload-device-only`):
VGPR / AGPR / SGPR, static LDS, scratch (spill) bytes, code size, and the occupancy the register count allows on gfx950
(512 VGPR+AGPR per SIMD lane slot). usage: kernel_resources.py file.s [...]  -- see the header of profiles/r01_kernel_resources.txt"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
        return out.splitlines()
    except OSError:
        return names


def main(paths):
    rows = []
    for p in paths:
        txt = open(p).read()
        sizes = {m.group(1): int(m.group(2)) for m in re.finditer(r"^; (\S+) codeLenInByte = (\d+)", txt, re.M)}
        # codeLenInByte comments sit inside each function; map them by order of appearance of the kernel labels
        for blk in re.finditer(r"\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel", txt, re.S):
            name, body = blk.group(1), blk.group(2)
            f = lambda k: int(re.search(rf"\.amdhsa_{k} (\d+)", body).group(1)) if re.search(rf"\.amdhsa_{k} (\d+)", body) else 0
            nv = f("next_free_vgpr")
            acc = f("accum_offset")
            rows.append(dict(name=name, vgpr_total=nv, arch_vgpr=acc if acc else nv, sgpr=f("next_free_sgpr"),
                             lds=f("group_segment_fixed_size"), scratch=f("private_segment_fixed_size")))
        for m in re.finditer(r"^\s*\.type\s+(\S+),@function.*?; codeLenInByte = (\d+)", txt, re.S | re.M):
            sizes[m.group(1)] = int(m.group(2))
        for r in rows:
            r.setdefault("code", sizes.get(r["name"], 0))
    names = demangle([r["name"] for r in rows])
    print(f"{'VGPR+AGPR':>9} {'archV':>5} {'SGPR':>4} {'LDS(static)':>11} {'scratch':>7} {'code B':>7} {'waves/SIMD':>10}  kernel")
    for r, n in sorted(zip(rows, names), key=lambda t: t[1]):
        occ = min(8, 512 // max(r["vgpr_total"], 1))
        n = re.sub(r"\(.*$", "", n).replace("sfast::", "").replace("void ", "")
        print(f"{r['vgpr_total']:9d} {r['arch_vgpr']:5d} {r['sgpr']:4d} {r['lds']:11d} {r['scratch']:7d} {r['code']:7d} {occ:10d}  {n}")


if __name__ == "__main__":
    main(sys.argv[1:])
