#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace: per-kernel calls / total / avg / min / max.

    python tools/rocpd_summary.py gpurun_out/prof/bench_results.db [--csv out.csv] [--top 40]
"""
import argparse
import re
import sqlite3


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "").replace("sfast::", "")
    name = name.replace("_Float16", "f16").replace("__bf16", "bf16")
    return name[:110]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--csv", default=None)
    ap.add_argument("--top", type=int, default=40)
    ap.add_argument("--step-marker", default=None,
                    help="kernel-name substring that ends one step (e.g. cfg_ddim): restrict to the last --steps steps")
    ap.add_argument("--steps", type=int, default=8)
    a = ap.parse_args()
    con = sqlite3.connect(a.db)
    rows = con.execute("select name, grid_x, grid_y, grid_z, workgroup_x, duration, start, end from kernels order by start").fetchall()
    if a.step_marker:
        marks = [r for r in rows if a.step_marker in r[0]]
        if len(marks) > a.steps:
            t0, t1 = marks[-a.steps - 1][7], marks[-1][7]
            rows = [r for r in rows if r[6] >= t0 and r[7] <= t1]
            # union of busy intervals (kernels may overlap across streams)
            busy, cur_s, cur_e = 0, None, None
            for r in rows:
                if cur_e is None or r[6] > cur_e:
                    if cur_e is not None:
                        busy += cur_e - cur_s
                    cur_s, cur_e = r[6], r[7]
                else:
                    cur_e = max(cur_e, r[7])
            busy += cur_e - cur_s
            n = a.steps
            print(f"# steady window: {n} steps, {(t1 - t0) / n / 1e6:.3f} ms/step wall, {busy / n / 1e6:.3f} ms/step GPU-busy (union), "
                  f"{sum(r[5] for r in rows) / n / 1e6:.3f} ms/step summed kernel time, {len(rows) / n:.0f} dispatches/step")
    rows = [r[:6] for r in rows]
    agg = {}
    for name, gx, gy, gz, wx, dur in rows:
        k = short(name)
        e = agg.setdefault(k, [0, 0, 1 << 62, 0])
        e[0] += 1
        e[1] += dur
        e[2] = min(e[2], dur)
        e[3] = max(e[3], dur)
    total = sum(e[1] for e in agg.values())
    lines = ["kernel,calls,total_us,avg_us,min_us,max_us,percent"]
    for k, e in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f'"{k}",{e[0]},{e[1] / 1e3:.1f},{e[1] / e[0] / 1e3:.2f},{e[2] / 1e3:.2f},{e[3] / 1e3:.2f},{100.0 * e[1] / total:.2f}')
    if a.csv:
        with open(a.csv, "w") as f:
            f.write("\n".join(lines) + "\n")
    print(f"# {len(rows)} dispatches, {total / 1e6:.3f} ms total kernel time")
    for ln in lines[: a.top + 1]:
        print(ln)


if __name__ == "__main__":
    main()
