#!/usr/bin/env python3
"""Reduce a rocprofv3 --pmc rocpd database to a small JSON: per (kernel, grid) mean counter values."""
import collections
import json
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"\(.*$", "", n).replace("void ", "").replace("sfast::", "")
    return n[:90]


def main(db, out, by_symbol=False, last_frac=1.0):
    con = sqlite3.connect(db)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info('counters_collection')")]
    sel = "select dispatch_id, kernel_name, counter_name, value, grid_size_x, grid_size_y, workgroup_size_x from counters_collection"
    rows = [r for r in cur.execute(sel) if "sfast" in r[1]]
    ids = sorted({r[0] for r in rows})
    cut = ids[int(len(ids) * (1.0 - last_frac))] if ids and last_frac < 1.0 else -1
    name_of = (lambda n: re.sub(r"\(.*$", "", n).replace(" [clone .kd]", "").strip()) if by_symbol else short
    d = collections.OrderedDict()
    for did, kn, cn, val, gx, gy, wx in rows:
        if did < cut:
            continue
        key = (name_of(kn), 0, 0) if by_symbol else (short(kn), gx // max(wx, 1), gy)
        e = d.setdefault(key, {})
        per = e.setdefault(did, {})
        per[cn] = per.get(cn, 0) + val
    res = []
    for (kn, gx, gy), per in d.items():
        agg = collections.defaultdict(float)
        for did, cs in per.items():
            for c, v in cs.items():
                agg[c] += v
        n = len(per)
        res.append(dict(kernel=kn, grid_x=gx, grid_y=gy, dispatches=n, counters={c: v / n for c, v in agg.items()}))
    # kernel durations from the trace in the same db
    dur = collections.defaultdict(list)
    for name, gx, gy, wx, du in cur.execute("select name, grid_x, grid_y, workgroup_x, duration from kernels"):
        if "sfast" in name:
            dur[(name_of(name), 0, 0) if by_symbol else (short(name), gx // max(wx, 1), gy)].append(du)
    for r in res:
        ds = dur.get((r["kernel"], r["grid_x"], r["grid_y"]), [])
        r["avg_us"] = sum(ds) / len(ds) / 1e3 if ds else None
    json.dump(dict(columns=cols, rows=res), open(out, "w"), indent=1)


if __name__ == "__main__":
    argv = sys.argv[1:]
    by_symbol = "--by-symbol" in argv
    frac = 1.0
    if "--last-frac" in argv:
        frac = float(argv[argv.index("--last-frac") + 1])
    pos = [a for i, a in enumerate(argv) if not a.startswith("--") and (i == 0 or argv[i - 1] != "--last-frac")]
    main(pos[0], pos[1], by_symbol, frac)
