#!/usr/bin/env python3
"""Put the per-symbol durations of a --kernel-trace-ONLY pass (tools/rocpd_summary.py CSV of the steady window of graph replays) into a
PMC traffic file as `avg_us_trace` / `launches_trace`, and say so in its `_meta` block.

Why a separate step in round 5: the two sessions that produced the clean counter passes (`tools/gpu_r5.sh final2`) ran ~20 % slow end to
end (150 it/s where every other session of the round measured 180 - 186; profiles/r05_final2_slow_sessions_run7_8.log) -- bytes per
launch do not depend on the clock, durations do. The trace-only passes of the next session (`final3`, 186.2 it/s) supply the durations.

    python tools/merge_trace_avg.py profiles/r05_pmc_traffic_by_symbol.json gpurun_out/trace_sd15.csv "final3 session (186.2 it/s box)"
"""
import csv
import json
import sys


def main(path, csv_path, note):
    doc = json.load(open(path))
    trace = {}
    with open(csv_path) as fh:
        for row in csv.DictReader(l for l in fh if not l.startswith("#")):
            trace[row["kernel"]] = (float(row["avg_us"]), int(row["calls"]))
    n = 0
    for k, v in doc.items():
        if k == "_meta":
            continue
        t = trace.get(k[:110])   # rocpd_summary.py cuts names at 110 characters
        if t:
            if "avg_us_trace" in v:
                v["avg_us_trace_same_session_as_counters"] = v["avg_us_trace"]
            v["avg_us_trace"], v["launches_trace"] = t
            n += 1
    doc["_meta"]["avg_us_trace_from"] = note
    json.dump(doc, open(path, "w"), indent=1)
    print(f"{path}: {n} symbols updated")


if __name__ == "__main__":
    main(*sys.argv[1:4])
