#!/usr/bin/env python3
"""profiles/traffic_latest.json <- one workload's HBM traffic per launch from a committed PMC summary (tools/pmc_summary.py output).

  tools/update_traffic.py <workload> <profiles/rNN_x_pmc_summary.txt> <streams per GPU> <round> [launches per step = 1]

A step (one NA_BatchProcessDevice call) that runs as two half-batch launches moves two dispatches' bytes: the counters are averaged per
dispatch, bench.py reports per step -> hbm_bytes_per_launch here is the per-STEP figure (dispatch average x launches per step).

Per the guide's HBM / rocprofv3 section: FETCH_SIZE and WRITE_SIZE come from their own --pmc passes, are in KB, and on gfx950 FETCH_SIZE
counts a 128-byte request of a coalesced 16-B-per-lane read as 64 bytes -> x 2.  bench.py reads the entry of its workload.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    workload, summary, streams, rnd = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    launches = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    vals = {}
    for line in open(summary):
        m = re.match(r"(\S+)\s+n=\s*\d+\s+avg=(\S+)", line)
        if m:
            vals[m.group(1)] = float(m.group(2))
    fetch, write = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    doc = {"workloads": {}}
    if os.path.exists(path):
        old = json.load(open(path))
        doc = old if "workloads" in old else {"workloads": {"standard": dict(old, streams=1024)}}
    doc["fetch_correction"] = "x2 (gfx950: FETCH_SIZE counts 128-B requests as 64 B for 16 B/lane coalesced reads)"
    doc["workloads"][workload] = {
        "round": rnd, "streams": streams,
        "source": "%s (tools/profile_round.sh -> tools/pmc_passes.sh: one rocprofv3 --pmc pass per counter group, dispatches averaged)" % os.path.relpath(os.path.abspath(summary), ROOT),
        "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "launches_per_step": launches,
        "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0 * launches,
    }
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(doc["workloads"][workload]))


if __name__ == "__main__":
    main()
