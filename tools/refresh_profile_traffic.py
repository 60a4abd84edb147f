#!/usr/bin/env python3
"""profiles/<tag>*_bench.json: fill roofline.traffic / frac_measured_bytes from profiles/traffic_latest.json.

A profile set's bench line is printed BEFORE the PMC passes of the same set have run, so the `traffic` it carries is the previous set's;
after tools/update_traffic.py this rewrites the three traffic fields of the committed lines from the counters of their own set
(nothing else of the line is touched).   tools/refresh_profile_traffic.py r04_p2
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0
NAMES = {"": "standard", "_cfg3": "config3", "_cfg4": "config4", "_cfg5": "config5", "_nano1024": "nano", "_feather1024": "feather"}


def main():
    tag = sys.argv[1]
    table = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))["workloads"]
    for suffix, workload in NAMES.items():
        path = os.path.join(ROOT, "profiles", "%s%s_bench.json" % (tag, suffix))
        if not os.path.exists(path) or workload not in table:
            continue
        lines = open(path).read().strip().splitlines()
        j = json.loads(lines[-1])
        te = table[workload]
        r = j["roofline"]
        r["traffic"] = te["hbm_bytes_per_launch"]
        r["traffic_source"] = "profiles/traffic_latest.json: " + te["source"] + " (filled in after the PMC passes of this set: tools/refresh_profile_traffic.py)"
        r["frac_measured_bytes"] = te["hbm_bytes_per_launch"] / (j["kernel_ms_avg"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        lines[-1] = json.dumps(j)
        open(path, "w").write("\n".join(lines) + "\n")
        print(path, round(r["traffic"] / 1e6, 1), "MB", round(r["frac_measured_bytes"], 4))


if __name__ == "__main__":
    main()
