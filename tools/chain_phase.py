#!/usr/bin/env python3
"""How the two free-running half-batch chains sit in time: from a rocprofv3 --kernel-trace -f csv run of bench.py, the
start / end times of the WaveNet dispatches per HIP stream (queue), over the last N steps.

  rocprofv3 --kernel-trace -f csv -d /tmp/ct -o ct -- python bench.py --steps 400 --no-cpu-baseline --no-parity-check --no-host-path
  tools/chain_phase.py /tmp/ct 200

Prints per chain the launch period, duration and gap, and the phase of chain B's starts inside chain A's period."""
import csv
import glob
import os
import sys


def main():
    root, last = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 200
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under " + root)
    rows = []
    for path in files:
        with open(path) as f:
            for r in csv.DictReader(f):
                if "WaveNetSpecKernel" in r["Kernel_Name"]:
                    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r.get("Stream_Id", ""), r["Kernel_Name"].split("(")[0]))
    rows.sort()
    by_queue = {}
    for s, e, q, st, k in rows:
        by_queue.setdefault((q, st), []).append((s, e))
    chains = sorted(by_queue.items(), key=lambda kv: -len(kv[1]))[:2]
    print("dispatches of %s per (queue, stream): %s" % (rows[0][4], {k: len(v) for k, v in by_queue.items()}))
    tails = []
    for (q, st), v in chains:
        v = v[-last - 1:-1]
        per = [(v[i + 1][0] - v[i][0]) / 1e3 for i in range(len(v) - 1)]
        dur = [(e - s) / 1e3 for s, e in v]
        gap = [(v[i + 1][0] - v[i][1]) / 1e3 for i in range(len(v) - 1)]
        print("chain on queue %s stream %s: last %d launches: period %.2f us (min %.2f max %.2f), duration %.2f us, gap between launches %.2f us"
              % (q, st, len(v), sum(per) / len(per), min(per), max(per), sum(dur) / len(dur), sum(gap) / len(gap)))
        tails.append(v)
    if len(tails) == 2:
        a, b = tails
        period = (a[-1][0] - a[0][0]) / (len(a) - 1)
        phases = []
        for s, e in b:
            prev = [x for x in a if x[0] <= s]
            if prev:
                phases.append(((s - prev[-1][0]) % period) / period)
        phases.sort()
        n = len(phases)
        print("start of a chain-B launch inside chain A's period (0 = together, 0.5 = opposite): median %.2f, quartiles %.2f .. %.2f, min %.2f max %.2f"
              % (phases[n // 2], phases[n // 4], phases[3 * n // 4], phases[0], phases[-1]))
        both = sum(min(ea, eb) - max(sa, sb) for (sa, ea) in a for (sb, eb) in b if min(ea, eb) > max(sa, sb))
        span = max(a[-1][1], b[-1][1]) - min(a[0][0], b[0][0])
        print("time with a launch of BOTH chains on the chip: %.1f %% of the span (%.2f ms)" % (100.0 * both / span, span / 1e6))


if __name__ == "__main__":
    main()
