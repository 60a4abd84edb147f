#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as CSV/markdown for profiles/."""
import sqlite3
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    note = sys.argv[3] if len(sys.argv) > 3 else ""
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w") as f:
        if note:
            f.write("# " + note + "\n")
        f.write("# durations in microseconds (rocprofv3 --kernel-trace --stats, top_kernels view)\n")
        f.write("kernel,calls,total_us,avg_us,percent\n")
        for name, calls, total, avg, pct in rows:
            short = name.split("(")[0].replace("void ", "")
            f.write('"%s",%d,%.3f,%.3f,%.2f\n' % (short, calls, total, avg, pct))


if __name__ == "__main__":
    main()
