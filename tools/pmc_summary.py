#!/usr/bin/env python3
"""Average PMC counter values per dispatch of one kernel from tools/pmc_passes.sh output."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "WaveNetSpecKernel"
    acc = defaultdict(list)
    for f in sorted(glob.glob(os.path.join(root, "pass*", "*counter_collection.csv"))):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if pat in row["Kernel_Name"]:
                    acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in sorted(acc):
        v = acc[k]
        print("%-32s n=%3d avg=%.4g" % (k, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
