#!/usr/bin/env python3
"""Shader clock and socket power while a workload runs back to back: tools/power_sample.py <seconds> -- <bench.py arguments ...>
Starts `python bench.py <arguments> --steps <many>` and samples rocm-smi once a second after a 6 s lead-in; prints the samples' range and
the bench line's step time.  (A step that holds the 1400 W cap is power-limited: it gets shorter by spending less energy, not fewer idle cycles.)"""
import json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
secs = float(sys.argv[1])
args = sys.argv[3:] if len(sys.argv) > 2 and sys.argv[2] == "--" else sys.argv[2:]
steps_per_s = float(os.environ.get("NA_PS_STEPS_PER_S", "25000"))
steps = int((secs + 8) * steps_per_s)
cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-host-path", "--no-exact-f32", "--no-parity-check", "--rotate", "0",
       "--steps", str(steps)] + args
p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
time.sleep(6.0)
sclk, power = [], []
t_end = time.time() + secs
while time.time() < t_end and p.poll() is None:
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    w = re.search(r"Power \(W\): ([0-9.]+)", out)
    if m and w:
        sclk.append(int(m.group(1))); power.append(float(w.group(1)))
    time.sleep(0.8)
line = p.communicate()[0].strip().splitlines()[-1]
d = json.loads(line)
print("%-60s %7.2f us/step  sclk %s MHz  power %s W  (%d samples)" % (" ".join(args) or "(headline)", d["ms_per_step"] * 1e3,
      ("%d-%d" % (min(sclk), max(sclk))) if sclk else "?", ("%.0f-%.0f" % (min(power), max(power))) if power else "?", len(sclk)))
