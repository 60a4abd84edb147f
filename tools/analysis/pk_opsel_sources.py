"""Static check for the gfx950 hazard of profiles/r06_quad_race.txt: v_pk_*_f32 with a non-default op_sel / op_sel_hi on a source register
whose LAST WRITER (scanning the disassembly backwards, same kernel) is a memory instruction (ds_read*, global / buffer / scratch load)
rather than a VALU instruction.  Linear scan, no control-flow graph: a report is a place to look at, not a proof; no report for a kernel
means every such operand was last written by ALU / MFMA / accumulator-read instructions on the straight-line path in front of it.
The probe (tools/microbench/pk_lds_opsel.hip) shows the hazard for registers delivered by ds_read_b128 and global_load_dwordx4, not for
ds_read_b32 / b64, and not for v_fma_mix_f32 (the other op_sel user of the library, which IS fed loaded registers: probed clean); the scan
reports every memory-delivered packed-f32 operand all the same -- the library has none (tests/test_host_cpu.py).
usage: python tools/analysis/pk_opsel_sources.py <libNeuralAudioCAPI.so | disassembly.s> ...   (.s: llvm-objdump -d of a gfx950 code object)"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile
from collections import Counter

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

REG = re.compile(r"\bv(?:\[(\d+):(\d+)\]|(\d+))")


def regs(tok):
    m = REG.search(tok)
    if not m:
        return None
    if m.group(3) is not None:
        return (int(m.group(3)), int(m.group(3)))
    return (int(m.group(1)), int(m.group(2)))


def parse(line):
    # "\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[8:9] op_sel:[0,1,0] op_sel_hi:[1,0,1]   // 0000..: ..."
    body = line.split("//")[0].strip()
    if not body or body.endswith(":"):
        return None
    parts = body.split(None, 1)
    op = parts[0]
    rest = parts[1] if len(parts) > 1 else ""
    mods = {}
    for name in ("op_sel_hi", "op_sel"):
        m = re.search(name + r":\[([0-9,]+)\]", rest)
        if m:
            mods[name] = [int(x) for x in m.group(1).split(",")]
            rest = rest.replace(m.group(0), "")
    rest = re.sub(r"\b\w+:\[[0-9,]+\]", "", rest)  # neg_lo / neg_hi / ...
    ops = [o.strip() for o in rest.split(",")] if rest.strip() else []
    return op, ops, mods


def scan_disassembly(lines):
    """-> (operands with a non-default op_sel per kernel, {(kernel, writer): count} for the memory-delivered ones)"""
    kernel, hist = None, []
    found, total = Counter(), Counter()
    for line in lines:
        m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
        if m:
            kernel, hist = m.group(1), []
            continue
        p = parse(line)
        if p is None:
            continue
        op, ops, mods = p
        if re.match(r"v_pk_(fma|mul|add)_f32", op):
            nsrc = len(ops) - 1
            sel = (mods.get("op_sel", []) + [0] * nsrc)[:nsrc]
            hi = (mods.get("op_sel_hi", []) + [1] * nsrc)[:nsrc]
            for i in range(nsrc):
                if sel[i] == 0 and hi[i] == 1:
                    continue
                r = regs(ops[1 + i]) if ops[1 + i].startswith("v") else None
                if r is None:
                    continue
                total[kernel] += 1
                for prev_op, prev_dst in reversed(hist):
                    if prev_dst and not (prev_dst[1] < r[0] or prev_dst[0] > r[1]):
                        if re.match(r"(ds_read|ds_load|global_load|buffer_load|scratch_load|flat_load)", prev_op):
                            found[(kernel, prev_op)] += 1
                        break
        dst = regs(ops[0]) if ops and ops[0].startswith("v") and not op.startswith(("ds_write", "global_store", "buffer_store", "scratch_store", "ds_store", "v_cmp")) else None
        hist.append((op, dst))
        if len(hist) > 4000:
            del hist[:2000]
    return total, found


def scan_library(so_path):
    """every gfx950 code object of a shared library -> (code objects, total, found) as scan_disassembly, summed"""
    tmp = tempfile.mkdtemp(prefix="pk_opsel_")
    try:
        lib = os.path.join(tmp, os.path.basename(so_path))
        shutil.copy(so_path, lib)  # (--offloading writes the code objects beside the file it reads)
        subprocess.run([OBJDUMP, "--offloading", lib], cwd=tmp, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        total, found, n = Counter(), Counter(), 0
        for co in sorted(glob.glob(lib + ".*gfx950*")):
            d = subprocess.run([OBJDUMP, "-d", co], check=True, capture_output=True, text=True).stdout
            t, f = scan_disassembly(d.splitlines())
            total.update(t)
            found.update(f)
            n += 1
        return n, total, found
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main(paths):
    for path in paths:
        if path.endswith(".so"):
            n, total, found = scan_library(path)
            print("%s: %d code objects" % (path, n))
        else:
            total, found = scan_disassembly(open(path))
        print("%s: %d packed-f32 operands with a non-default op_sel in %d kernels" % (path, sum(total.values()), len(total)))
        for (k, w), c in sorted(found.items()):
            print("   %5d x last written by %-22s in %s" % (c, w, k[:110]))
        if not found:
            print("   none of them last written by a memory instruction")


if __name__ == "__main__":
    main(sys.argv[1:])
