#!/bin/bash
# Register / LDS / code-size table of every kernel in a built object or library (amdhsa metadata notes of the gfx950 code object).
#   tools/kernel_resources.sh neuralaudio_amd/csrc/build/wavenet_spec_kernels.o [name filter]
set -e
OBJ=${1:?object or shared library}
FILTER=${2:-.}
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
LLVM=/opt/rocm/lib/llvm/bin
# the device code object sits in the .hip_fatbin section as a clang offload bundle
$LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$OBJ" --output="$TMP/dev.co" --unbundle 2>/dev/null || \
	{ $LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$OBJ" "$TMP/fat.bin" && \
	  $LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$TMP/fat.bin" --output="$TMP/dev.co" --unbundle; }
$LLVM/llvm-readelf --notes "$TMP/dev.co" | python3 -c '
import sys, re
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    print("%-6s vgpr %-4s agpr %-3s sgpr %-4s spill %-3s lds %-6s  %s" % ("", g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("vgpr_spill_count"), g("group_segment_fixed_size"), name[:150]))
' | grep -E "$FILTER" || true
$LLVM/llvm-size "$TMP/dev.co" | tail -1
