#!/bin/bash
# Per-kernel resource usage of the shipped library's sources (VGPRs, AGPRs, scratch bytes per lane, LDS, occupancy) from
# hipcc's -Rpass-analysis=kernel-resource-usage; compiles to /dev/null, touches nothing.  CPU-only (cross-compiles gfx950).
#   usage: tools/kernel_resources.sh [file.hip ...]      (default: every csrc/*.hip)   -> one line per kernel instantiation:
#   <file> <kernel> vgprs=<n> agprs=<n> scratch=<bytes/lane> lds=<bytes> occupancy=<waves/SIMD>
set -u
cd "$(dirname "$0")/../regularizedleastsquares.jl_amd/csrc" || exit 1
FLAGS=$(sed -n "s/^CXXFLAGS = //p" Makefile | sed "s/\$(ARCH)/gfx950/")
FILES=${@:-$(ls *.hip)}
for f in $FILES; do
  /opt/rocm/bin/hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -c "$f" -o /dev/null 2>&1 | python3 -c '
import re, sys
fname = sys.argv[1]
name = None
rec = {}
def flush():
    if name:
        print(fname, name, "vgprs=%s agprs=%s scratch=%s lds=%s occupancy=%s" % (rec.get("VGPRs", "?"), rec.get("AGPRs", "?"), rec.get("ScratchSize [bytes/lane]", "?"), rec.get("LDS Size [bytes/block]", "?"), rec.get("Occupancy [waves/SIMD]", "?")))
for line in sys.stdin:
    m = re.search(r"remark: [^:]*: Function Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
    if m:
        flush()
        name = m.group(1)
        rec = {}
        continue
    m = re.search(r"remark: [^:]*:\s+([A-Za-z /\[\]]+): (\d+)", line) or re.search(r"\s+([A-Za-z /\[\]]+): (\d+)\s*\[-Rpass", line)
    if m:
        rec[m.group(1).strip()] = m.group(2)
flush()
' "$f"
done
