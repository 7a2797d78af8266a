"""where does a kernel spill?  Lists the scratch loads / stores of the named kernel instantiations in a hipcc -S dump,
with the basic-block label and the synchronisation instructions around them (inside or outside the iteration loop).
usage: hipcc <flags> -S --cuda-device-only -o /tmp/x.s file.hip ; python3 tools/asm_spills.py /tmp/x.s 'cgnr_resident_kernel<c32, 8, 32, 8, 2, true>' ..."""
import re
import subprocess
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2:]
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\S+:\s+; @", l)]
for i, sym in starts:
    dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
    m = re.search(r"(\w+<[^(]*>)\(", dem)
    name = (m.group(1) if m else dem).replace("HIP_vector_type<float, 2u>", "c32")
    if want and name not in want:
        continue
    j = i + 1
    while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
        j += 1
    print("==", name, "lines", j - i)
    lab = None
    for k in range(i, j):
        l = lines[k]
        if re.match(r"^\.LBB\d+_\d+:", l):
            lab = l.split(":")[0]
        s = l.strip()
        if "scratch_" in s:
            print(f"  {k - i:6d} {lab} {s}")
        elif s.startswith(("s_barrier", "s_sleep", "s_cbranch", "s_branch")) or "atomic" in s:
            print(f"  {k - i:6d} {lab}       | {s[:60]}")
