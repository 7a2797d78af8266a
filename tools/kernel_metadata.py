"""Per-kernel resources of the SHIPPED library, read from the code objects inside librls_mi355x.so (no compiler, no GPU):
the `.hip_fatbin` section is a sequence of clang offload bundles (one per translation unit); the gfx950 entry of each is
an ELF whose AMDGPU metadata note lists every kernel with its VGPR / SGPR count, scratch (`private_segment_fixed_size`,
bytes per lane) and static LDS.  `python tools/kernel_metadata.py [lib.so]` prints the kernels that use scratch;
tests/test_kernel_resources.py gates on it.  (tools/kernel_resources.sh gives the same figures from a fresh compile.)"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "regularizedleastsquares.jl_amd", "librls_mi355x.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    short = []
    for d in out[:len(names)]:
        d = d.replace("(anonymous namespace)::", "").replace("HIP_vector_type<float, 2u>", "c32").replace("c32 >", "c32>")
        m = re.match(r"(?:void )?([\w:]+(?:<.*?>)?)\(", d)
        short.append(m.group(1) if m else d)
    return short


def kernels(lib=LIB):
    """{short demangled name: {"scratch", "vgprs", "sgprs", "lds"}}"""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run(["objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "copy.so")], check=True)
        data = open(fat, "rb").read()
        raw = {}
        for m in re.finditer(re.escape(MAGIC), data):
            bo = m.start()
            n = struct.unpack_from("<Q", data, bo + 24)[0]
            p = bo + 32
            for _ in range(n):
                off, size, tl = struct.unpack_from("<QQQ", data, p)
                p += 24
                triple = data[p:p + tl].decode()
                p += tl
                if "gfx950" not in triple or size == 0:
                    continue
                co = os.path.join(tmp, "dev.co")
                open(co, "wb").write(data[bo + off:bo + off + size])
                txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
                for blk in txt.split("- .agpr_count")[1:]:
                    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
                    raw[re.search(r"\.name:\s+(\S+)", blk).group(1)] = {
                        "scratch": g("private_segment_fixed_size"), "vgprs": g("vgpr_count"), "sgprs": g("sgpr_count"),
                        "lds": g("group_segment_fixed_size")}
    names = sorted(raw)
    return dict(zip(demangle(names), (raw[n] for n in names)))


def disassemble(pattern, lib=LIB):
    """{short demangled name: [instruction lines]} of the gfx950 kernels whose MANGLED name contains `pattern`, disassembled
    (llvm-objdump) from the code objects inside the shipped library"""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run(["objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.path.join(tmp, "copy.so")], check=True)
        data = open(fat, "rb").read()
        for m in re.finditer(re.escape(MAGIC), data):
            bo = m.start()
            n = struct.unpack_from("<Q", data, bo + 24)[0]
            p = bo + 32
            for _ in range(n):
                off, size, tl = struct.unpack_from("<QQQ", data, p)
                p += 24
                triple = data[p:p + tl].decode()
                p += tl
                if "gfx950" not in triple or size == 0 or pattern.encode() not in data[bo + off:bo + off + size]:
                    continue
                co = os.path.join(tmp, "dev.co")
                open(co, "wb").write(data[bo + off:bo + off + size])
                txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True,
                                     check=True).stdout
                cur = None
                for line in txt.split("\n"):
                    h = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                    if h:
                        cur = h.group(1) if pattern in h.group(1) and not h.group(1).startswith(("L", ".L")) else (cur if h.group(1).startswith(("L", ".L")) else None)
                        if cur is not None and cur not in out:
                            out[cur] = []
                        continue
                    if cur is not None and line.strip():
                        out[cur].append(re.sub(r"\s*//.*$", "", line.strip()))
    names = sorted(out)
    return dict(zip(demangle(names), (out[n] for n in names)))


if __name__ == "__main__":
    ks = kernels(sys.argv[1] if len(sys.argv) > 1 else LIB)
    spill = {k: v for k, v in ks.items() if v["scratch"] > 0}
    print(f"{len(ks)} kernels, {len(spill)} with scratch")
    for k, v in sorted(spill.items(), key=lambda kv: -kv[1]["scratch"]):
        print(f"  {v['scratch']:4d} B/lane  {v['vgprs']:3d} VGPRs  {k}")
