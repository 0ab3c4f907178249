#!/usr/bin/env python3
"""Registers / scratch / LDS of every kernel in libmtm_hip.so (gfx950 code object unbundled from the fat binary,
metadata read with llvm-readelf).  Usage: tools/kernel_resources.py [path/to/libmtm_hip.so] [name filter]
The score kernel must not spill accumulators (packed K): tests/test_abi_cpu.py checks its scratch."""
import os, re, struct, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(so_path, arch="gfx950"):
    """Every gfx950 code object of the library: one offload bundle per translation unit."""
    blob = open(so_path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = blob.find(magic)
    found = []
    while pos >= 0:
        n = struct.unpack_from("<Q", blob, pos + len(magic))[0]
        q = pos + len(magic) + 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if arch in triple and size > 0:
                found.append(blob[pos + off:pos + off + size])
        pos = blob.find(magic, pos + 1)
    if not found:
        raise RuntimeError("no %s code object in %s" % (arch, so_path))
    return found


def code_object(so_path, arch="gfx950"):
    return code_objects(so_path, arch)[0]


def kernels(so_path):
    out = []
    for co in code_objects(so_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            notes = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        for blk in notes.split("- .agpr_count")[1:]:
            g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
            out.append({"name": re.search(r"\.name:\s+(\S+)", blk).group(1), "vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"),
                        "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size")})
    return out


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def loop_scratch_ops(so_path, name_filter="ncc_mfma_kernel"):
    """{kernel: scratch instructions inside the innermost loops that contain MFMAs} - the K loops.  A loop is a backward
    branch and everything between its target and itself.  A spill there means the register allocation tipped over (accumulators
    moved around the inline-asm steps)."""
    out = {}
    for co in code_objects(so_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        name, ins = None, []          # (address, opcode, first operand)

        def flush():
            if not (name and name_filter in name):
                return
            loops = []
            for addr, op, arg in ins:
                if not op.startswith("s_cbranch") and op != "s_branch":
                    continue
                off = int(arg)
                if off < 32768:
                    continue                        # forward
                target = addr + 4 + (off - 65536) * 4
                if any(o_.startswith("v_mfma") for a_, o_, _ in ins if target <= a_ <= addr):
                    loops.append((target, addr))
            # innermost MFMA loops only (the item / channel / chunk loops around them contain the whole kernel)
            inner = [l for l in loops if not any(m != l and l[0] <= m[0] and m[1] <= l[1] for m in loops)]
            out[name] = sum(1 for a_, o_, _ in ins if o_.startswith("scratch_") and any(lo <= a_ <= hi for lo, hi in inner))
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                flush()
                name, ins = m.group(1), []
                continue
            m = re.match(r"\s+(\S+)\s*([^/]*)//\s*([0-9A-Fa-f]+):", line)
            if m and name:
                ins.append((int(m.group(3), 16), m.group(1), m.group(2).split(",")[0].strip()))
        flush()
    return out


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(ROOT, "multitemplatematching-python_amd", "MTM", "libmtm_hip.so")
    flt = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ""
    ks = [k for k in kernels(so) if flt in k["name"]]
    for k in sorted(ks, key=lambda k: -k["scratch"])[:400]:
        print("%4d vgpr %4d sgpr %5d B scratch  %s" % (k["vgpr"], k["sgpr"], k["scratch"], k["name"][:150]))
    print("%d kernels; max scratch %d B" % (len(ks), max(k["scratch"] for k in ks)))
