#!/usr/bin/env python3
"""Detector for the code-generation defect behind round 5's state-dependent uint16 scores (profiles/r06_flake/):

the register allocator of this ROCm's LLVM places VGPR spill stores (and reloads) at the top of a JOIN block *ahead of* the
`s_or_b64 exec, exec, s[a:b]` that re-opens the execution mask there.  The spill then runs under the mask of the divergent
region that just ended: only that region's lanes save their register, every lane reloads it later, and the others get
whatever the scratch backing store held - memory left behind by earlier launches of the process.

    .LBB6_126:                                  ; join block of `if (lane_on) { ... }`
        v_writelane_b32 v241, s50, 26
        scratch_store_dwordx4 off, v[232:235], off offset:88 ; 16-byte Folded Spill     <- lanes of the if-body only
        s_or_b64 exec, exec, s[18:19]                                                  <- mask restored HERE

The scan works on the device assembly hipcc leaves next to every object (`-save-temps=obj`, what build.py compiles with):
for each exec-restoring instruction it walks back to the start of its basic block and reports every spill / reload / AGPR copy
in between.  A value that is private to the lanes of the region survives this; an accumulator another lane will read does not -
the scan does not try to tell them apart: no such placement may exist in any kernel of the library
(tests/test_abi_cpu.py::test_no_spill_ahead_of_an_exec_restore).

Usage: tools/spill_exec_scan.py [file.s ...]      (default: every device .s under csrc/build/)"""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "multitemplatematching-python_amd", "csrc", "build")

_FUNC = re.compile(r"^([A-Za-z_][\w.$]*):\s*(;.*)?$")
_BLOCK = re.compile(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)")
# instructions that widen the execution mask again (end of a divergent region / loop exit)
_RESTORE = re.compile(r"^\s*(s_or_b64 exec, exec, s\[\d+:\d+\]|s_or_saveexec_b64 s\[\d+:\d+\], s\[\d+:\d+\]|s_mov_b64 exec, s\[\d+:\d+\])")
_SPILL = re.compile(r"^\s*(scratch_store|scratch_load|buffer_store|buffer_load|v_accvgpr_write|v_accvgpr_read)\S*\s.*")
_BRANCH = re.compile(r"^\s*(s_cbranch|s_branch|s_endpgm|s_setpc)")


_VREG = re.compile(r"\bv(?:\[(\d+):(\d+)\]|(\d+))")


def _vregs(operand_text):
    regs = set()
    for m in _VREG.finditer(operand_text):
        if m.group(3) is not None:
            regs.add(int(m.group(3)))
        else:
            regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return regs


def _split(text):
    """(mnemonic, registers written, registers read) of one instruction line - first operand = destination, except for stores"""
    body = text.split(";")[0].strip()
    if not body or body.endswith(":"):
        return "", set(), set()
    parts = body.split(None, 1)
    mn = parts[0]
    ops = parts[1].split(",") if len(parts) > 1 else []
    if mn.startswith(("scratch_store", "buffer_store", "global_store", "flat_store", "ds_write", "ds_store")) or mn.startswith("s_"):
        return mn, set(), _vregs(",".join(ops))
    return mn, _vregs(ops[0]) if ops else set(), _vregs(",".join(ops[1:]))


def _offenders(block):
    """Spill code of a basic block that sits ahead of the block's exec restore AND moves a value across it:
    a spill STORE of a register the block has not written itself (the value comes from outside, every lane owns one, only the
    region's lanes save it), or a RELOAD nothing reads before the restore (it is meant for the lanes behind it)."""
    bad = []
    for k, (ln, t) in enumerate(block):
        if not _SPILL.match(t):
            continue
        mn, wr, rd = _split(t)
        if "store" in mn or mn.startswith("v_accvgpr_write"):
            src = rd if "store" in mn else rd
            written_here = set()
            for _, t2 in block[:k]:
                written_here |= _split(t2)[1]
            if src and not (src & written_here):
                bad.append((ln, t.strip()))
        else:
            used = False
            for _, t2 in block[k + 1:]:
                _, wr2, rd2 = _split(t2)
                if wr & rd2:
                    used = True
                    break
            if not used:
                bad.append((ln, t.strip()))
    return bad


def device_asm_files():
    return sorted(glob.glob(os.path.join(BUILD, "*-hip-amdgcn-amd-amdhsa-gfx950.s")))


def scan(path):
    """[(kernel, line number of the restore, restore text, [offending lines])]"""
    found = []
    func = None
    block = []              # (line number, text) since the start of the current basic block
    with open(path) as fh:
        lines = [l.rstrip("\n") for l in fh]
    if True:
        for ln, text in enumerate(lines, 1):
            m = _FUNC.match(text)
            if m and not text.startswith("."):
                func = m.group(1)
                block = []
                continue
            if func is None:
                continue
            if text.startswith(".Lfunc_end"):
                func = None
                continue
            if _BLOCK.match(text):
                block = []
                continue
            if _RESTORE.match(text):
                # `s_mov_b64 exec, sX` + `s_cbranch_execz` is the ENTRY of a region (the mask narrows): what sits ahead of it
                # ran under the wider mask
                nxt = next((t for t in lines[ln:ln + 4] if t.strip() and not t.lstrip().startswith(";")), "")
                if text.lstrip().startswith("s_mov_b64") and re.match(r"^\s*s_cbranch_exec", nxt):
                    block = []
                    continue
                bad = _offenders(block)
                if bad:
                    found.append((func, ln, text.strip(), bad))
                # what follows the restore runs under the wider mask: start over
                block = []
                continue
            if _BRANCH.match(text):
                block = []
                continue
            block.append((ln, text))
    return found


def scan_all(paths=None):
    paths = paths or device_asm_files()
    out = []
    for p in paths:
        out.extend((os.path.basename(p),) + f for f in scan(p))
    return paths, out


if __name__ == "__main__":
    files, hits = scan_all(sys.argv[1:] or None)
    if not files:
        sys.exit("no device assembly found (build with multitemplatematching-python_amd/build.py first)")
    for fname, func, ln, restore, bad in hits:
        print("%s:%d  %s\n    in %s" % (fname, ln, restore, func))
        for l, t in bad:
            print("      %6d  %s" % (l, t))
    print("%d file(s), %d exec restore(s) with spill code ahead of them" % (len(files), len(hits)))
    sys.exit(1 if hits else 0)
