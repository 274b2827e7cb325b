#!/usr/bin/env python3
"""gfx950 hazard screen for the HIP sources (finding of round 2, csrc/mlp_fused.hip):

a `buffer_store_dwordx3/x4` whose soffset is an SGPR, followed within two issue slots by a VALU instruction that overwrites one of
its data VGPRs, stores corrupted data for some lanes (the store has not finished reading its data registers).  LLVM's hazard
recognizer inserts the wait states for MUBUF stores with an immediate soffset and for FLAT/global stores, but treats the
SGPR-soffset form as hazard-free, so nothing protects a raw_buffer_store_b128 whose result registers are recycled at once.

Second screen: reads of scalar-load destinations before the wait (scan_smem below).

Usage: check_store_hazard.py [file.s ...]   (no arguments: compiles every csrc/*.hip with -save-temps and scans the ISA)
Exit code 1 if any unprotected pair is found."""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

STORE = re.compile(r"^\s*buffer_store_dwordx([34])\s+(v\[(\d+):(\d+)\]|a\[(\d+):(\d+)\]),\s*\S+,\s*s\[\d+:\d+\],\s*(s\d+|m0|\S+)")
VDST = re.compile(r"^\s*(v_\w+)\s+(v\[(\d+):(\d+)\]|v(\d+))")
SKIP = re.compile(r"^\s*(;|\.|$)|:\s*(;.*)?$")


def written(line):
    m = VDST.match(line)
    if not m or m.group(1).startswith(("v_cmp", "v_cmpx", "v_readfirstlane", "v_readlane")):
        return set()
    if m.group(3) is not None:
        return set(range(int(m.group(3)), int(m.group(4)) + 1))
    return {int(m.group(5))}


def scan(path):
    lines = Path(path).read_text().splitlines()
    hits = []
    for i, ln in enumerate(lines):
        m = STORE.match(ln)
        if not m or m.group(3) is None:      # AGPR data cannot be overwritten by a following VALU op other than accvgpr_write
            continue
        if not re.fullmatch(r"s\d+|m0", m.group(7)):
            continue                          # immediate soffset: LLVM inserts the wait states itself
        data = set(range(int(m.group(3)), int(m.group(4)) + 1))
        slots, j = 0, i + 1
        while j < len(lines) and slots < 2:
            nxt = lines[j]
            j += 1
            if SKIP.search(nxt):
                continue
            if re.match(r"^\s*s_nop\s+(\d+)", nxt):
                slots += 1 + int(re.match(r"^\s*s_nop\s+(\d+)", nxt).group(1))
                continue
            if written(nxt) & data:
                hits.append((i + 1, ln.strip(), nxt.strip()))
                break
            slots += 1
    return hits


SLOAD = re.compile(r"^\s*s_(?:buffer_)?load_dword(?:x(\d+))?\s+(s\[(\d+):(\d+)\]|s(\d+)),")
SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")


def sregs(text):
    out = set()
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def scan_smem(path):
    """Second screen (round 4, csrc/stego.hip): the k-means assign kernels issue their scalar loads by hand (inline asm) and wait for them
    later; the compiler sees the destination registers as defined from the load on, and under scalar-register pressure its allocator
    has satisfied the tied operand of the wait with COPIES of the still in-flight registers in front of the wait.  No instruction
    between an s_load and the next `s_waitcnt ... lgkmcnt(0)` may touch the load's destination registers (scalar loads return out of
    order: only a zero count is a guarantee).  Tracks the loads inside inline-asm brackets, linearly over each function."""
    hits = []
    pending = {}   # first register of a pending destination range -> (set of registers, line number, text)
    in_asm = False
    for i, ln in enumerate(Path(path).read_text().splitlines()):
        if "#ASMSTART" in ln or "#ASMEND" in ln:   # only HAND-issued loads are tracked: hipcc waits for its own before every use
            in_asm = "#ASMSTART" in ln
            continue
        if SKIP.search(ln) and not re.match(r"^\s*(;|\.|$)", ln):   # a label: function entry labels reset the state
            if not ln.lstrip().startswith(".L"):
                pending = {}
            continue
        if SKIP.search(ln):
            continue
        code = ln.split(";")[0]
        if re.match(r"^\s*s_endpgm", code):
            pending = {}
            continue
        if re.match(r"^\s*s_waitcnt\b", code):
            if "lgkmcnt(0)" in code or re.match(r"^\s*s_waitcnt\s+0\s*$", code):
                pending = {}
            continue
        m = SLOAD.match(code)
        touched = sregs(code)
        if m:
            dst = set(range(int(m.group(3)), int(m.group(4)) + 1)) if m.group(3) is not None else {int(m.group(5))}
            rest = sregs(code[m.end():])          # the address operands (may overlap the load's own destination: read at issue)
            for regs, l0, t0 in list(pending.values()):
                if (rest | dst) & regs:
                    hits.append((l0, t0, ln.strip()))
            if in_asm:
                pending[min(dst)] = (dst, i + 1, ln.strip())
            continue
        for regs, l0, t0 in list(pending.values()):
            if touched & regs:
                hits.append((l0, t0, ln.strip()))
    return hits


def main():
    files = sys.argv[1:]
    tmp = None
    if not files:
        root = Path(__file__).resolve().parents[1] / "wild_visual_navigation_amd" / "csrc"
        sys.path.insert(0, str(root.parents[1]))
        from wild_visual_navigation_amd.csrc import build as b
        tmp = tempfile.mkdtemp(prefix="wvn_isa_")
        for src in b.SOURCES:
            extra = b.EXTRA.get(src, []) if hasattr(b, "EXTRA") else []
            cmd = ["/opt/rocm/bin/hipcc"] + list(b.FLAGS) + list(extra) + ["-S", "--cuda-device-only", str(root / src), "-o", f"{tmp}/{src}.s"]
            subprocess.run(cmd, check=True, capture_output=True)
            files.append(f"{tmp}/{src}.s")
    bad = 0
    for f in files:
        for ln, st, nx in scan(f):
            bad += 1
            print(f"{Path(f).name}:{ln}: {st}\n    overwritten by: {nx}")
    print(f"{len(files)} files scanned, {bad} unprotected store/overwrite pairs")
    bad2 = 0
    for f in files:
        for ln, ld, nx in scan_smem(f):
            bad2 += 1
            if bad2 <= 40:
                print(f"{Path(f).name}:{ln}: {ld}\n    destination touched before the wait by: {nx}")
    print(f"{len(files)} files scanned, {bad2} uses of in-flight scalar-load destinations")
    return 1 if bad or bad2 else 0


if __name__ == "__main__":
    sys.exit(main())
