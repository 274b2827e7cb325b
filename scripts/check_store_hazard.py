#!/usr/bin/env python3
"""gfx950 hazard screen for the HIP sources (finding of round 2, csrc/mlp_fused.hip):

a `buffer_store_dwordx3/x4` whose soffset is an SGPR, followed within two issue slots by a VALU instruction that overwrites one of
its data VGPRs, stores corrupted data for some lanes (the store has not finished reading its data registers).  LLVM's hazard
recognizer inserts the wait states for MUBUF stores with an immediate soffset and for FLAT/global stores, but treats the
SGPR-soffset form as hazard-free, so nothing protects a raw_buffer_store_b128 whose result registers are recycled at once.

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
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
