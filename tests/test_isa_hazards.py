"""Compile-time screen of the generated gfx950 ISA (no GPU needed): no buffer_store_dwordx3/x4 with an SGPR soffset may have one
of its data VGPRs overwritten by a VALU instruction in the next two issue slots -- the hazard found in round 2 (csrc/mlp_fused.hip
header, scripts/check_store_hazard.py): LLVM pads only the immediate-soffset form, the hardware needs both."""
import importlib.util
import shutil
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


@pytest.mark.skipif(shutil.which("hipcc") is None and not Path("/opt/rocm/bin/hipcc").exists(), reason="needs hipcc")
def test_no_unprotected_store_data_overwrite(capsys):
    spec = importlib.util.spec_from_file_location("check_store_hazard", ROOT / "scripts" / "check_store_hazard.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import sys
    argv, sys.argv = sys.argv, ["check_store_hazard.py"]
    try:
        rc = mod.main()
    finally:
        sys.argv = argv
    assert rc == 0, capsys.readouterr().out


def test_scanner_flags_the_pattern(tmp_path):
    spec = importlib.util.spec_from_file_location("check_store_hazard", ROOT / "scripts" / "check_store_hazard.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = tmp_path / "bad.s"
    bad.write_text("\tbuffer_store_dwordx4 v[10:13], v159, s[28:31], s60 offen\n\tv_pk_add_f32 v[10:11], v[14:15], v[2:3]\n")
    ok1 = tmp_path / "ok1.s"
    ok1.write_text("\tbuffer_store_dwordx4 v[10:13], v159, s[28:31], s60 offen\n\ts_nop 1\n\tv_pk_add_f32 v[10:11], v[14:15], v[2:3]\n")
    ok2 = tmp_path / "ok2.s"
    ok2.write_text("\tbuffer_store_dwordx4 v[10:13], v159, s[28:31], 0 offen\n\tv_pk_add_f32 v[10:11], v[14:15], v[2:3]\n")
    ok3 = tmp_path / "ok3.s"
    ok3.write_text("\tbuffer_store_dwordx4 v[10:13], v159, s[28:31], s60 offen\n\tv_add_f32 v20, v1, v2\n\tv_add_f32 v21, v1, v2\n\tv_mov_b32 v10, v1\n")
    assert len(mod.scan(bad)) == 1
    assert mod.scan(ok1) == [] and mod.scan(ok2) == [] and mod.scan(ok3) == []


def test_scanner_flags_reads_of_in_flight_scalar_loads(tmp_path):
    """Round 4: hand-issued scalar loads (csrc/stego.hip) whose destination registers were COPIED by the register allocator in front of
    the wait -- the pattern that mislabelled 203 of 44,100 pixels in the first packed k-means kernel."""
    spec = importlib.util.spec_from_file_location("check_store_hazard", ROOT / "scripts" / "check_store_hazard.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    load = "\t;;#ASMSTART\n\ts_load_dwordx16 s[36:51], s[18:19], 0x0\n\t;;#ASMEND\n"
    wait = "\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\t;;#ASMEND\n"
    bad = tmp_path / "bad.s"
    bad.write_text(load + "\tv_pk_fma_f32 v[8:9], v[40:41], s[52:53], v[8:9]\n\ts_mov_b64 s[66:67], s[50:51]\n" + wait)
    ok1 = tmp_path / "ok1.s"
    ok1.write_text(load + "\tv_pk_fma_f32 v[8:9], v[40:41], s[52:53], v[8:9]\n" + wait + "\ts_mov_b64 s[66:67], s[50:51]\n")
    ok2 = tmp_path / "ok2.s"   # the compiler's own loads are not tracked (it waits before every use, on every path)
    ok2.write_text("\ts_load_dwordx4 s[28:31], s[0:1], 0x10\n\ts_cbranch_scc1 .LBB0_2\n.LBB0_1:\n\ts_waitcnt lgkmcnt(0)\n.LBB0_2:\n\ts_mov_b32 s2, s28\n")
    ok3 = tmp_path / "ok3.s"   # a count above zero guarantees nothing for scalar loads: still pending
    ok3.write_text(load + "\ts_waitcnt lgkmcnt(1)\n\ts_mov_b32 s2, s40\n")
    assert len(mod.scan_smem(bad)) == 1
    assert mod.scan_smem(ok1) == [] and mod.scan_smem(ok2) == []
    assert len(mod.scan_smem(ok3)) == 1
