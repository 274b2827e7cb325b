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
