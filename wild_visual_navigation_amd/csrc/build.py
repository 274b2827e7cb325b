"""Build libwvn_hip.so (gfx950 only) in-tree with hipcc.

    python -m wild_visual_navigation_amd.csrc.build [--force]

hipcc cross-compiles without a GPU.  Objects go to csrc/_build/, the library to
wild_visual_navigation_amd/lib/libwvn_hip.so (git-ignored, shipped to the GPU box by gpurun).
Code-object v5 so that the HIP 7.0 runtime bundled with the PyTorch wheel loads what the ROCm 7.2
compiler emits; the library links libamdhip64.so.7 by soname, which resolves to the runtime torch
already loaded (one HIP runtime per process, so torch's stream handles are valid inside the library).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(PKG, "lib", "libwvn_hip.so")
SOURCES = [
    "api.hip", "gemm_bf16.hip", "gemm_a384.hip", "gemm_n384.hip", "mlp_fused.hip", "qkv_fused.hip", "gemm_proj.hip", "gemm_x3.hip", "gemm_a384_x3.hip", "gemm_n384_x3.hip", "gemm_fp8.hip", "gemm_fp8_dma.hip", "gemm_a768_fp8.hip", "fp8.hip", "gemm_f32.hip", "elementwise.hip", "attention_bf16.hip",
    "attention_x3.hip", "attention_f32.hip",
    "segments.hip", "stego.hip", "stego_linear.hip", "mlp.hip", "mlp_train.hip", "pixel_mlp.hip", "supervision.hip", "slic.hip", "wire.hip",
]
# the kernels of the 16-bit-operand speed path are compiled twice (operand.h): bf16 operands, and fp16 operands (-> <name>_f16.o)
DUAL_OPERAND = ["gemm_bf16.hip", "gemm_a384.hip", "gemm_n384.hip", "mlp_fused.hip", "qkv_fused.hip", "gemm_proj.hip", "attention_bf16.hip"]
HEADERS = ["common.h", "operand.h", "mlp_device.h", "wvn_internal.h", os.path.join("..", "..", "include", "wvn_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mcode-object-version=5", "-Wall",
         "-Wno-unused-function"]
# bit-exact integer outputs need un-fused multiply/add in the k-means kernels (see stego.hip)
EXTRA = {"stego.hip": ["-ffp-contract=off"], "stego_linear.hip": ["-ffp-contract=off"], "supervision.hip": ["-ffp-contract=off"],
         "attention_bf16.hip": ["-fno-honor-nans"] + os.environ.get("WVN_ATTN_FLAGS", "").split(),
         "attention_x3.hip": ["-fno-honor-nans"], "mlp_fused.hip": os.environ.get("WVN_MLP_FLAGS", "").split()}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _cmd_changed(obj, cmd):
    """The full compile command is part of the staleness key (WVN_ATTN_FLAGS / WVN_MLP_FLAGS builds were reused by a
    later normal build and the reverse): it is kept beside the object and compared."""
    stamp = obj + ".cmd"
    line = " ".join(cmd)
    try:
        with open(stamp) as f:
            if f.read() == line:
                return False
    except OSError:
        pass
    return True


def _write_cmd(obj, cmd):
    with open(obj + ".cmd", "w") as f:
        f.write(" ".join(cmd))


def _hazard_scanner():
    """scripts/check_store_hazard.py as a module (None when the scripts directory is absent or WVN_SKIP_HAZARD_SCREEN is set)."""
    if os.environ.get("WVN_SKIP_HAZARD_SCREEN", "0") not in ("", "0"):
        return None
    path = os.path.join(os.path.dirname(PKG), "scripts", "check_store_hazard.py")
    if not os.path.exists(path):
        return None
    import importlib.util

    spec = importlib.util.spec_from_file_location("check_store_hazard", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hdrs = [os.path.normpath(os.path.join(HERE, h)) for h in HEADERS] + [os.path.abspath(__file__)]
    hipcc = _hipcc()
    jobs = []
    units = [(s, s.replace(".hip", ".o"), []) for s in SOURCES] + \
            [(s, s.replace(".hip", "_f16.o"), ["-DWVN_OPERAND_F16=1"]) for s in DUAL_OPERAND]
    for s, o, defs in units:
        src = os.path.join(HERE, s)
        obj = os.path.join(OBJ, o)
        cmd = [hipcc] + FLAGS + EXTRA.get(s, []) + defs + ["-c", src, "-o", obj]
        if force or _stale(obj, [src] + hdrs) or _cmd_changed(obj, cmd):
            jobs.append((o, cmd))

    screen = _hazard_scanner()

    def run(job):
        name, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        log = r.stdout + r.stderr
        if r.returncode == 0 and screen is not None:
            # the gfx950 hazards LLVM does not pad (common.h: wvn_store_b128_guarded; hand-issued scalar loads) are
            # screened on the ISA of EVERY unit that is recompiled, and a hit fails the build -- not only tests/test_isa_hazards.py
            asm = cmd[-1] + ".s"
            ra = subprocess.run(cmd[:-4] + ["-S", "--cuda-device-only", cmd[-3], "-o", asm], capture_output=True, text=True)
            if ra.returncode == 0:
                hits = [f"{name}:{ln}: {st}\n    overwritten by: {nx}" for ln, st, nx in screen.scan(asm)]
                hits += [f"{name}:{ln}: {ld}\n    destination touched before the wait by: {nx}" for ln, ld, nx in screen.scan_smem(asm)]
                os.remove(asm)
                if hits:
                    # (the object must not survive: it is newer than its sources, and the next build would link it unscreened)
                    for stale in (cmd[-1], cmd[-1] + ".cmd"):
                        if os.path.exists(stale):
                            os.remove(stale)
                    return name, 1, log + "gfx950 hazard screen (scripts/check_store_hazard.py):\n" + "\n".join(hits[:20])
            else:
                # an unscreened unit is a build failure, not a log line (WVN_SKIP_HAZARD_SCREEN=1 switches the screen off knowingly)
                for stale in (cmd[-1], cmd[-1] + ".cmd"):
                    if os.path.exists(stale):
                        os.remove(stale)
                return name, 1, log + "\nhazard screen: the -S compile failed; unit not screened\n" + ra.stdout + ra.stderr
        if r.returncode == 0:
            _write_cmd(cmd[-1], cmd)
        return name, r.returncode, log

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for name, rc, log in ex.map(run, jobs):
                if verbose and log.strip():
                    print(f"[hipcc {name}]\n{log}", file=sys.stderr)
                if rc != 0:
                    raise RuntimeError(f"hipcc failed on {name}:\n{log}")
    objs = [os.path.join(OBJ, o) for _, o, _ in units]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB, {len(jobs)} objects recompiled)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
