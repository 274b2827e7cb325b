"""Build the C part of the oracle (oracle/kmeans_ref.c, oracle/kmeans_linear_ref.c) into oracle/_build/libwvn_oracle.so with gcc.

    python -m oracle.build_oracle

Test infrastructure only: nothing under wild_visual_navigation_amd/ loads it.  -ffp-contract=off keeps every multiply / add a
separate rounding (the fused ones are explicit fmaf calls); no -march flag: the FMA clone of the hot loops is picked at load time."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "libwvn_oracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(HERE, "kmeans_ref.c"), os.path.join(HERE, "kmeans_linear_ref.c")]
    if force or not os.path.exists(OUT) or any(os.path.getmtime(OUT) < os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        cmd = ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC"] + srcs + ["-o", OUT, "-lm"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("gcc failed on the oracle's C restatements:\n" + r.stdout + r.stderr)
    return OUT


if __name__ == "__main__":
    print("built", build(force=True))
