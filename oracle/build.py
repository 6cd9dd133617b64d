"""Builds the oracle's C restatement (oracle/csrc/*.c -> oracle/_build/liboracle.so) with gcc."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "liboracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "csrc", "portable_rng.c")
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(src):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    # -ffp-contract=off: keep f64 arithmetic exactly as written (no FMA contraction) so the
    # normals match the reference's numba/CPython double arithmetic bit-for-bit.
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", LIB, src, "-lm"])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
