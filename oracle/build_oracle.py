"""Builds oracle/libmtm_cpu.so - the C++ CPU port of the reference pipeline that bench.py times as `cpu_baseline`
(TEST INFRASTRUCTURE; the product never loads it).  g++ only, no external library; AVX2 + FMA code so that the
binary built in the build container also runs on the GPU box's host."""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cpu", "mtm_cpu.cpp")
LIB = os.path.join(HERE, "libmtm_cpu.so")
STAMP = LIB + ".stamp"
FLAGS = ["-O3", "-std=c++17", "-mavx2", "-mfma", "-fcx-limited-range", "-fno-math-errno", "-fPIC", "-shared", "-pthread", "-Wall"]


def build(force=False):
    with open(SRC, "rb") as f:
        dig = hashlib.sha256(f.read() + " ".join(FLAGS).encode()).hexdigest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read().strip() == dig:
        return LIB
    subprocess.run(["g++"] + FLAGS + [SRC, "-o", LIB], check=True)
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
