"""Build libzng_amd.so (the C-ABI HIP library) in-tree for gfx950.

    python build.py            # build if sources are newer than the library
    python build.py --force

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "zlib_ng_amd", "libzng_amd.so")
SOURCES = ["zng_amd.hip"]
DEPS = ["zng_amd.hip", "zng_stream.hip", "za_common.h", "za_crc.h", "za_deflate.hip", "za_inflate.hip", "za_checksum.hip",
        os.path.join("..", "..", "include", "zng_amd.h")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
           "-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
