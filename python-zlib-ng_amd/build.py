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


def build_host_asan(out):
    """The library with its HOST side under AddressSanitizer + UndefinedBehaviorSanitizer (hipcc applies -fsanitize to the host
    compilation only for a plain gfx950 target and says so; the device code is the ordinary one), for the entry points that need
    no GPU (tests/test_cpu_sanitizers.py; the reference's tox.ini:23-30 does the same for its extension).  GPU-side sanitizers
    are not available on this pool."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-Wno-option-ignored", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-o", out] + [os.path.join(CSRC, s) for s in SOURCES]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    if "--host-asan" in sys.argv:
        print(build_host_asan(sys.argv[sys.argv.index("--host-asan") + 1]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
