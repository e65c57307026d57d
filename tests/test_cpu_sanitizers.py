"""CPU builds under AddressSanitizer + UndefinedBehaviorSanitizer (what the reference does for its extension in tox.ini:23-30):
(1) the oracle's C restatement runs its pinning suite (tests/test_oracle_pinning.py) as a sanitizer build;
(2) the host side of the product library (hipcc -fsanitize=...: host compilation only for gfx950) is exercised through every entry point that
    needs no GPU -- CRC folding, the placement rule of the exchange, unit counting, argument checks of the stream API.
Each runs in a child process with the sanitizer runtime preloaded; any report aborts the child (exit code != 0)."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import PKG_DIR, ROOT

ASAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=77", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}


def _run(preload, extra_env, argv, timeout=900):
    env = dict(os.environ, LD_PRELOAD=preload, **ASAN_ENV, **extra_env)
    env.pop("ZNGAMD_LIB", None) if "ZNGAMD_LIB" not in extra_env else None
    return subprocess.run(argv, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_oracle_pinning_suite_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "oracle", "libza_oracle_asan.so")
    runtime = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(runtime) or not os.path.exists(runtime):
        pytest.skip("no shared libasan for gcc on this host")
    r = _run(runtime, {"ZA_ORACLE_SO": so}, [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                                             os.path.join(ROOT, "tests", "test_oracle_pinning.py")])
    assert r.returncode == 0, (r.returncode, r.stdout[-3000:], r.stderr[-3000:])
    assert "passed" in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr


HOST_SCRIPT = textwrap.dedent("""
    import ctypes as C, os, sys, zlib, random
    sys.path.insert(0, %r)
    from zlib_ng_amd import _lib
    assert _lib.LIB_PATH == os.environ["ZNGAMD_LIB"]
    L = _lib.load()
    assert L.zngamd_version().startswith(b"zng_amd")
    # crc32_combine: three-integer GF(2) arithmetic, all lengths incl. 0 and > 4 GiB
    rnd = random.Random(5)
    blob = bytes(rnd.getrandbits(8) for _ in range(5000))
    for cut in (0, 1, 7, 2500, 4999, 5000):
        a, b = blob[:cut], blob[cut:]
        assert L.zngamd_crc32_combine(zlib.crc32(a), zlib.crc32(b), len(b)) == zlib.crc32(blob)
    L.zngamd_crc32_combine.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
    L.zngamd_crc32_combine(0x12345678, 0x9abcdef0, (1 << 40) + 12345)
    # placement rule of the exchange
    L.zngamd_comm_offsets.restype = C.c_uint64
    L.zngamd_comm_offsets.argtypes = [C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_uint64)]
    sizes = (C.c_uint64 * 8)(5, 0, 7, 1 << 33, 3, 0, 0, 9)
    offs = (C.c_uint64 * 8)()
    tot = L.zngamd_comm_offsets(sizes, 8, offs)
    assert tot == sum(sizes) and list(offs) == [sum(list(sizes)[:r]) for r in range(8)]
    assert L.zngamd_comm_offsets(None, 0, None) == 0
    # unit counting and level checks
    blocks = (_lib.Block * 4)(_lib.Block(0, 0, 0, 0, 0), _lib.Block(0, 131072, 0, 0, 0), _lib.Block(0, 131073, 0, 0, 0),
                              _lib.Block(0, 0xFFFFFFFF, 32768, 0, 0))
    assert L.zngamd_count_units(blocks, 4) == 1 + 1 + 2 + 32768
    assert L.zngamd_count_units(None, 0) == 0
    assert [L.zngamd_level_ok(v) for v in (-2, -1, 0, 9, 10)] == [0, 1, 1, 1, 0]
    # no GPU: contexts fail, nothing is dereferenced
    assert L.zngamd_device_count() == 0
    h = C.c_void_p()
    assert L.zngamd_ctx_create(0, C.byref(h)) != 0 and not h.value
    # the stream API rejects what zng_*Init2 rejects before it touches a context
    from zlib_ng_amd import zlib_ng
    zst = zlib_ng._ZStream()
    S = zlib_ng._slib()
    assert S.zngamd_stream_deflate_init(None, C.byref(zst), 6, 8, 15, 8, 0) == _lib.STREAM_ERROR
    assert S.zngamd_stream_inflate_init(None, C.byref(zst), 15) == _lib.STREAM_ERROR
    assert S.zngamd_stream_deflate(C.byref(zst), 0) == _lib.STREAM_ERROR
    assert S.zngamd_stream_inflate(C.byref(zst), 0) == _lib.STREAM_ERROR
    assert S.zngamd_stream_deflate_reset(C.byref(zst)) == _lib.STREAM_ERROR
    assert S.zngamd_stream_inflate_reset(C.byref(zst)) == _lib.STREAM_ERROR
    assert S.zngamd_stream_deflate_end(C.byref(zst)) == _lib.STREAM_ERROR
    # the launcher-free hand-over of the communicator id (Python + sockets; runs here so that the preloaded runtime sees it too)
    import threading
    from zlib_ng_amd import shard
    port = int(os.environ["SAN_PORT"])
    got = {}
    def peer():
        got[1] = shard.rendezvous_bytes(1, 2, "127.0.0.1", port, None)
    t = threading.Thread(target=peer); t.start()
    got[0] = shard.rendezvous_bytes(0, 2, "127.0.0.1", port, bytes(range(128)))
    t.join()
    assert got[0] == got[1] == bytes(range(128))
    print("host entry points clean")
""")


def test_host_entry_points_under_asan_ubsan(tmp_path):
    import importlib.util
    import socket
    spec = importlib.util.spec_from_file_location("zng_amd_build_asan", os.path.join(PKG_DIR, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    so = str(tmp_path / "libzng_amd_host_asan.so")
    mod.build_host_asan(so)
    clang = os.path.join(os.path.dirname(os.path.realpath(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))), "..", "lib", "llvm", "bin", "clang")
    if not os.path.exists(clang):
        clang = "/opt/rocm/lib/llvm/bin/clang"
    runtime = subprocess.run([clang, "--print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(runtime) or not os.path.exists(runtime):
        pytest.skip("no shared AddressSanitizer runtime for hipcc's clang on this host")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = _run(runtime, {"ZNGAMD_LIB": so, "SAN_PORT": str(port), "CUDA_VISIBLE_DEVICES": "", "HIP_VISIBLE_DEVICES": ""},
             [sys.executable, "-c", HOST_SCRIPT % PKG_DIR])
    assert r.returncode == 0, (r.returncode, r.stdout[-3000:], r.stderr[-3000:])
    assert "host entry points clean" in r.stdout
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
