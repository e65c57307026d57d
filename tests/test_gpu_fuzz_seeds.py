"""The wide fuzz scripts of profiles/ (thousands of random cases when run by hand on a GPU box) with their smallest seeds in
the GPU suite: HIP deflate == oracle byte for byte, HIP inflate == oracle / system zlib with identical verdicts on damaged
input, chain-kernel runs of every length, the streaming objects, the windowed reader.  Each script is a process of its own
(some set the engine's environment switches before they load it) and ends with a line `cases N mismatches M`."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,seed,cases", [
    ("fuzz_deflate_parity.py", 301, 60),
    ("fuzz_inflate_parity.py", 301, 60),
    ("fuzz_chain_runs.py", 7, 6),
    ("fuzz_compress_objects.py", 301, 40),
    ("fuzz_stream_objects.py", 301, 40),
    ("fuzz_reader_windows.py", 1, 6),
    ("fuzz_indexed_chain.py", 3, 25),        # the writer's segment index: units of every kind and size decoded side by side, files through writer and readers
    ("fuzz_small_calls.py", 5, 40),          # the one-shot calls' small paths: 16 KiB units, one-copy results, checksums beside the kernels, the small inflate path
    ("fuzz_long_matches.py", 11, 30),        # zero runs, periods and sparse bytes at levels 4-9: the dynamic programme's long-match path, multi-unit batches
])
def test_fuzz_script_smallest_seed(script, seed, cases):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", script), str(seed), str(cases)], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    m = re.search(r"cases (\d+) mismatches (\d+)", r.stdout)
    assert m, r.stdout[-2000:]
    assert int(m.group(1)) == cases and int(m.group(2)) == 0, r.stdout[-3000:]
