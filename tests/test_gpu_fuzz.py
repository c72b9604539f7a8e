"""Randomised soaks inside the GPU suite: the fuzz tools of tools/experiments/ with a FIXED first seed and a bounded
time budget each, so that the driver's `pytest -m gpu` carries random scenes / flags / layouts / sizes and not only the hand-picked
cases — whole three-frame protocols, nv_taskcull in every pinned kernel form, nv_trianglecull, nv_clustercull over sizes up to tens of
millions of meshlets, and the single passes.  Every case is compared with the CPU oracle by the tool itself (exit code 1 on any difference);
the same tools run for minutes in development sessions (their headers say how)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (tool, seconds, first seed, what its summary line must show at least)
SOAKS = [("fuzz_frames", 30, 1000, r"fuzz_frames: (\d+) scenes, 0 with mismatches", 20),
         ("fuzz_taskcull", 25, 1000, r"fuzz_taskcull: (\d+) passes, 0 with mismatches", 40),
         ("fuzz_triangles", 25, 9000, r"fuzz_triangles: (\d+) scenes, \d+ slots, \d+ s, differences: none", 20),
         ("fuzz_sizes", 30, 7000, r"fuzz_sizes: (\d+) sizes \(\d+ M meshlets in total\), mismatches: 0", 8),
         ("fuzz_passes", 25, 5000, r"fuzz_passes: .* mismatches: 0 in", 0)]


@pytest.mark.parametrize("tool, seconds, seed, summary, at_least", SOAKS, ids=[s[0] for s in SOAKS])
def test_soak_against_the_oracle(tool, seconds, seed, summary, at_least):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "experiments", tool + ".py"), str(seconds), str(seed)], capture_output=True, text=True,
                       timeout=seconds + 240, cwd=ROOT)
    tail = (p.stdout[-2000:], p.stderr[-2000:])
    assert p.returncode == 0, tail
    m = re.search(summary, p.stdout)
    assert m, tail
    if at_least:
        assert int(m.group(1)) >= at_least, tail  # (the budget bought a meaningful number of random cases)
