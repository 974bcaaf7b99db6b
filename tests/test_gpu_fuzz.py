"""GPU: a short stretch of every fuzz tool (tools/fuzz_*.py; each is the randomised form of one parity statement and was
run over hundreds of seeds in round 3).  Each tool runs as a child process -- they are scripts -- and
must end with "0 mismatches" and without an exception in any seed."""

import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TOOLS = [('fuzz_bits.py', 900, 912),       # HIP vs oracle, bit for bit
         ('fuzz_gain.py', 600, 612),       # gain kernels vs oracle, 1-16 frequencies, both forms
         ('fuzz_maps.py', 1300, 1316),     # stored / one-pass / fused maps against each other
         ('fuzz_product.py', 900, 912),    # the product's map vs the oracle's: shape, dropped points, cells, normalised map
         ('fuzz_multi.py', 400, 410),      # several frequencies at once vs one map per frequency
         ('fuzz_packs.py', 100, 110),      # packs from overlapping unsorted casts
         ('fuzz_dist.py', 200, 216),       # distance to target: in the stepping kernel vs from stored trajectories
         ('fuzz_cutoff.py', 150, 160),     # power cut-off vs the oracle's rule
         ('fuzz_eigen.py', 60, 68),        # sweep selection, target products
         ('fuzz_ranks.py', 400, 406)]      # two ranks on the one card (gloo) against one process


@pytest.mark.parametrize('tool,first,last', TOOLS, ids=[t[0][:-3] for t in TOOLS])
def test_fuzz_tool(tool, first, last):
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', tool), str(first), str(last)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    out = p.stdout
    assert p.returncode == 0, out[-3000:]
    lines = [ln for ln in out.splitlines() if ln.startswith('seed ') or 'mismatches' in ln]
    assert lines and lines[-1].startswith(f'{last - first} seeds, 0 mismatches'), out[-3000:]
    assert not any('Error:' in ln or 'MISMATCH' in ln for ln in lines), out[-3000:]
    assert sum(ln.startswith('seed ') for ln in lines) == last - first, out[-3000:]
