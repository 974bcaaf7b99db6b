"""One-off extension of tests/test_gpu_random.py::test_random_environment_bit_exact to more seeds:
python tools/fuzz_bits.py [first] [last].  HIP against the oracle (test infrastructure) on random environments: every
integer and every stored double identical."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import rt2d_loader
import test_gpu_random as t
from oracle import oracle
oracle.compile()
rt = rt2d_loader.load()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 12
last = int(sys.argv[2]) if len(sys.argv) > 2 else 112
bad = []
for seed in range(first, last):
    try:
        t.test_random_environment_bit_exact(rt, oracle, seed)
    except AssertionError as e:
        bad.append(seed); print(f'seed {seed}: MISMATCH {str(e)[:300]}', flush=True)
    except Exception as e:
        print(f'seed {seed}: {type(e).__name__}: {str(e)[:160]}', flush=True)
print(f'{last - first} seeds, {len(bad)} mismatches {bad}')
