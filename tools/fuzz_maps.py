"""One-off extension of tests/test_gpu_random.py::test_random_environment_heat_maps_agree to more seeds:
python tools/fuzz_maps.py [first] [last].  Stored / one-pass / fused maps of random environments must agree to 1e-12."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import rt2d_loader
import test_gpu_random as t
rt = rt2d_loader.load()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 8
last = int(sys.argv[2]) if len(sys.argv) > 2 else 72
bad = []
for seed in range(first, last):
    try:
        t.test_random_environment_heat_maps_agree(rt, seed)
    except AssertionError as e:
        bad.append(seed); print(f'seed {seed}: MISMATCH {e}', flush=True)
    except Exception as e:                                   # (an environment the generator makes invalid: reported, not counted)
        print(f'seed {seed}: {type(e).__name__}: {str(e)[:120]}', flush=True)
print(f'{last - first} seeds, {len(bad)} mismatches {bad}')
