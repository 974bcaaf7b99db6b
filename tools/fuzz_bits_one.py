"""Traceback of one seed of tools/fuzz_bits.py: python tools/fuzz_bits_one.py SEED [SEED ...]"""
import os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import rt2d_loader
import test_gpu_random as t
from oracle import oracle
oracle.compile()
rt = rt2d_loader.load()
for seed in [int(v) for v in sys.argv[1:]]:
    try:
        t.test_random_environment_bit_exact(rt, oracle, seed)
        print('seed', seed, 'ok')
    except AssertionError:
        tb = traceback.format_exc().splitlines()
        print('seed', seed, '\n'.join(tb[-6:]))
