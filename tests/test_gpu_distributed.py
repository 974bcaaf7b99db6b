"""GPU: the product's distributed mode with two ranks.  The GPU box has one card, so both ranks compute on cuda:0 and
gloo carries the collectives (on an 8-GPU node the same code runs one rank per GPU over RCCL: bench.py --gpus N).
The sharded result must equal the single-process one: plot range and shape exactly, the heat map to 1e-12 (the
summation order of the f64 atomics and of the all-reduce differs), the sweep's selection exactly."""

import os
import subprocess
import sys

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_equal_one_process(rt, tmp_path):
    port = str(29700 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dist_worker.py'), str(r), '2', port, str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = [p.communicate(timeout=110)[0] for p in procs]
    for p, log in zip(procs, logs):
        assert p.returncode == 0, log[-3000:]
    got = [np.load(os.path.join(str(tmp_path), f'rank{r}.npz')) for r in range(2)]

    g = common.load('c2_fan999')
    sim = rt.Simulation2D(common.product_env(rt, g), g['fan_source'])
    sim.cast(g['fan_angles'], **common.prop_kwargs(g))
    H = sim.heatmap_full(resolution=[50, 25])[0].cpu().numpy()
    plot = sim.heatmap(resolution=[50, 25], cutoff=.1)
    for mode in ('interleave', 'block'):
        assert int(got[0][f'{mode}_n_local']) + int(got[1][f'{mode}_n_local']) == len(g['fan_angles'])
        for r in range(2):
            Hr = got[r][f'{mode}_H']
            assert Hr.shape == H.shape
            assert np.abs(Hr - H).max() <= 1e-12 * H.max()
            assert np.abs(got[r][f'{mode}_plot'] - plot).max() < 1e-12
        assert np.array_equal(got[0][f'{mode}_H'], got[1][f'{mode}_H'])       # every rank holds the same map

    g = common.load('c5_eigen')
    assert int(got[0]['scan_n_local']) + int(got[1]['scan_n_local']) == int(g['n_rays_scan'])
    for r in range(2):
        assert np.array_equal(got[r]['scan_selected'], g['scan_selected_angles'])
        assert abs(float(got[r]['final_dist_avg']) - float(g['final_dist_avg'])) < 1e-6
    assert np.array_equal(got[0]['final_angles'], got[1]['final_angles'])
