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

    dup = np.array(g['fan_angles'][:200], dtype=np.float64)
    dup[5] = dup[3]
    sim = rt.Simulation2D(common.product_env(rt, g), g['fan_source'])
    sim.cast(dup, sort=False, **common.prop_kwargs(g))
    Hd = sim.heatmap_full(resolution=[50, 25])[0].cpu().numpy()
    for r in range(2):
        assert got[r]['dup_H'].shape == Hd.shape and np.abs(got[r]['dup_H'] - Hd).max() <= 1e-12 * Hd.max()

    # (one rank's fused binner met rays it could not take: both ranks started over, the map is the single-process one)
    assert int(got[1]['viol_redone']) == 1 and int(got[0]['viol_redone']) == 0
    for r in range(2):
        assert got[r]['viol_H'].shape == H.shape and np.abs(got[r]['viol_H'] - H).max() <= 1e-12 * H.max()

    g = common.load('c5_eigen')
    assert int(got[0]['scan_n_local']) + int(got[1]['scan_n_local']) == int(g['n_rays_scan'])
    for r in range(2):
        assert np.array_equal(got[r]['scan_selected'], g['scan_selected_angles'])
        assert abs(float(got[r]['final_dist_avg']) - float(g['final_dist_avg'])) < 1e-6
    assert np.array_equal(got[0]['final_angles'], got[1]['final_angles'])


def test_bench_gpus_2_from_a_plain_invocation():
    """`python bench.py --gpus 2` with no launcher around it prints ONE JSON line with n_gpus 2 (two ranks on the one
    card here, so the collectives go over gloo and the line says so; with a GPU per rank they go over RCCL)."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                          '--large-rays', '0', '--no-cpu'], capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['ranks'] == 2 and rec['scaling'] == 'weak'
    assert rec['collective_backend'] in ('gloo', 'nccl (RCCL)')
    assert rec['config']['rays_per_gpu'] == 999 and rec['value'] > 0


def test_collectives_over_rccl_in_a_group_of_one():
    """The two all-reduces of the step through the nccl backend (= RCCL on ROCm) on this box's one GPU: a group of one
    rank, so the RCCL path has run before an 8-GPU node sees it.  Same line otherwise as the plain N = 1 run."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--force-collectives', '--steps', '3', '--warmup', '1',
                          '--large-rays', '0', '--no-cpu'], capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][0])
    assert rec['n_gpus'] == 1 and rec['ranks'] == 1 and rec['collective_backend'] == 'nccl (RCCL)'
    assert rec['config']['heatmap_shape'][0] > 40 and rec['value'] > 0
