"""GPU: rt2d_heatmap_accumulate_noalloc (the entry point for callers that cannot allocate: its work list is a device array
built into the library) against rt2d_heatmap_accumulate with a caller-provided workspace, through the C ABI.  Two calls in
a row, the second on a smaller fan, so that whatever the first call left in the built-in list must not matter.
Bar: the heat map's, 1e-12 relative per cell (f64 atomics arrive in a different order from run to run); same empty cells."""

import ctypes as C

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


def _maps(rt, fan, shape, res=(50, 25)):
    import torch
    L = rt._lib.lib()
    stream = torch.cuda.current_stream().cuda_stream
    fan.populate(1)
    mask = fan.heatmap_mask(res, shape)
    res_c, shape_c = (C.c_int32 * 2)(*res), (C.c_int32 * 2)(*shape)
    H_ref = torch.zeros(shape[0] * shape[1], dtype=torch.float64, device=fan.device)
    H_new = torch.zeros_like(H_ref)
    drop_ref = torch.zeros(1, dtype=torch.int64, device=fan.device)
    drop_new = torch.zeros_like(drop_ref)
    work = torch.empty(fan.n + 2, dtype=torch.int32, device=fan.device)
    args = (fan.X.data_ptr(), fan.Z.data_ptr(), fan.G[1].data_ptr(), fan.tile_off.data_ptr(), fan.n_pts.data_ptr(), fan.n,
            mask.data_ptr(), res_c, shape_c)
    rt._lib.check(L.rt2d_heatmap_accumulate(*args, H_ref.data_ptr(), drop_ref.data_ptr(), work.data_ptr(), stream))
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    rt._lib.check(L.rt2d_heatmap_accumulate_noalloc(*args, H_new.data_ptr(), drop_new.data_ptr(), stream))
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before, 'the allocation-free entry point changed the free device memory'
    return H_ref.cpu().numpy(), H_new.cpu().numpy(), int(drop_ref.item()), int(drop_new.item()), int(work[0].item())


def test_noalloc_entry_point_equals_the_workspace_one(rt):
    g = common.load('c2_fan999')
    env = common.product_env(rt, g)
    kw = common.prop_kwargs(g)
    shape = (1503, 62)
    # first call: 20 000 rays (tile kernel + the queue of turned-back rays for the general kernel); second call: the
    # 999-ray fan, smaller than what the first call left in the built-in list
    big = rt.RayFan(env, g['fan_source'], -(np.arange(20000) + 1) * (0.999 / 20000), **kw)
    small = rt.RayFan(env, g['fan_source'], g['fan_angles'], **kw)
    for fan in (big, small):
        H_ref, H_new, d_ref, d_new, queued = _maps(rt, fan, shape)
        assert queued > 0, 'the tile kernel should have queued the turned-back rays'
        assert d_ref == d_new
        assert np.array_equal(H_ref == 0, H_new == 0)
        nz = H_ref != 0
        assert (np.abs(H_new - H_ref)[nz] / np.abs(H_ref[nz])).max() < 1e-12
        assert H_ref.sum() > fan.n


def test_noalloc_entry_point_rejects_fans_beyond_its_list(rt):
    L = rt._lib.lib()
    one = (C.c_int32 * 2)(50, 25)
    rc = L.rt2d_heatmap_accumulate_noalloc(1, 1, 1, 1, 1, (1 << 20) + 1, None, one, (C.c_int32 * 2)(10, 10), 1, None, None)
    assert rc < 0 and b'exceed' in L.rt2d_last_error()
