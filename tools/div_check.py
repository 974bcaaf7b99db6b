import sys, ctypes as C
sys.path.insert(0, '/root/repo')
import torch, rt2d_loader
rt = rt2d_loader.load()
L = rt._lib.lib()
L.rt2d_debug_div_check.argtypes = [C.c_int64, C.c_uint64, C.c_double, C.c_double, C.c_void_p, C.c_int32, C.c_void_p]
L.rt2d_debug_div_check.restype = C.c_int64
stream = torch.cuda.current_stream().cuda_stream
for mode in (0, 1):
  for seed, lo, hi in ((1, -30., 30.), (2, -3., 3.), (3, -1., 1.), (4, -100., 100.), (5, 0., 10.), (6, -10., 0.)):
      bad = torch.zeros(1, dtype=torch.int64, device='cuda')
      L.rt2d_debug_div_check(1 << 26, seed, lo, hi, bad.data_ptr(), mode, stream)
      torch.cuda.synchronize()
      print(mode, seed, lo, hi, int(bad.item()), 'of', 1 << 26)
