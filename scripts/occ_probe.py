import ctypes, sys
sys.path.insert(0, '.')
import torch; torch.cuda.init()
lib = ctypes.CDLL('libmultirobotplanning_amd/lib/libmrp_ll.so')
lib.mrp_ll_persistent_occupancy.restype = ctypes.c_int
for kind in (1, 2):
    print("kind", kind, [(b, lib.mrp_ll_persistent_occupancy(kind, b)) for b in (8192, 12288, 12800, 13056, 13168, 13312, 13424, 13680, 14336, 16384, 29000)])
