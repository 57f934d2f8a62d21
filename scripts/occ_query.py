"""Occupancy of the front / heavy kernels as the HIP runtime reports it (dev tool, GPU box)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libmultirobotplanning_amd import ll
lib = ll.load_library()
eng = ll.LowLevelEngine(device=0, n_tickets=1, slots=64)
lib.mrp_ll_front_heavy_occupancy.restype = ctypes.c_int
lib.mrp_ll_front_heavy_occupancy.argtypes = [ctypes.c_int, ctypes.c_uint32]
lib.mrp_ll_persistent_occupancy.restype = ctypes.c_int
lib.mrp_ll_persistent_occupancy.argtypes = [ctypes.c_int, ctypes.c_uint32]
print("heavy kernel (41600 B):", lib.mrp_ll_front_heavy_occupancy(1, 0))
for b in (8192, 10240, 12928, 13312, 13653, 14080, 16384, 20480, 27000, 32768, 40960, 41600, 54000, 65536, 81000):
    print("front kernel with %6d B dynamic LDS: %d per CU; all-tier kernel: %d" % (b, lib.mrp_ll_front_heavy_occupancy(0, b), lib.mrp_ll_persistent_occupancy(1, b)))
eng.close()
