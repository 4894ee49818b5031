import ctypes, sys
sys.path.insert(0, ".")
import force2vec_amd as F
L = F._lib.lib()
g = ctypes.c_double()
for mib in (1, 2, 4, 8, 16, 32, 64, 150, 1024):
    rc = L.f2v_diag_gather_rate(0, mib << 20, 3, ctypes.byref(g))
    print("table %5d MiB: rc %d, %.2f TB/s" % (mib, rc, g.value * 1e-3), flush=True)
