import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
from godot_atmosphere_shader_amd import _native as N
lib = N.load()
ctx = C.c_void_p()
assert lib.atmo_create(0, N.VARIANT_NO_CLOUDS, 0, 0, N.LIGHT_LUT, 0, C.byref(ctx)) == N.ATMO_OK
bad = {}
for e in range(1, 255):
    bs, bd = C.c_uint32(0), C.c_uint32(0)
    assert lib.atmo_selftest_exact_math(ctx, e << 23, 1 << 23, 2.4, C.byref(bs), C.byref(bd)) == N.ATMO_OK
    if bs.value:
        bad[e] = bs.value
print("exponents with sqrt mismatches:", bad if bad else "none", "(biased exponents 1..254, all 2^23 significands each)")
lib.atmo_destroy(ctx)
