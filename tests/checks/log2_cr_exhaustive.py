"""log2_cr (oracle/atmo_oracle.c; the kernels carry the same operation sequence) against the 80-bit logarithm rounded once, on every float of
[2^-4, 2^40): python tests/checks/log2_cr_exhaustive.py   (CPU, ~18 s; round 6: 0 mismatches of 369 098 752 arguments)"""
import os, sys, time
sys.path.insert(0, os.getcwd())
from oracle.oracle import Oracle

o = Oracle("f32")
t0, total, n = time.time(), 0, 0
for e in range(-4, 40):
    bad, first = o.log2_cr_check((127 + e) << 23, 1 << 23)
    total += bad
    n += 1 << 23
    if bad:
        print(f"binade 2^{e}: {bad} mismatches, first at bits {first:#x}")
print(f"log2_cr vs (float)log2l: {total} mismatches of {n} arguments, binades 2^-4 .. 2^39, {time.time() - t0:.1f} s")
