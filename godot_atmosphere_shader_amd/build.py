"""Builds libatmo_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU.  The .so lands next to this file so that it travels
with the source tree; it is git-ignored, never pip-installed.
"""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# ATMO_HIP_LIB: load a differently-built library instead (A/B experiments on compiler flags; see tools/ab_build.sh)
LIB_PATH = os.environ.get("ATMO_HIP_LIB") or os.path.join(_HERE, "libatmo_hip.so")
SOURCES = ["atmo_api.hip", "atmo_kernels.hip"]
HEADERS = ["atmo_device.h", "atmo_layout.h", os.path.join("..", "..", "include", "atmo.h"), os.path.join("..", "..", "include", "atmo_debug.h")]

# -ffp-contract=off: the kernels and the host-side per-frame constants must round exactly like a scalar
# fp32 evaluation of the shader wherever control flow or the ill-conditioned cloud chain is involved;
# hot loops opt back into FMA with `#pragma clang fp contract(fast)`.
# -fno-slp-vectorize: hipcc's SLP pass packs neighbouring scalar f32 ops into v_pk_*_f32 plus the v_mov
# shuffles that feed them; on gfx950 a v_pk_fma_f32 issues in ~5 cycles against 2.8 for v_fma_f32
# (profiles/round1/valu_peak_mi355x.jsonl), so the packing loses: measured +8..19 % Mrays/s on every workload
# with it off (profiles/round1/ab_slp_vectorize.txt).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
               "-shared", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the gfx950 extension cannot be built")


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def source_id() -> str:
    """What a library was built from: sha256 over the kernel / API sources, their headers and the compiler flags, 16 hex digits.  Compiled into the
    library (atmo_build_id) and written into every profiles/round<N>/pmc_*.json by tools/profile.sh, so that bench.py can tell when committed
    counters belong to other kernels than the ones it is timing."""
    import hashlib

    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    h.update(" ".join(HIPCC_FLAGS).encode())
    return h.hexdigest()[:16]


def build_native(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP extension if it is missing or stale; returns the library path."""
    if force or needs_build():
        cmd = [_hipcc()] + HIPCC_FLAGS + [f'-DATMO_BUILD_ID="{source_id()}"', "-o", LIB_PATH] + [os.path.join(CSRC, s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=CSRC)
    return LIB_PATH


def build_sanitized(verbose: bool = False) -> tuple[str, str]:
    """`make sanitize-host`: the same two sources with the HOST code under AddressSanitizer + UndefinedBehaviorSanitizer (the device code is
    compiled as always: GPU sanitizers are not available on this pool).  Returns (library, the ASan runtime to LD_PRELOAD into python)."""
    import glob

    lib = os.path.join(_HERE, "libatmo_hip_san.so")
    flags = [f for f in HIPCC_FLAGS if f != "-O3"] + ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                                                      "-fno-gpu-sanitize", "-shared-libsan"]
    cmd = [_hipcc()] + flags + ["-o", lib] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt:
        raise RuntimeError("the ASan runtime of ROCm's clang was not found")
    return lib, rt[-1]


if __name__ == "__main__":
    import sys

    if "--sanitize" in sys.argv:
        print(*build_sanitized(verbose=True))
    else:
        print(build_native(force=True, verbose=True))
