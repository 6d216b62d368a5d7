#!/usr/bin/env python
"""CPU tool (hipcc cross-compiles): audit the main loop of every gemm4p_kernel instantiation after an edit.
The kernel's weight loads are inline asm with hand-counted s_waitcnt vmcnt(N) (csrc/xb_encoder.hip): that is only sound
while the compiler (a) never copies a B register between its load and its wait (v_mov), (b) adds no vmcnt wait or branch of
its own inside the loop, (c) spills nothing.  A runtime-selected cache hint on those loads once produced a branchy loop that
faulted on the GPU; this check would have caught it on the CPU.  Exit status 1 on any violation."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "xna_basecaller_amd", "csrc")


def main():
    with tempfile.TemporaryDirectory() as d:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                               "-save-temps", "-c", os.path.join(SRC, "xb_encoder.hip"), "-I" + SRC, "-o", os.path.join(d, "x.o")],
                              cwd=d, stderr=subprocess.DEVNULL)
        text = open(os.path.join(d, "xb_encoder-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    bad = 0
    for m in re.finditer(r"^(_ZN12_GLOBAL__N_113gemm4p_kernelILi(\d)ELi(\d)EEEvN2xb10GemmParamsE):[^\n]*\n(.*?)s_endpgm", text, re.M | re.S):
        name, epi, nsplit, body = m.group(1), int(m.group(2)), int(m.group(3)), m.group(4)
        s16 = nsplit == 3 and "v_mfma_f32_16x16x32_f16" in body   # XB_GEMM_S16: the three-product arithmetic on 16x16x32 (two counted waits per k-tile:
        # B(t) landed in front of phase 0 and -- XB_GEMM_XTILE -- A(t + 1) landed in front of the barrier between phases 2 and 3, vmcnt(12) both)
        want_mfma = {1: 32, 2: 48, 3: 192 if s16 else 96}[nsplit]   # per two k-tiles (the loop is unrolled by two)
        want_loads = {1: 8, 2: 16, 3: 16}[nsplit]
        blocks = re.split(r"^\.LBB\d+_\d+:.*$", body, flags=re.M)
        loops = [b for b in blocks if b.count("v_mfma") == want_mfma]
        scratch = body.count("scratch_")
        ok = len(loops) == 1 and scratch == 0
        info = "no main loop found"
        if len(loops) == 1:
            b = loops[0]
            waits = [int(w) for w in re.findall(r"s_waitcnt vmcnt\((\d+)\)", b)]
            moves, loads, branches = b.count("v_mov"), b.count("global_load_dwordx4"), b.count("s_cbranch")
            ok = ok and moves == 0 and loads == want_loads and branches <= 2 and all(w >= 4 for w in waits) and len(waits) in ((4,) if s16 else (6, 8))
            info = "v_mov %d, loads %d, branches %d, vmcnt waits %s, scratch %d" % (moves, loads, branches, waits, scratch)
        print("gemm4p_kernel<%d, %d>: %s  %s" % (epi, nsplit, "ok " if ok else "BAD", info))
        bad += 0 if ok else 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
