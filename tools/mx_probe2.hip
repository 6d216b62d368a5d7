// mx_probe2.hip -- diagnostic: which lane's scale byte applies to which (row, k-block) of v_mfma_scale_f32_32x32x64_f8f6f4.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// A = B = 1.0 everywhere except: only k-block kb of A is non-zero when kb >= 0
__global__ void probe(float *D, int L, int which, int kb)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }
    if (kb >= 0 && h != kb) for (int i = 0; i < 8; ++i) a[i] = 0;
    int sa = 127, sb = 127;
    if (which == 0 && lane == L) sa = 126;
    if (which == 1 && lane == L) sb = 126;
    v16f c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}

int main()
{
    float *dD, hD[1024];
    hipMalloc(&dD, 4096);
    for (int which = 0; which < 2; ++which)
        for (int kb = -1; kb < 2; ++kb)
            for (int L : {0, 1, 5, 31, 32, 33, 37, 63}) {
                probe<<<1, 64>>>(dD, L, which, kb);
                hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
                const float base = kb < 0 ? 64.f : 32.f;
                int nchg = 0, rmin = 99, rmax = -1, cmin = 99, cmax = -1; float val = 0;
                for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c)
                    if (hD[r * 32 + c] != base) { ++nchg; val = hD[r * 32 + c]; if (r < rmin) rmin = r; if (r > rmax) rmax = r; if (c < cmin) cmin = c; if (c > cmax) cmax = c; }
                printf("scale_%c lane %2d x0.5, A k-block %2d live: %4d changed, rows %d..%d cols %d..%d, value %g (base %g)\n",
                       which ? 'b' : 'a', L, kb, nchg, rmin, rmax, cmin, cmax, val, base);
            }
    return 0;
}
