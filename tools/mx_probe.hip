// mx_probe.hip -- diagnostic: pins down the operand/scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 and of the
// fp8 conversion on gfx950 with exact small-integer data.  Not part of the product.
//   hipcc --offload-arch=gfx950 -O2 -o mx_probe tools/mx_probe.hip && ./mx_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void probe(const unsigned char *A, const unsigned char *B, float *D, int sa_lo, int sa_hi, int sb_lo, int sb_hi)
{
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    v8i a, b;
    memcpy(&a, A + r * 64 + 32 * h, 32);     // lane (r,h): bytes j = 0..31 <- A[r][32h + j]
    memcpy(&b, B + r * 64 + 32 * h, 32);     //             B[c = r][32h + j]
    v16f c = {};
    const int sa = h ? sa_hi : sa_lo, sb = h ? sb_hi : sb_lo;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}

__global__ void cvt(const float *x, unsigned *out, int n)
{
    const int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], 0.0f, 0, false) & 0xffff;
}

static unsigned char enc(int v)   // small integers in OCP e4m3fn
{
    static const unsigned char t[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50};
    return v < 0 ? (0x80 | t[-v]) : t[v];
}

int main()
{
    unsigned char hA[32 * 64], hB[32 * 64];
    int iA[32 * 64], iB[32 * 64];
    srand(5);
    for (int i = 0; i < 32 * 64; ++i) {
        iA[i] = rand() % 9 - 4; iB[i] = rand() % 9 - 4;
        hA[i] = enc(iA[i]); hB[i] = enc(iB[i]);
    }
    unsigned char *dA, *dB; float *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, 4096);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    float hD[1024];
    struct { int al, ah, bl, bh; const char *what; } cases[] = {
        {127, 127, 127, 127, "all scales 1.0"},
        {126, 127, 127, 127, "A lanes<32 x0.5"},
        {127, 127, 127, 125, "B lanes>=32 x0.25"},
        {127 | (120 << 8), 127, 127, 127, "A byte1=120 (opsel 0 must ignore)"},
    };
    for (auto &cs : cases) {
        probe<<<1, 64>>>(dA, dB, dD, cs.al, cs.ah, cs.bl, cs.bh);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
        const double fa0 = ldexp(1.0, (cs.al & 255) - 127), fa1 = ldexp(1.0, (cs.ah & 255) - 127);
        const double fb0 = ldexp(1.0, (cs.bl & 255) - 127), fb1 = ldexp(1.0, (cs.bh & 255) - 127);
        int bad = 0; double maxd = 0;
        for (int r = 0; r < 32; ++r)
            for (int c = 0; c < 32; ++c) {
                double s0 = 0, s1 = 0;
                for (int k = 0; k < 32; ++k) { s0 += iA[r * 64 + k] * iB[c * 64 + k]; s1 += iA[r * 64 + 32 + k] * iB[c * 64 + 32 + k]; }
                const double ref = s0 * fa0 * fb0 + s1 * fa1 * fb1;
                const double d = fabs(ref - hD[r * 32 + c]);
                if (d > maxd) maxd = d;
                if (d > 1e-6) ++bad;
            }
        printf("%-40s mismatches %d / 1024  max |diff| %.4g   D[0][0]=%g D[1][2]=%g\n", cs.what, bad, maxd, hD[0], hD[34]);
    }
    // conversions
    float hx[16] = {0.f, 1.f, -1.f, 0.5f, 448.f, 449.f, 480.f, 1000.f, -1000.f, 0.015625f, 0.001953125f, 0.0009765625f, 3.3f, 1e-8f, INFINITY, 240.f};
    float *dx; unsigned *dout, hout[16];
    hipMalloc(&dx, sizeof hx); hipMalloc(&dout, sizeof hout);
    hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice);
    cvt<<<1, 64>>>(dx, dout, 16);
    hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) printf("cvt_pk_fp8_f32(%g) = 0x%02x\n", hx[i], hout[i] & 0xff);
    return 0;
}
