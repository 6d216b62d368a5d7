// GPU probe: can a stream wait (hipStreamWaitValue32) on a word that a RUNNING kernel on another stream writes, and how long
// after the write does the waiting stream's next kernel start?  (Question behind it: one recurrence launch per layer that
// signals its time slabs to the GEMM stream, instead of one launch per slab.)
//   hipcc --offload-arch=gfx950 -O2 tools/waitvalue_probe.hip -o tools/waitvalue_probe && tools/waitvalue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <thread>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void producer(unsigned *flag, unsigned long long *tw, int n, unsigned long long gap_ticks)
{
    for (int i = 1; i <= n; ++i) {
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < gap_ticks) __builtin_amdgcn_s_sleep(8);
        tw[i] = wall_clock64();
        __hip_atomic_store(flag, (unsigned)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ void consumer(unsigned long long *tr, int i) { tr[i] = wall_clock64(); }

int main()
{
    int can = -1;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    const int n = 16;
    for (int kind = 0; kind < 3; ++kind) {
        unsigned *flag = nullptr;
        hipError_t e;
        const char *name = kind == 0 ? "hipMallocSignalMemory" : kind == 1 ? "hipMalloc" : "hipHostMalloc(coherent)";
        if (kind == 0) e = hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory);
        else if (kind == 1) e = hipMalloc((void **)&flag, 8);
        else e = hipHostMalloc((void **)&flag, 8, hipHostMallocCoherent);
        if (e != hipSuccess) { printf("%s: allocation failed: %s\n", name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        hipStream_t a, b, c;
        CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
        CK(hipMemsetAsync(flag, 0, 8, a));
        CK(hipStreamSynchronize(a));
        unsigned long long *tw, *tr;
        CK(hipHostMalloc((void **)&tw, 8 * (n + 2), 0));
        CK(hipHostMalloc((void **)&tr, 8 * (n + 2), 0));
        for (int i = 0; i < n + 2; ++i) tw[i] = tr[i] = 0;
        bool ok = true;
        for (int i = 1; i <= n && ok; ++i) {
            e = hipStreamWaitValue32(b, flag, (unsigned)i, hipStreamWaitValueGte, 0xffffffffu);
            if (e != hipSuccess) { printf("%s: hipStreamWaitValue32 -> %s\n", name, hipGetErrorString(e)); (void)hipGetLastError(); ok = false; break; }
            hipLaunchKernelGGL(consumer, dim3(1), dim3(64), 0, b, tr, i);
        }
        if (ok) {
            hipLaunchKernelGGL(producer, dim3(1), dim3(64), 0, a, flag, tw, n, 20000ull /* 200 us at 100 MHz */);
            // watchdog: release the waiting stream if the flag never satisfies it
            bool done = false;
            for (int ms = 0; ms < 3000; ++ms) {
                if (hipStreamQuery(b) == hipSuccess) { done = true; break; }
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            if (!done) {
                printf("%s: waits not satisfied by the kernel's stores; releasing with hipStreamWriteValue32\n", name);
                CK(hipStreamSynchronize(a));
                CK(hipStreamWriteValue32(c, flag, 1000u, 0));
                CK(hipStreamSynchronize(c));
            }
            CK(hipStreamSynchronize(a));
            CK(hipStreamSynchronize(b));
            if (done) {
                printf("%s: wake-up latency (flag store -> next kernel on the waiting stream), us:", name);
                for (int i = 1; i <= n; ++i) printf(" %.1f", (double)((long long)tr[i] - (long long)tw[i]) / 100.0);
                printf("\n");
            }
        }
        (void)hipStreamDestroy(a); (void)hipStreamDestroy(b); (void)hipStreamDestroy(c);
        (void)hipHostFree(tw); (void)hipHostFree(tr);
        if (kind == 2) (void)hipHostFree(flag); else (void)hipFree(flag);
    }
    return 0;
}
