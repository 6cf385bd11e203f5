// Host <-> device copy rates for a caller-owned (pageable) buffer: plain hipMemcpy, against
// hipHostRegister + hipMemcpy + hipHostUnregister.  Decides how the stateless entry points move
// the p x d matrices.   build: hipcc -O2 scripts/perf_pcie.cpp -o scripts/bin/perf_pcie
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char** argv)
{
    size_t gb = argc > 1 ? atoi(argv[1]) : 4;
    size_t bytes = gb << 30;
    char* h = (char*)malloc(bytes);
    memset(h, 1, bytes);
    void* d;
    CK(hipMalloc(&d, bytes));
    CK(hipMemset(d, 0, bytes));
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
        double t1 = now();
        CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
        double t2 = now();
        printf("pageable: H2D %.3f s (%.1f GB/s), D2H %.3f s (%.1f GB/s)\n", t1 - t0, bytes / (t1 - t0) / 1e9, t2 - t1, bytes / (t2 - t1) / 1e9);
        t0 = now();
        hipError_t er = hipHostRegister(h, bytes, hipHostRegisterDefault);
        if (er != hipSuccess) {   // refused on the round-1 box ("invalid argument")
            printf("hipHostRegister: %s -- skipped\n", hipGetErrorString(er));
            (void)hipGetLastError();
            continue;
        }
        t1 = now();
        CK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
        t2 = now();
        CK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
        double t3 = now();
        CK(hipHostUnregister(h));
        double t4 = now();
        printf("registered: register %.3f s, H2D %.3f s (%.1f GB/s), D2H %.3f s (%.1f GB/s), unregister %.3f s\n",
               t1 - t0, t2 - t1, bytes / (t2 - t1) / 1e9, t3 - t2, bytes / (t3 - t2) / 1e9, t4 - t3);
        fflush(stdout);
    }
    return 0;
}
