// Where a fresh process's first call spends its time before any arithmetic (the `export.hip_init` phase of the export path, bench.py `export_path`):
// runtime start, device open, stream creation, first allocation, loading libzkmi's code object (first kernel launch of the library).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/hip_start_bench tools/hip_start_bench.hip -ldl && /tmp/hip_start_bench noir_backend_using_gnark_amd/libzkmi.so
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_nop(int* p) { if (p) *p = 1; }
int main(int argc, char** argv) {
    double t = now_ms(), t0 = t;
    auto lap = [&](const char* what) { double n = now_ms(); printf("  %-46s %9.3f ms\n", what, n - t); t = n; };
    void* lib = argc > 1 ? dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL) : nullptr;
    lap("dlopen(libzkmi.so) (fat binary registered)");
    int cnt = 0;
    (void)hipGetDeviceCount(&cnt);
    lap("hipGetDeviceCount (runtime start)");
    (void)hipSetDevice(0);
    lap("hipSetDevice(0)");
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    lap("hipGetDeviceProperties");
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    void* d = nullptr;
    (void)hipMalloc(&d, 1 << 20);
    lap("first hipMalloc (1 MB)");
    std::vector<hipStream_t> st(16);
    for (int i = 0; i < 16; i++) {
        const double a = now_ms();
        (void)hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, i & 1 ? hi : lo);
        printf("    stream %2d (%s priority) %8.3f ms\n", i, i & 1 ? "high" : "normal", now_ms() - a);
    }
    lap("16 x hipStreamCreateWithPriority");
    k_nop<<<1, 64, 0, st[0]>>>((int*)d);
    (void)hipStreamSynchronize(st[0]);
    lap("first kernel of THIS program (own code object)");
    for (int i = 1; i < 16; i++) { k_nop<<<1, 64, 0, st[i]>>>((int*)d); }
    (void)hipDeviceSynchronize();
    lap("first launch on each of the other 15 streams");
    if (lib) {
        typedef int (*ntt_fn)(void*, unsigned, int, int, int);
        ntt_fn ntt = (ntt_fn)dlsym(lib, "zk_bn254_ntt");
        std::vector<uint64_t> a(4 << 10, 1);
        if (ntt) {
            int rc = ntt(a.data(), 10, 0, 0, 0);
            lap("first zk_bn254_ntt 2^10 (library init + code object)");
            rc |= ntt(a.data(), 10, 0, 0, 0);
            lap("second zk_bn254_ntt 2^10");
            printf("  rc %d\n", rc);
        }
    }
    void* big = nullptr;
    (void)hipMalloc(&big, (size_t)4 << 30);
    lap("hipMalloc 4 GB");
    (void)hipMemsetAsync(big, 0, (size_t)4 << 30, st[0]);
    (void)hipStreamSynchronize(st[0]);
    lap("first touch of the 4 GB (memset)");
    void* pin = nullptr;
    (void)hipHostMalloc(&pin, (size_t)256 << 20, 0);
    lap("hipHostMalloc 256 MB pinned");
    printf("  total %.3f ms\n", now_ms() - t0);
    return 0;
}
