// Measurement tool: what a host-slice transform (zk_bn254_ntt on a caller's pageable []fr.Element) can get out of PCIe.  32 MB up + 32 MB down per 2^20-point
// transform is 1.22 of its 1.36 ms; three transforms issued from three threads overlap only partly.  This prints, for 32 MB buffers:
//   pageable H2D / D2H alone, back to back, and from 3 concurrent threads (each up then down on its own stream);
//   the same after hipHostRegister of the caller's buffer (cost of register / unregister included and separately);
//   through pinned staging buffers with a host memcpy.
//     hipcc -O2 --offload-arch=gfx950 tools/pcie_bench.hip -o /tmp/pcie_bench -lpthread && /tmp/pcie_bench
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
    const size_t B = 32u << 20;
    const int T = 3, REP = 10;
    std::vector<char*> host(T);
    std::vector<void*> dev(T), pin(T);
    std::vector<hipStream_t> st(T);
    for (int k = 0; k < T; k++) {
        host[k] = (char*)malloc(B);
        memset(host[k], k + 1, B);
        CK(hipMalloc(&dev[k], B));
        CK(hipHostMalloc(&pin[k], B, hipHostMallocDefault));
        CK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
    }
    auto updown = [&](int k, bool registered) {
        if (registered) CK(hipHostRegister(host[k], B, hipHostRegisterDefault));
        CK(hipMemcpyAsync(dev[k], host[k], B, hipMemcpyHostToDevice, st[k]));
        CK(hipMemcpyAsync(host[k], dev[k], B, hipMemcpyDeviceToHost, st[k]));
        CK(hipStreamSynchronize(st[k]));
        if (registered) CK(hipHostUnregister(host[k]));
    };
    auto staged = [&](int k) {
        memcpy(pin[k], host[k], B);
        CK(hipMemcpyAsync(dev[k], pin[k], B, hipMemcpyHostToDevice, st[k]));
        CK(hipMemcpyAsync(pin[k], dev[k], B, hipMemcpyDeviceToHost, st[k]));
        CK(hipStreamSynchronize(st[k]));
        memcpy(host[k], pin[k], B);
    };
    for (int k = 0; k < T; k++) { updown(k, false); updown(k, true); staged(k); }  // warm
    printf("{");
    {
        double t0 = now_ms();
        for (int r = 0; r < REP; r++) { CK(hipMemcpy(dev[0], host[0], B, hipMemcpyHostToDevice)); }
        double up = (now_ms() - t0) / REP;
        t0 = now_ms();
        for (int r = 0; r < REP; r++) { CK(hipMemcpy(host[0], dev[0], B, hipMemcpyDeviceToHost)); }
        double down = (now_ms() - t0) / REP;
        printf("\"pageable_h2d_ms\": %.3f, \"pageable_d2h_ms\": %.3f, ", up, down);
    }
    for (int mode = 0; mode < 3; mode++) {
        const char* name = mode == 0 ? "pageable" : mode == 1 ? "registered_per_call" : "pinned_staging_memcpy";
        double t0 = now_ms();
        for (int r = 0; r < REP; r++)
            for (int k = 0; k < T; k++) { if (mode == 2) staged(k); else updown(k, mode == 1); }
        double seq = (now_ms() - t0) / REP;
        t0 = now_ms();
        for (int r = 0; r < REP; r++) {
            std::vector<std::thread> th;
            for (int k = 0; k < T; k++) th.emplace_back([&, k] { if (mode == 2) staged(k); else updown(k, mode == 1); });
            for (auto& t : th) t.join();
        }
        double conc = (now_ms() - t0) / REP;
        printf("\"%s_3x_updown_sequential_ms\": %.3f, \"%s_3x_updown_three_threads_ms\": %.3f, ", name, seq, name, conc);
    }
    {
        double t0 = now_ms();
        for (int r = 0; r < REP; r++) { CK(hipHostRegister(host[0], B, hipHostRegisterDefault)); CK(hipHostUnregister(host[0])); }
        printf("\"register_unregister_32MB_ms\": %.3f, ", (now_ms() - t0) / REP);
        CK(hipHostRegister(host[0], B, hipHostRegisterDefault));
        CK(hipHostRegister(host[1], B, hipHostRegisterDefault));
        t0 = now_ms();
        for (int r = 0; r < REP; r++) {  // full duplex on registered memory: buffer 0 goes up while buffer 1 comes down
            CK(hipMemcpyAsync(dev[0], host[0], B, hipMemcpyHostToDevice, st[0]));
            CK(hipMemcpyAsync(host[1], dev[1], B, hipMemcpyDeviceToHost, st[1]));
            CK(hipStreamSynchronize(st[0]));
            CK(hipStreamSynchronize(st[1]));
        }
        printf("\"registered_duplex_32MB_each_way_ms\": %.3f, ", (now_ms() - t0) / REP);
        t0 = now_ms();
        for (int r = 0; r < REP; r++) { CK(hipMemcpyAsync(dev[0], host[0], B, hipMemcpyHostToDevice, st[0])); CK(hipStreamSynchronize(st[0])); }
        printf("\"registered_h2d_ms\": %.3f}\n", (now_ms() - t0) / REP);
    }
    return 0;
}
