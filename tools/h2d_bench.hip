// How fast can a FRESH process move a 0.37 GB pageable text (the caller's key hex, csrc/keyio.hip zk_bn254_groth16_pk_read) into HBM?
// One variant per process (argv[1]), so every variant sees the runtime as a cold export call does:
//   plain      hipMalloc + one hipMemcpyAsync from the pageable buffer (what pk_read does), then the same copy again
//   two        the same, while a second thread does nine hipMalloc + hipMemcpy of 16 MB each (the circuit's upload beside it)
//   register   hipHostRegister the caller's buffer, copy, unregister
//   staged N   N host threads copy 8 MB pieces into a ring of four pinned buffers, one hipMemcpyAsync per piece
//   decode N   as staged, but the threads DECODE the hex while they copy: half the bytes cross PCIe and no device-side decoding is left
// Prints one JSON object.  Build: hipcc -O2 --offload-arch=gfx950 tools/h2d_bench.hip -o tools/h2d_bench -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static inline unsigned nib(unsigned char c) { return (c & 15) + 9 * (c >> 6); }

int main(int argc, char** argv) {
    const std::string mode = argc > 1 ? argv[1] : "plain";
    const int nthreads = argc > 2 ? atoi(argv[2]) : 8;
    const size_t bytes = (size_t)370 << 20;
    char* host = (char*)malloc(bytes);
    for (size_t i = 0; i < bytes; i++) host[i] = "0123456789abcdef"[(i * 2654435761u >> 7) & 15];  // touched, like a text that was just read
    const double t_start = now_ms();
    CK(hipSetDevice(0));
    CK(hipFree(nullptr));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const double t_up = now_ms();
    char* d = nullptr;
    CK(hipMalloc((void**)&d, bytes));
    const double t_alloc = now_ms();
    double t_copy1 = 0, t_copy2 = 0, t_extra = 0;
    if (mode == "plain" || mode == "two") {
        std::thread other;
        double other_ms = 0;
        if (mode == "two")
            other = std::thread([&] {
                const double a = now_ms();
                std::vector<char> src((size_t)16 << 20, 1);
                for (int k = 0; k < 9; k++) {
                    void* q = nullptr;
                    CK(hipMalloc(&q, (size_t)48 << 20));
                    CK(hipMemcpy(q, src.data(), src.size(), hipMemcpyHostToDevice));
                }
                other_ms = now_ms() - a;
            });
        double a = now_ms();
        CK(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        t_copy1 = now_ms() - a;
        if (other.joinable()) other.join();
        t_extra = other_ms;
        a = now_ms();
        CK(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        t_copy2 = now_ms() - a;
    } else if (mode == "register") {
        double a = now_ms();
        CK(hipHostRegister(host, bytes, hipHostRegisterDefault));
        t_extra = now_ms() - a;
        a = now_ms();
        CK(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        t_copy1 = now_ms() - a;
        a = now_ms();
        CK(hipHostUnregister(host));
        t_copy2 = now_ms() - a;  // (reported as copy2: the unregister)
    } else {  // staged / decode
        const bool dec = mode == "decode";
        const size_t piece = (size_t)8 << 20;  // bytes of TEXT per piece
        const int ring = 4;
        char* pin[ring];
        hipEvent_t done[ring];
        double a = now_ms();
        for (int i = 0; i < ring; i++) { CK(hipHostMalloc((void**)&pin[i], piece, hipHostMallocDefault)); CK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming)); }
        t_extra = now_ms() - a;  // the ring's pinned buffers
        a = now_ms();
        const size_t npieces = (bytes + piece - 1) / piece;
        // the pieces are filled in order by a team: every thread takes a slice of the current piece
        std::vector<std::thread> team;
        std::atomic<size_t> arrived{0};
        std::mutex mu;
        std::condition_variable cv;
        size_t phase = 0;
        auto barrier = [&](size_t target) {
            std::unique_lock<std::mutex> lk(mu);
            if (++arrived == (size_t)nthreads * target) { phase = target; cv.notify_all(); }
            else cv.wait(lk, [&] { return phase >= target; });
        };
        for (int t = 0; t < nthreads; t++)
            team.emplace_back([&, t] {
                for (size_t p = 0; p < npieces; p++) {
                    if (t == 0 && p >= (size_t)ring) CK(hipEventSynchronize(done[p % ring]));  // the buffer's previous copy has left
                    barrier(2 * p + 1);
                    const size_t from = p * piece, len = (from + piece <= bytes ? piece : bytes - from);
                    const size_t lo = len * t / nthreads & ~(size_t)1, hi = (t + 1 == nthreads) ? len : (len * (t + 1) / nthreads & ~(size_t)1);
                    if (dec) {
                        const unsigned char* s = (const unsigned char*)host + from;
                        unsigned char* o = (unsigned char*)pin[p % ring];
                        for (size_t i = lo; i < hi; i += 2) o[i >> 1] = (unsigned char)(nib(s[i]) << 4 | nib(s[i + 1]));
                    } else {
                        memcpy(pin[p % ring] + lo, host + from + lo, hi - lo);
                    }
                    barrier(2 * p + 2);
                    if (t == 0) {
                        if (dec) CK(hipMemcpyAsync(d + from / 2, pin[p % ring], len / 2, hipMemcpyHostToDevice, st));
                        else CK(hipMemcpyAsync(d + from, pin[p % ring], len, hipMemcpyHostToDevice, st));
                        CK(hipEventRecord(done[p % ring], st));
                    }
                }
            });
        for (auto& th : team) th.join();
        CK(hipStreamSynchronize(st));
        t_copy1 = now_ms() - a;
    }
    const double t_end = now_ms();
    printf("{\"mode\": \"%s\", \"threads\": %d, \"text_MB\": %zu, \"runtime_start_ms\": %.1f, \"hipMalloc_ms\": %.2f, \"copy_ms\": %.2f, \"GBps\": %.1f, \"second_ms\": %.2f, "
           "\"extra_ms\": %.2f, \"total_after_start_ms\": %.1f}\n",
           mode.c_str(), nthreads, bytes >> 20, t_up - t_start, t_alloc - t_up, t_copy1, bytes / t_copy1 / 1e6, t_copy2, t_extra, t_end - t_up);
    return 0;
}
