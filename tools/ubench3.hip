// Decides the batched-affine question (VERDICT r1 "next" #4) by measurement: what a bucket addition costs per wave on gfx950 when it is
// done in affine coordinates with a shared (Montgomery-trick) inversion, against the XYZZ mixed addition the accumulate kernel uses.
//   k_madd29        XYZZ += affine, the production routine (ff29.hpp xyzz_madd29): the baseline, cycles per addition per wave
//   k_affine_core   the arithmetic of ONE batched affine addition with the inverse already known: prefix product (1 M), the two products
//                   that peel the inverse off the batch (2 M), lambda = dy * inv (1 M), lambda^2 (1 S), y3 = lambda (x1 - x3) - y1 (1 M):
//                   5 M + 1 S in the 29-bit-limb form -- memory traffic, point re-reads and the inversion itself NOT included
//   k_inverse29     one Fermat inversion (254 squarings + ~127 products, the only constant-flow inversion available without a dedicated
//                   binary-GCD kernel): what every wave pays once per batch, whatever the number of lanes that need it
// => break-even batch size per lane:  inv / (madd - affine_core) additions per lane per inversion.
// build: hipcc -O3 --offload-arch=gfx950 -I../noir_backend_using_gnark_amd/csrc ubench3.hip -o ubench3 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "ff.hpp"
#include "curve.hpp"
#include "ff29.hpp"
using namespace zkmi;

__global__ __launch_bounds__(256) void k_madd29(G1XYZZ* out, const G1Affine* in, int iters) {
    size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Acc29 acc;
    acc.inf = true;
    G1Affine p = in[i + 1];
    xyzz_madd29(acc, in[i].x, in[i].y);
    for (int it = 0; it < iters; it++) { xyzz_madd29(acc, p.x, p.y); p.x.l[0] ^= acc.x.l[0] & 0xff; }
    out[i] = acc29_to_xyzz(acc);
}
__global__ __launch_bounds__(256) void k_affine_core(Fp* out, const Fp* in, int iters) {
    size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    U29 x1 = u29_load(in[i]), y1 = u29_load(in[i + 1]), x2 = u29_load(in[i + 2]), y2 = u29_load(in[i + 3]);
    U29 run = u29_load(in[i + 4]), inv = u29_load(in[i + 5]);
    for (int it = 0; it < iters; it++) {
        U29 dx = u29_wnorm(u29_sub<4>(x2, x1));
        U29 pre = run;
        run = u29_mul(run, dx);                     // forward sweep: prefix product
        U29 dinv = u29_mul(inv, pre);               // backward sweep: 1/dx ...
        inv = u29_mul(inv, dx);                     // ... and the inverse of the shorter prefix
        U29 lam = u29_mul(u29_wnorm(u29_sub<4>(y2, y1)), dinv);
        U29 x3 = u29_wnorm(u29_sub<8>(u29_sqr(lam), u29_add(x1, x2)));
        U29 y3 = u29_wnorm(u29_sub<4>(u29_mul(lam, u29_wnorm(u29_sub<4>(x1, x3))), y1));
        x1 = x3; y1 = y3;
        x2.l[0] ^= y3.l[0] & 0xff;
    }
    out[i] = u29_store(u29_mul(u29_mul(x1, y1), u29_mul(run, inv)));
}
__global__ __launch_bounds__(256) void k_inverse29(Fp* out, const Fp* in, int iters) {
    size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    U29 a = u29_load(in[i]);
    const uint32_t* e = FpParams::MOD;  // exponent p - 2 (bit 1 handled below: p ends in ...47, p - 2 in ...45)
    for (int it = 0; it < iters; it++) {
        U29 acc = u29_one(), base = a;
        for (int b = 0; b < 254; b++) {
            uint32_t w = e[b >> 5] - (b < 32 ? 2u : 0u);
            if ((w >> (b & 31)) & 1) acc = u29_mul(acc, base);
            base = u29_sqr(base);
        }
        a = acc;
    }
    out[i] = u29_store(a);
}

template <class K, class... A>
static double timeit(int blocks, int threads, K k, A... args) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, args...);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, args...);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { printf("no device\n"); return 1; }
    int cus = prop.multiProcessorCount;
    Fp *in, *out;
    size_t nmax = (size_t)cus * 8 * 256 + 64;
    hipMalloc(&in, nmax * sizeof(G1Affine));
    hipMalloc(&out, nmax * sizeof(G1XYZZ));
    std::vector<uint32_t> h(nmax * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) & ((i % 8 == 7) ? 0x0fffffffu : 0xffffffffu);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("{\"device\": \"%s\", \"cus\": %d, \"rows\": [\n", prop.name, cus);
    bool first = true;
    for (int wps : {1, 2, 4}) {
        int blocks = cus * wps;
        double c_madd, c_aff, c_inv;
        { int it = 200; double ms = timeit(blocks, 256, k_madd29, (G1XYZZ*)out, (const G1Affine*)in, it); c_madd = ms * 1e-3 * 2.4e9 / (it * 1.0 * wps); }
        { int it = 200; double ms = timeit(blocks, 256, k_affine_core, out, (const Fp*)in, it); c_aff = ms * 1e-3 * 2.4e9 / (it * 1.0 * wps); }
        { int it = 4; double ms = timeit(blocks, 256, k_inverse29, out, (const Fp*)in, it); c_inv = ms * 1e-3 * 2.4e9 / (it * 1.0 * wps); }
        printf("%s {\"waves_per_simd\": %d, \"cycles_per_wave\": {\"xyzz_madd29\": %.0f, \"affine_add_core_5M1S\": %.0f, \"fermat_inverse29\": %.0f}, "
               "\"break_even_additions_per_lane_per_inversion\": %.1f, \"additions_per_lane_for_1p25x\": %.1f}",
               first ? "" : ",\n", wps, c_madd, c_aff, c_inv, c_inv / (c_madd - c_aff), c_inv / (c_madd / 1.25 - c_aff));
        first = false;
    }
    printf("\n]}\n");
    return 0;
}
