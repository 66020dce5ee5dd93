// Are byte-unaligned ds_read_b32 usable for bit-row views?  (1) do they return the right
// bytes, (2) what do they cost when the 64 lanes of a wave read 64 different byte addresses
// inside a dozen dwords (the pattern the match kernel's right-row views would have)?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_lds_unaligned.hip -o tools/ubench_lds_unaligned.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned *out, unsigned long long *cyc, int iters)
{
    __shared__ unsigned char lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    const int lane = threadIdx.x, wi = lane >> 3, s = lane & 7;
    unsigned addr;
    if (MODE == 0) addr = 4 * wi;                       // aligned, 8 distinct dwords (left-row views)
    else if (MODE == 1) addr = 4 * wi + 2 * s;          // 2-byte steps: half of them unaligned
    else if (MODE == 2) addr = 4 * wi + 2 * s + 1;      // all odd addresses
    else if (MODE == 3) addr = 4 * lane;                // aligned, 64 distinct dwords, conflict-free
    else addr = 4 * lane + 1;                           // unaligned, 64 distinct
    unsigned acc = 0;
    const unsigned base = (unsigned)(size_t)lds + addr;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        unsigned v0, v1, v2, v3;
        asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:64\n ds_read_b32 %2, %4 offset:128\n"
                     "ds_read_b32 %3, %4 offset:192\n s_waitcnt lgkmcnt(0)"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(base));
        acc += v0 ^ v1 ^ v2 ^ v3;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    // correctness of one read
    unsigned v;
    asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base));
    unsigned want = 0;
    for (int b = 0; b < 4; b++) want |= (unsigned)lds[addr + b] << (8 * b);
    out[blockIdx.x * 64 + lane] = (v == want ? 0u : 1u) + (acc & 0);
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> static void run(const char *name, unsigned *out, unsigned long long *cyc)
{
    const int iters = 2000, grid = 2048;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned> h(grid * 64); std::vector<unsigned long long> c(grid);
    (void)hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    long bad = 0; double cy = 0;
    for (auto v : h) bad += v;
    for (auto v : c) cy += (double)v;
    printf("%-44s wrong values %ld, %.1f cycles per ds_read_b32 per wave (8 waves per CU)\n", name, bad, cy / grid / (iters * 4.0));
}

int main()
{
    unsigned *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 2048 * 64 * 4); (void)hipMalloc(&cyc, 2048 * 8);
    run<0>("aligned, 8 dwords x 8 lanes (broadcast)", out, cyc);
    run<1>("4*wi + 2*s: half unaligned", out, cyc);
    run<2>("4*wi + 2*s + 1: all odd", out, cyc);
    run<3>("aligned, 64 distinct dwords", out, cyc);
    run<4>("unaligned (+1), 64 distinct", out, cyc);
    return 0;
}
