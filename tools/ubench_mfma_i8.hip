// v_mfma_i32_32x32x32_i8 on gfx950: (1) the operand / result layout the SSD kernel (csrc/sm_cost_mfma.hip)
// relies on, checked against a host product; (2) what one instruction costs, back to back, at 1 and 2 waves
// per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_i8.hip -o tools/ubench_mfma_i8.bin
// Layout assumed (and verified here):
//   A (32 x 32, M x K): lane l holds row i = l % 32, K-slots (h = l / 32, t = 0..15) as 16 bytes;
//   B (32 x 32, K x N): lane l holds column j = l % 32, the SAME K-slots (h, t);
//   C (32 x 32 i32): lane l, register r holds C[8 (r / 4) + 4 (l / 32) + r % 4][l % 32].
// Which k a slot (h, t) is does not matter to a product as long as A and B agree -- the test feeds every
// slot its own value and compares with sum over slots.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(64) void k_layout(const signed char *A, const signed char *B, int *C)
{
    // A[i][h][t], B[j][h][t] as [32][2][16] bytes
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    v4i a = *reinterpret_cast<const v4i *>(A + (i * 2 + h) * 16);
    v4i b = *reinterpret_cast<const v4i *>(B + (i * 2 + h) * 16);
    v16i c = {};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; r++) C[(8 * (r / 4) + 4 * h + r % 4) * 32 + i] = c[r];
}

template <int W, int CH>
__global__ __launch_bounds__(64) void k_rate(int *out, unsigned long long *cyc, int iters)
{
    if (W == 1) asm volatile("" ::: "v250", "a16");
    if (W == 2) asm volatile("" ::: "v200");
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)blockIdx.x, 8};
    v16i c[CH];
    for (int k = 0; k < CH; k++) c[k] = v16i{};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int k = 0; k < CH; k++) c[k] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[k], 0, 0, 0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int k = 0; k < CH; k++) for (int r = 0; r < 16; r++) s += c[k][r];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    std::vector<signed char> A(1024), B(1024);
    srand(7);
    for (auto &v : A) v = (signed char)(rand() % 256 - 128);
    for (auto &v : B) v = (signed char)(rand() % 256 - 128);
    signed char *dA, *dB; int *dC;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
    k_layout<<<1, 64>>>(dA, dB, dC);
    std::vector<int> C(1024);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) {
            int s = 0;
            for (int k = 0; k < 32; k++) s += (int)A[i * 32 + k] * (int)B[j * 32 + k];
            if (s != C[i * 32 + j]) bad++;
        }
    printf("layout: %d of 1024 results differ from the host product (signed bytes, the layout in the header)\n", bad);

    int *out; unsigned long long *cyc;
    const int blocks = 256 * 4 * 2;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto kern, int nb, const char *name, int ch) {
        const int iters = 2000;
        kern<<<nb, 64>>>(out, cyc, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        kern<<<nb, 64>>>(out, cyc, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(nb);
        hipMemcpy(c.data(), cyc, nb * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : c) avg += v; avg /= nb;
        const double n = (double)iters * ch;
        // s_memtime counts at 100 MHz: wall time per wave
        printf("%-34s %7.2f ns per MFMA per wave (s_memtime), launch %8.1f us, %6.2f TOP/s chip-wide\n", name,
               avg * 10.0 / n, ms * 1e3, (double)nb * n * 65536.0 / (ms * 1e-3) / 1e12);
    };
    run(k_rate<1, 1>, 1024, "1 wave/SIMD, one dependent chain", 1);
    run(k_rate<1, 4>, 1024, "1 wave/SIMD, 4 accumulators", 4);
    run(k_rate<1, 9>, 1024, "1 wave/SIMD, 9 accumulators", 9);
    run(k_rate<2, 4>, 2048, "2 waves/SIMD, 4 accumulators", 4);
    run(k_rate<2, 9>, 2048, "2 waves/SIMD, 9 accumulators", 9);
    return bad != 0;
}
