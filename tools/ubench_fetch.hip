// What FETCH_SIZE counts for 4-byte-per-lane loads on gfx950 (the edge kernel's access width):
// the guide's "FETCH_SIZE reports half the bytes" was calibrated on 16-byte-per-lane streams.
// Reads a known number of bytes once, with dword / dwordx2 / dwordx4 loads per lane; run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./tools/ubench_fetch.bin
// and compare FETCH_SIZE (KiB) with the bytes printed here.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch.hip -o tools/ubench_fetch.bin
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T>
__global__ void k_read(const T *__restrict__ src, unsigned *__restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned acc = 0;
    for (; i < n; i += stride) {
        const T v = src[i];
        const unsigned *w = reinterpret_cast<const unsigned *>(&v);
        for (unsigned k = 0; k < sizeof(T) / 4; k++) acc ^= w[k];
    }
    if (acc == 0x12345678u) out[0] = acc;      // keeps the loads alive, practically never stores
}

int main()
{
    const size_t bytes = 512ull << 20;          // 512 MiB: twice the Infinity Cache, read once
    void *buf; unsigned *out;
    (void)hipMalloc(&buf, bytes); (void)hipMalloc(&out, 4);
    (void)hipMemset(buf, 1, bytes);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k_read<unsigned>), dim3(4096), dim3(256), 0, 0, (const unsigned *)buf, out, bytes / 4);
    hipLaunchKernelGGL((k_read<uint2>), dim3(4096), dim3(256), 0, 0, (const uint2 *)buf, out, bytes / 8);
    hipLaunchKernelGGL((k_read<uint4>), dim3(4096), dim3(256), 0, 0, (const uint4 *)buf, out, bytes / 16);
    (void)hipDeviceSynchronize();
    printf("each kernel read %zu bytes = %zu KiB once (k_read<unsigned int>: 4 B per lane, <uint2>: 8, <uint4>: 16)\n", bytes, bytes >> 10);
    return 0;
}
