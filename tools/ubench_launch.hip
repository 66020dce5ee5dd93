// Workgroup launch ramp on gfx950: how long does it take to get G one-wave (or
// four-wave) workgroups resident, as a function of their register / LDS footprint,
// back to back and behind a different kernel?  Each wave spins for a fixed time
// (s_memtime), so  kernel time - spin time = launch ramp + tail.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_launch.hip -o tools/ubench_launch.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int VG>
__global__ __launch_bounds__(256) void k_spin(unsigned *out, long long spin_cycles)
{
    extern __shared__ unsigned lds[];
    // claim VG vector registers
    unsigned acc = threadIdx.x;
    if (VG >= 200) { asm volatile("v_mov_b32 v200, %0" :: "v"(acc) : "v200"); }
    if (VG >= 120) { asm volatile("v_mov_b32 v120, %0" :: "v"(acc) : "v120"); }
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin_cycles) { __builtin_amdgcn_s_sleep(8); }
    if (spin_cycles < 0) { lds[threadIdx.x] = acc; out[blockIdx.x] = lds[0]; }
}

__global__ void k_fill(unsigned *p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1;
}

template <int VG>
static double run(int G, int threads, int lds, long long spin, bool behind_fill, unsigned *junk, unsigned *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ts;
    for (int rep = 0; rep < 12; rep++) {
        if (behind_fill) hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, junk, (size_t)(16 << 20));
        else hipLaunchKernelGGL((k_spin<VG>), dim3(G), dim3(threads), lds, 0, out, spin);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_spin<VG>), dim3(G), dim3(threads), lds, 0, out, spin);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) ts.push_back(ms * 1e3f);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2];
}

int main()
{
    unsigned *junk, *out;
    hipMalloc(&junk, 64 << 20); hipMalloc(&out, 1 << 20);
    // s_memtime / readcyclecounter ticks at 100 MHz on gfx9: 2000 ticks = 20 us
    const long long spin = 2000;
    printf("%-44s %10s %10s\n", "grid", "behind same", "behind fill");
    struct Cfg { const char *name; int G, threads, lds, vg; } cfgs[] = {
        {"2048 x 64 thr, 211-ish VGPR, 2.5 KB LDS", 2048, 64, 2560, 200},
        {"2048 x 64 thr, 211-ish VGPR, 20 KB LDS", 2048, 64, 20480, 200},
        {"2048 x 64 thr, 120-ish VGPR, 2.5 KB LDS", 2048, 64, 2560, 120},
        {"2048 x 64 thr, few VGPR, 0 LDS", 2048, 64, 0, 0},
        {" 512 x 256 thr, 211-ish VGPR, 10 KB LDS", 512, 256, 10240, 200},
        {" 512 x 256 thr, few VGPR, 0 LDS", 512, 256, 0, 0},
        {"1024 x 64 thr, 211-ish VGPR, 2.5 KB LDS", 1024, 64, 2560, 200},
        {"4096 x 64 thr, few VGPR, 0 LDS", 4096, 64, 0, 0},
    };
    for (auto &c : cfgs) {
        double a, b;
        if (c.vg >= 200) { a = run<200>(c.G, c.threads, c.lds, spin, false, junk, out); b = run<200>(c.G, c.threads, c.lds, spin, true, junk, out); }
        else if (c.vg >= 120) { a = run<120>(c.G, c.threads, c.lds, spin, false, junk, out); b = run<120>(c.G, c.threads, c.lds, spin, true, junk, out); }
        else { a = run<0>(c.G, c.threads, c.lds, spin, false, junk, out); b = run<0>(c.G, c.threads, c.lds, spin, true, junk, out); }
        printf("%-44s %8.1f us %8.1f us   (spin 20 us)\n", c.name, a, b);
    }
    return 0;
}
