// Does s_setprio decide which of the two waves of a SIMD is served first on THIS box?
// Two waves on every SIMD (a register claim caps the occupancy) run the same stream -- 15 v_bitop3 : 1
// v_alignbit, the match kernel's mix -- for about 40 us.  The wave in the odd hardware slot sets the
// priority given on the command line (default 3), the wave in the even slot 0; every wave stamps its
// end with s_memtime.  Where the priority is honoured the odd-slot waves finish first by a wide margin;
// where the arbiter serves the older wave regardless, the even-slot (older) waves do, as with no priority.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_prio.hip -o tools/ubench_prio.bin ; tools/ubench_prio.bin [prio_odd [prio_even]]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
typedef unsigned u32;

__global__ __launch_bounds__(64) void k_race(u32 *out, u64 *info, int iters, int prio_odd, int prio_even)
{
    asm volatile("" ::: "v200");                      // 2 waves per SIMD
    const u32 hw = __builtin_amdgcn_s_getreg(63492);
    const int odd = hw & 1;
    if (odd ? prio_odd == 3 : prio_even == 3) __builtin_amdgcn_s_setprio(3);
    else if (odd ? prio_odd == 2 : prio_even == 2) __builtin_amdgcn_s_setprio(2);
    else if (odd ? prio_odd == 1 : prio_even == 1) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
    u32 a[8];
    const u32 x = threadIdx.x * 2654435761u, y = blockIdx.x + 12345u;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = x + i;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if ((r * 8 + i) % 16 == 15) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(x));
                else asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(a[i]) : "v"(x), "v"(y));
            }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    u32 s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s ^= a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) { info[2 * blockIdx.x] = t1 - t0; info[2 * blockIdx.x + 1] = hw; }
}

int main(int argc, char **argv)
{
    const int prio_odd = argc > 1 ? atoi(argv[1]) : 3, prio_even = argc > 2 ? atoi(argv[2]) : 0;
    const int grid = 256 * 4 * 2, iters = 140;
    u32 *out; u64 *info;
    (void)hipMalloc(&out, grid * 64 * sizeof(u32));
    (void)hipMalloc(&info, grid * 2 * sizeof(u64));
    std::vector<u64> h(grid * 2);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_race, dim3(grid), dim3(64), 0, 0, out, info, iters, prio_odd, prio_even);
        (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(h.data(), info, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc[2];
    for (int b = 0; b < grid; b++) cyc[h[2 * b + 1] & 1].push_back((double)h[2 * b]);
    double med[2];
    for (int k = 0; k < 2; k++) {
        std::sort(cyc[k].begin(), cyc[k].end());
        med[k] = cyc[k].empty() ? 0 : cyc[k][cyc[k].size() / 2];
    }
    const double n_instr = (double)iters * 128;
    printf("priority odd slot %d / even slot %d: even-slot waves %zu, %.2f cycles per instruction (median); odd-slot waves %zu, %.2f\n",
           prio_odd, prio_even, cyc[0].size(), med[0] / n_instr, cyc[1].size(), med[1] / n_instr);
    if (prio_odd != prio_even) {
        const int fav = prio_odd > prio_even;
        printf("  -> the favoured (%s-slot) waves run at %.2f of the others' cycles per instruction: s_setprio %s here\n",
               fav ? "odd" : "even", med[fav] / med[1 - fav], med[fav] < 0.97 * med[1 - fav] ? "decides the race" : "does NOT decide the race");
    }
    return 0;
}
