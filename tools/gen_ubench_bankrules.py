#!/usr/bin/env python3
"""Generate tools/ubench_bankrules.hip: which VGPR bank patterns of a VALU instruction's
sources cost issue cycles on gfx950 when TWO waves share a SIMD (one wave alone hides
them: it issues one instruction per 4 cycles anyway).  Straight-line bodies of 1024
instructions, 8 independent chains in v8..v15 (dst = src0), sources from v16..v199.
Bank of a register = its number mod 4."""
from pathlib import Path

N = 1024

def reg_in_bank(bank, k):
    """k-th register of the pool v16..v199 that lies in `bank`"""
    return 16 + ((bank - 16) % 4) + 4 * (k % 46)

def body(kind):
    kind = kind.replace(" +skew", "").replace(" +prio", "").replace(" +far", "")
    out = []
    for i in range(N):
        d = 8 + (i % 8)
        b0 = d % 4
        k = i // 8
        if kind == "b3 distinct" or kind.startswith("skew "):
            s1, s2 = reg_in_bank(b0 + 1, k), reg_in_bank(b0 + 2, k)
        elif kind == "b3 s1=s0bank":
            s1, s2 = reg_in_bank(b0, k), reg_in_bank(b0 + 2, k)
        elif kind == "b3 s2=s0bank":
            s1, s2 = reg_in_bank(b0 + 1, k), reg_in_bank(b0, k)
        elif kind == "b3 s1=s2bank":
            s1, s2 = reg_in_bank(b0 + 1, k), reg_in_bank(b0 + 1, k + 1)
        elif kind == "b3 all same":
            s1, s2 = reg_in_bank(b0, k), reg_in_bank(b0, k + 1)
        elif kind == "b3 s1==s2 reg":
            s1 = s2 = reg_in_bank(b0 + 1, k)
        elif kind == "b3 s1==s0 reg":
            s1, s2 = d, reg_in_bank(b0 + 1, k)
        elif kind == "xor distinct":
            out.append(f"v_xor_b32 v{d}, v{d}, v{reg_in_bank(b0 + 1, k)}"); continue
        elif kind == "xor same bank":
            out.append(f"v_xor_b32 v{d}, v{d}, v{reg_in_bank(b0, k)}"); continue
        elif kind == "xor e64 distinct":
            out.append(f"v_xor_b32_e64 v{d}, v{d}, v{reg_in_bank(b0 + 1, k)}"); continue
        elif kind == "b3 dst other":      # not in place: dst chain register in another bank than src0
            # chains of two registers alternating: v8 <- f(v9..), v9 <- f(v8..): distance 4 via 8 regs
            src0 = 8 + ((i + 1) % 8)
            s1, s2 = reg_in_bank(src0 % 4 + 1, k), reg_in_bank(src0 % 4 + 2, k)
            out.append(f"v_bitop3_b32 v{d}, v{src0}, v{s1}, v{s2} bitop3:0x96"); continue
        elif kind == "alignbit distinct":
            out.append(f"v_alignbit_b32 v{d}, v{d}, v{reg_in_bank(b0 + 1, k)}, 7"); continue
        elif kind == "alignbit same bank":
            out.append(f"v_alignbit_b32 v{d}, v{d}, v{reg_in_bank(b0, k)}, 7"); continue
        elif kind == "alignbit vshift":
            out.append(f"v_alignbit_b32 v{d}, v{d}, v{reg_in_bank(b0 + 1, k)}, v{reg_in_bank(b0 + 2, k)}"); continue
        elif kind.startswith("dist "):
            c = int(kind.split()[1])
            d = 8 + (i % c)
            b0 = d % 4
            out.append(f"v_bitop3_b32 v{d}, v{d}, v{reg_in_bank(b0 + 1, k)}, v{reg_in_bank(b0 + 2, k)} bitop3:0x96"); continue
        elif kind.startswith("mix "):
            # every m-th instruction is a half-rate v_alignbit, 8 chains
            m = int(kind.split()[1])
            if i % m == m - 1:
                out.append(f"v_alignbit_b32 v{d}, v{d}, v{reg_in_bank(b0 + 1, k)}, 7"); continue
            s1, s2 = reg_in_bank(b0 + 1, k), reg_in_bank(b0 + 2, k)
        elif kind.startswith("burst "):
            # the first b instructions of every 256 are half-rate v_alignbit, the rest full rate
            b = int(kind.split()[1])
            if i % 256 < b:
                out.append(f"v_alignbit_b32 v{d}, v{d}, v{reg_in_bank(b0 + 1, k)}, 7"); continue
            s1, s2 = reg_in_bank(b0 + 1, k), reg_in_bank(b0 + 2, k)
        elif kind.startswith("hr "):
            # every 16th instruction is the named half-rate candidate
            m = kind.split()[1]
            if i % 16 == 15:
                r1 = reg_in_bank(b0 + 1, k)
                op = {"lshl": f"v_lshlrev_b32 v{d}, 3, v{d}", "lshr": f"v_lshrrev_b32 v{d}, 3, v{d}",
                      "mul24": f"v_mul_u32_u24 v{d}, v{d}, v{r1}", "bfe": f"v_bfe_u32 v{d}, v{d}, 3, 9",
                      "dpp": f"v_mov_b32_dpp v{d}, v{r1} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf",
                      "mov": f"v_mov_b32 v{d}, v{r1}", "add": f"v_add_u32 v{d}, v{d}, v{r1}",
                      "lshlor": f"v_lshl_or_b32 v{d}, v{d}, 3, v{r1}", "andor": f"v_and_or_b32 v{d}, v{d}, v{r1}, v{r1}",
                      "cndmask": f"v_cndmask_b32 v{d}, v{d}, v{r1}, vcc", "readlane": f"v_readfirstlane_b32 s30, v{d}",
                      "perm": f"v_perm_b32 v{d}, v{d}, v{r1}, v{r1}",
                      "alignbyte": f"v_alignbyte_b32 v{d}, v{d}, v{r1}, 1",
                      "bfi": f"v_bfi_b32 v{d}, v{d}, v{r1}, v{r1}",
                      "lshl_add": f"v_lshl_add_u32 v{d}, v{d}, 3, v{r1}",
                      "add_lshl": f"v_add_lshl_u32 v{d}, v{d}, v{r1}, 3",
                      "pk_lshl16": f"v_pk_lshlrev_b16 v{d}, 3, v{d}",
                      "lshl16": f"v_lshlrev_b16 v{d}, 3, v{d}",
                      "mul_lo16": f"v_mul_lo_u16 v{d}, v{d}, v{r1}",
                      "mad24": f"v_mad_u32_u24 v{d}, v{d}, v{r1}, v{r1}",
                      "sdwa_mov": f"v_mov_b32_sdwa v{d}, v{r1} dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0",
                      "lshl64": f"v_lshlrev_b64 v[{d & ~1}:{(d & ~1) + 1}], 3, v[{d & ~1}:{(d & ~1) + 1}]",
                      "ashr": f"v_ashrrev_i32 v{d}, 3, v{d}",
                      "bfrev": f"v_bfrev_b32 v{d}, v{d}",
                      "cvt_ubyte": f"v_cvt_f32_ubyte1 v{d}, v{d}",
                      "pk_add_f32": f"v_pk_add_f32 v[{d & ~1}:{(d & ~1) + 1}], v[{d & ~1}:{(d & ~1) + 1}], v[{r1 & ~1}:{(r1 & ~1) + 1}]",
                      "fma": f"v_fma_f32 v{d}, v{d}, v{r1}, v{r1}",
                      "max": f"v_max_f32 v{d}, v{d}, v{r1}",
                      "max3": f"v_max3_f32 v{d}, v{d}, v{r1}, v{r1}",
                      "cmp": f"v_cmp_le_f32 vcc, v{d}, v{r1}",
                      "sad": f"v_sad_u32 v{d}, v{d}, v{r1}, v{r1}",
                      "sub": f"v_sub_u32 v{d}, v{d}, v{r1}",
                      "add3": f"v_add3_u32 v{d}, v{d}, v{r1}, v{r1}",
                      "or3": f"v_or3_b32 v{d}, v{d}, v{r1}, v{r1}",
                      "add_f32": f"v_add_f32 v{d}, v{d}, v{r1}",
                      "cvt_i32": f"v_cvt_i32_f32 v{d}, v{d}",
                      "lshl_by1_add": f"v_add_u32 v{d}, v{d}, v{d}",
                      "xad": f"v_xad_u32 v{d}, v{d}, v{r1}, v{r1}",
                      "mul_lo": f"v_mul_lo_u32 v{d}, v{d}, v{r1}",
                      "bcnt": f"v_bcnt_u32_b32 v{d}, v{d}, v{r1}",
                      "lshlrev_v": f"v_lshlrev_b32 v{d}, v{r1}, v{d}",
                      "bpermute": f"ds_bpermute_b32 v200, v201, v{d}",
                      "ds_read": "ds_read_b32 v200, v201",
                      "ds_read_un": "ds_read_b32 v200, v201 offset:3",
                      "ds_read2": "ds_read2_b32 v[202:203], v201 offset1:1",
                      "ds_swizzle": f"ds_swizzle_b32 v200, v{d} offset:swizzle(SWAP,1)",
                      "permlane": f"v_permlane32_swap_b32 v200, v{d}",
                      "readlane": f"v_readlane_b32 s30, v{d}, 3"}[m]
                out.append(op); continue
            s1, s2 = reg_in_bank(b0 + 1, k), reg_in_bank(b0 + 2, k)
        elif kind.startswith("after "):
            # a half-rate v_alignbit every m instructions, followed by a candidate re-phasing instruction
            _, m, what = kind.split()
            m = int(m)
            if i % m == m - 2:
                out.append(f"v_alignbit_b32 v{d}, v{d}, v{reg_in_bank(b0 + 1, k)}, 7"); continue
            if i % m == m - 1:
                out.append({"branch": "s_branch 0", "nop": "s_nop 0", "nop7": "s_nop 7", "sleep1": "s_sleep 1",
                            "cbranch": "s_cbranch_vccz 0", "setpc": "s_nop 0", "barrier": "s_barrier",
                            "waitcnt": "s_waitcnt vmcnt(0) expcnt(0) lgkmcnt(0)"}[what]); continue
            s1, s2 = reg_in_bank(b0 + 1, k), reg_in_bank(b0 + 2, k)
        elif kind.startswith("burst1 "):
            # ONE burst of b half-rate instructions per body, starting at position p0
            _, b, p0 = kind.split()
            b, p0 = int(b), int(p0)
            if p0 <= i < p0 + b:
                out.append(f"v_alignbit_b32 v{d}, v{d}, v{reg_in_bank(b0 + 1, k)}, 7"); continue
            s1, s2 = reg_in_bank(b0 + 1, k), reg_in_bank(b0 + 2, k)
        elif kind.startswith("burstb "):
            # b alignbits, then a re-phasing instruction, then full-rate code; period 512
            _, b, what = kind.split()
            b = int(b)
            if i % 512 < b:
                out.append(f"v_alignbit_b32 v{d}, v{d}, v{reg_in_bank(b0 + 1, k)}, 7"); continue
            if i % 512 == b:
                out.append({"branch": "s_branch 0", "nop": "s_nop 0", "sleep1": "s_sleep 1", "none": "v_nop"}[what]); continue
            s1, s2 = reg_in_bank(b0 + 1, k), reg_in_bank(b0 + 2, k)
        elif kind.startswith("fix "):
            # a half-rate v_alignbit every 64 instructions, FOLLOWED by a candidate re-sync instruction
            m = kind.split()[1]
            if i % 64 == 62:
                out.append(f"v_alignbit_b32 v{d}, v{d}, v{reg_in_bank(b0 + 1, k)}, 7"); continue
            if i % 64 == 63:
                out.append({"snop0": "s_nop 0", "snop3": "s_nop 3", "setprio": "s_setprio 0", "vnop": "v_nop",
                            "sadd": "s_add_u32 s30, s30, 1", "waitcnt": "s_waitcnt lgkmcnt(15)",
                            "smov": "s_mov_b32 s30, 0", "sleep": "s_sleep 0"}[m]); continue
            s1, s2 = reg_in_bank(b0 + 1, k), reg_in_bank(b0 + 2, k)
        elif kind.startswith("dst "):
            # destinations rotate over many registers (never read); sources from v16..v99
            m = kind.split()[1]
            srcs = list(range(16, 100))
            a, b, c = srcs[(3 * i) % 84], srcs[(3 * i + 1) % 84], srcs[(3 * i + 2) % 84]
            if m == "rot64":
                dd_ = 100 + (i % 64)
            elif m == "bank0":
                dd_ = 100 + 4 * (i % 16)
            elif m == "rot64mix":      # plus a half-rate alignbit every 16th
                dd_ = 100 + (i % 64)
                if i % 16 == 15:
                    out.append(f"v_alignbit_b32 v{dd_}, v{a}, v{b}, 7"); continue
            elif m == "inplace3":      # dst = one of the sources, all different registers each time
                dd_ = a
                out.append(f"v_bitop3_b32 v{dd_}, v{a}, v{b}, v{c} bitop3:0x96"); continue
            out.append(f"v_bitop3_b32 v{dd_}, v{a}, v{b}, v{c} bitop3:0x96"); continue
        elif kind == "xormix":
            # alternate 8-byte v_bitop3 and 4-byte v_xor
            if i % 2:
                out.append(f"v_xor_b32 v{d}, v{d}, v{reg_in_bank(b0 + 1, k)}"); continue
            s1, s2 = reg_in_bank(b0 + 1, k), reg_in_bank(b0 + 2, k)
        elif kind.startswith("nopevery "):
            m = int(kind.split()[1])
            if i % m == m - 1:
                out.append("s_nop 0"); continue
            s1, s2 = reg_in_bank(b0 + 1, k), reg_in_bank(b0 + 2, k)
        elif kind == "ldswait":
            if i % 256 == 100:
                out.append("ds_read_b32 v200, v201"); continue
            if i % 256 == 110:
                out.append("s_waitcnt lgkmcnt(0)"); continue
            s1, s2 = reg_in_bank(b0 + 1, k), reg_in_bank(b0 + 2, k)
        elif kind.startswith("rare "):
            # ONE half-rate instruction per body of N (position 0), the rest full rate
            m = kind.split()[1]
            if i == 0:
                op = {"alignbit": f"v_alignbit_b32 v{d}, v{d}, v{reg_in_bank(b0 + 1, k)}, 7",
                      "nop": "s_nop 0", "sleep": "s_sleep 1",
                      }[m]
                out.append(op); continue
            s1, s2 = reg_in_bank(b0 + 1, k), reg_in_bank(b0 + 2, k)
        elif kind == "mov_dpp":
            out.append(f"v_mov_b32_dpp v{d}, v{reg_in_bank(b0 + 1, k)} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"); continue
        else:
            raise ValueError(kind)
        out.append(f"v_bitop3_b32 v{d}, v{d}, v{s1}, v{s2} bitop3:0x96")
    return "\\n".join(out)

kinds = ["b3 distinct +skew", "burst1 16 0 +skew", "burst1 64 0 +skew", "burst1 16 8 +skew", "burst1 16 64 +skew", "burst1 16 512 +skew", "burst1 64 512 +skew", "burst1 1 1 +skew", "burst1 1 4 +skew", "burst1 1 16 +skew"]
src = ['// GENERATED by tools/gen_ubench_bankrules.py -- do not edit', '#include <hip/hip_runtime.h>',
       '#include <cstdio>', '#include <vector>', '#include <map>', '#include <algorithm>', '']
clob = ", ".join(f'"v{i}"' for i in range(8, 204)) + ', "vcc", "s30"'
for k, kind in enumerate(kinds):
    src.append(f'__global__ __launch_bounds__(64) void k{k}(unsigned *out, unsigned long long *info, int iters)\n{{')
    src.append('    unsigned seed = threadIdx.x * 2654435761u + blockIdx.x;')
    src.append('    asm volatile("' + "\\n".join(f"v_mov_b32 v{i}, %0" for i in range(8, 200)) + f'" :: "v"(seed) : {clob});')
    if kind.endswith(" +prio"):
        src.append('    if (__builtin_amdgcn_s_getreg(63492) & 1) asm volatile("s_setprio 3");')
    if kind.endswith(" +far"):
        src.append('    if (__builtin_amdgcn_s_getreg(63492) & 1) {   // odd slot: half a body ahead')
        half = "\\n".join(body(kind).split("\\n")[:N // 2])
        src.append(f'        asm volatile("{half}" ::: {clob});')
        src.append('    }')
    if kind.startswith("skew ") or kind.endswith(" +skew"):
        n_extra = int(kind.split()[1]) if kind.startswith("skew ") else 37
        src.append('    if (__builtin_amdgcn_s_getreg(63492) & 1) {   // odd hardware wave slot: run ahead / behind')
        src.append('        asm volatile("' + "\\n".join("v_bitop3_b32 v8, v8, v17, v18 bitop3:0x96" for _ in range(n_extra)) + f'" ::: {clob});')
        src.append('    }')
    src.append('    asm volatile("v_mov_b32 v201, 0" ::: "v201", "v200");')
    src.append('    const unsigned long long t0 = __builtin_amdgcn_s_memtime();')
    src.append('    for (int it = 0; it < iters; it++)')
    src.append(f'        asm volatile("{body(kind)}" ::: {clob});')
    src.append('    const unsigned long long t1 = __builtin_amdgcn_s_memtime();')
    src.append(f'    unsigned r_; asm volatile("v_xor_b32 %0, v8, v9\\nv_xor_b32 %0, %0, v12" : "=v"(r_) :: {clob});')
    src.append('    out[blockIdx.x * 64 + threadIdx.x] = r_;')
    src.append('    if (threadIdx.x == 0) { info[blockIdx.x * 3] = t0; info[blockIdx.x * 3 + 1] = t1;')
    src.append('        info[blockIdx.x * 3 + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492); }')
    src.append('}\n')
src.append(r'''
typedef void (*kern_t)(unsigned *, unsigned long long *, int);
static void run(kern_t k, const char *name, int n, unsigned *out, unsigned long long *info)
{
    const int total = 1 << 17, iters = total / n, grid = 2048;
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, info, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 3);
    (void)hipMemcpy(h.data(), info, h.size() * 8, hipMemcpyDeviceToHost);
    std::map<unsigned long long, std::vector<int>> simd;
    for (int b = 0; b < grid; b++) {
        const unsigned long long hw = h[b * 3 + 2]; const unsigned id = (unsigned)hw;
        simd[((hw >> 32) << 16) | (((id >> 13) & 7) << 12) | (((id >> 12) & 1) << 11) | (((id >> 8) & 15) << 4) | ((id >> 4) & 3)].push_back(b);
    }
    double older = 0, younger = 0, both = 0; int pairs = 0, odd = 0;
    for (auto &kv : simd) {
        if (kv.second.size() != 2) { odd++; continue; }
        int a = kv.second[0], b = kv.second[1];
        if (h[a * 3] > h[b * 3]) std::swap(a, b);                 // a started first
        older += (double)(h[a * 3 + 1] - h[a * 3]);
        younger += (double)(h[b * 3 + 1] - h[b * 3]);
        both += (double)(std::max(h[a * 3 + 1], h[b * 3 + 1]) - std::min(h[a * 3], h[b * 3]));
        pairs++;
    }
    const double ni = (double)iters * n;
    printf("%-22s %4d SIMD pairs (%d odd): cycles/instr older %.2f, younger %.2f, SIMD %.2f\n", name, pairs, odd,
           older / pairs / ni, younger / pairs / ni, both / pairs / (2 * ni));
}
int main()
{
    unsigned *out; unsigned long long *info;
    setvbuf(stdout, NULL, _IOLBF, 0);
    (void)hipMalloc(&out, 2048 * 64 * 4); (void)hipMalloc(&info, 2048 * 3 * 8);
''')
for k, kind in enumerate(kinds):
    src.append(f'    run(k{k}, "{kind}", {N}, out, info);')
src.append('    return 0;\n}')
Path(__file__).resolve().parent.joinpath("ubench_bankrules.hip").write_text("\n".join(src) + "\n")
print("wrote tools/ubench_bankrules.hip")
