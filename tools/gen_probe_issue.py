#!/usr/bin/env python3
"""gen_probe_issue.py -- emits probe_issue.hip: an instruction-issue microbenchmark for gfx950.

Each probe runs an unrolled block of one instruction kind (independent registers, or one dependent chain) inside a
loop and reports core clocks per instruction per wave at 1, 2 and 4 waves per SIMD.  The Viterbi kernel is bound by
VALU issue, so these numbers (not the data sheet) decide which formulation of the butterfly is cheapest.

    python tools/gen_probe_issue.py > /tmp/probe_issue.hip && hipcc --offload-arch=gfx950 -O2 /tmp/probe_issue.hip -o tools/_probe_issue
"""
N = 32  # instructions per asm block

def ind(fmt):
    """independent: dst k, sources k (+ a constant register %7)"""
    return [fmt.format(d="%%%d" % (i % 6), a="%%%d" % (i % 6), b="%7", c="%%%d" % ((i + 3) % 6)) for i in range(N)]

def dep(fmt, nop=None):
    out = []
    for i in range(N):
        out.append(fmt.format(d="%0", a="%0", b="%7", c="%0"))
        if nop:
            out.append(nop)
    return out

PROBES = {
    "add_ind": ind("v_add_u32 {d}, {a}, {b}"),
    "add_dep": dep("v_add_u32 {d}, {a}, {b}"),
    "pkadd_ind": ind("v_pk_add_u16 {d}, {a}, {b}"),
    "pkadd_dep": dep("v_pk_add_u16 {d}, {a}, {b}"),
    "pkmin_ind": ind("v_pk_min_u16 {d}, {a}, {b}"),
    "min_ind": ind("v_min_u32 {d}, {a}, {b}"),
    "sub_ind": ind("v_sub_u32 {d}, {a}, {b}"),
    "xor_ind": ind("v_xor_b32 {d}, {a}, {b}"),
    "and_ind": ind("v_and_b32 {d}, {a}, {b}"),
    "mov_ind": ind("v_mov_b32 {d}, {c}"),
    "lshl_ind": ind("v_lshlrev_b32 {d}, 1, {a}"),
    "max_ind": ind("v_max_u32 {d}, {a}, {b}"),
    "min_i32_ind": ind("v_min_i32 {d}, {a}, {b}"),
    "min_u16_ind": ind("v_min_u16 {d}, {a}, {b}"),
    "add_u16_ind": ind("v_add_u16 {d}, {a}, {b}"),
    "add_e64_ind": ind("v_add_u32_e64 {d}, {a}, {b}"),
    "add3_ind": ind("v_add3_u32 {d}, {a}, {b}, {c}"),
    "min3_ind": ind("v_min3_u32 {d}, {a}, {b}, {c}"),
    "bfe_ind": ind("v_bfe_u32 {d}, {a}, 3, 5"),
    "lshl_add_ind": ind("v_lshl_add_u32 {d}, {a}, 1, {b}"),
    "and_or_ind": ind("v_and_or_b32 {d}, {a}, {b}, {c}"),
    "addco_vcc": ind("v_add_co_u32 {d}, vcc, {a}, {b}"),
    "addc_vcc": ind("v_addc_co_u32 {d}, vcc, {a}, {a}, vcc"),
    "cndmask_vcc": ind("v_cndmask_b32 {d}, {a}, {b}, vcc"),
    "sdwa_add": ind("v_add_u32_sdwa {d}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0"),
    "pksub_ind": ind("v_pk_sub_u16 {d}, {a}, {b}"),
    "mad_u32_u24": ind("v_mad_u32_u24 {d}, {a}, {b}, {c}"),
    "sad_u8": ind("v_sad_u8 {d}, {a}, {b}, {c}"),
    "perm_b32": ind("v_perm_b32 {d}, {a}, {b}, {c}"),
    "xad_ind": ind("v_xad_u32 {d}, {a}, {b}, {c}"),
    "dppquad_ind": ind("v_mov_b32_dpp {d}, {c} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"),
    "dppquad_dep": dep("v_mov_b32_dpp {d}, {a} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "s_nop 1"),
    "dppror8_ind": ind("v_mov_b32_dpp {d}, {c} row_ror:8 row_mask:0xf bank_mask:0xf"),
    "dppshr4_bank_ind": ind("v_mov_b32_dpp {d}, {c} row_shr:4 row_mask:0xf bank_mask:0xa"),
    "adddpp_ind": ind("v_add_u32_dpp {d}, {c}, {b} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"),
    "perm32swap_ind": ind("v_permlane32_swap_b32 {d}, {c}"),
    "perm16swap_ind": ind("v_permlane16_swap_b32 {d}, {c}"),
    "cmp_sdwa": ["v_cmp_lt_u32_sdwa s[20:21], %%%d, %%7 src0_sel:WORD_0 src1_sel:WORD_0" % (i % 6) for i in range(N)],
    "cmp_e64": ["v_cmp_lt_u32_e64 s[20:21], %%%d, %%7" % (i % 6) for i in range(N)],
    "cmp_vcc": ["v_cmp_lt_u32_e32 vcc, %%%d, %%7" % (i % 6) for i in range(N)],
    "addc_ind": ["v_addc_co_u32 %%%d, s[22:23], %%%d, %%%d, s[20:21]" % (i % 6, i % 6, i % 6) for i in range(N)],
    "cmp_addc": sum([["v_cmp_lt_u32_e64 s[20:21], %%%d, %%7" % (i % 6),
                      "v_addc_co_u32 %%%d, s[22:23], %%%d, %%%d, s[20:21]" % ((i + 3) % 6, (i + 3) % 6, (i + 3) % 6)] for i in range(N // 2)], []),
    "cndmask_ind": ["v_cndmask_b32_e64 %%%d, %%%d, %%7, s[20:21]" % (i % 6, i % 6) for i in range(N)],
    "salu_and": ["s_and_b32 s24, s24, s25" for i in range(N)],
    "salu_ind": ["s_and_b32 s%d, s%d, s30" % (24 + i % 4, 24 + i % 4) for i in range(N)],
    "s_nop0": ["s_nop 0" for i in range(N)],
    "s_pack": ["s_pack_ll_b32_b16 s%d, s%d, s30" % (24 + i % 4, 24 + i % 4) for i in range(N)],
    "bpermute_ind": sum([["ds_bpermute_b32 %%%d, %%6, %%%d" % (i % 6, i % 6)] for i in range(N)], []) + ["s_waitcnt lgkmcnt(0)"],
    "readlane": ["v_readlane_b32 s%d, %%%d, %d" % (24 + i % 4, i % 6, i) for i in range(N)],
    # one packed butterfly step as the kernel issues it (quad_perm phase, two-input), registers: %0 = U, %1/%2 hist, %3..%5 temps, %7 = soft
    "step_quad": sum([[
        "v_xor_b32 %3, %7, %6",
        "v_xad_u32 %3, %7, %6, %3",
        "v_sub_u32 %4, %7, %3",
        "v_mov_b32_dpp %5, %0 quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf",
        "v_mov_b32_dpp %0, %0 quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf",
        "v_pk_add_u16 %5, %5, %3",
        "v_pk_add_u16 %4, %0, %4",
        "v_cmp_lt_u32_sdwa s[20:21], %4, %5 src0_sel:WORD_0 src1_sel:WORD_0",
        "v_cmp_lt_u32_sdwa s[22:23], %4, %5 src0_sel:WORD_1 src1_sel:WORD_1",
        "v_pk_min_u16 %0, %5, %4",
        "v_addc_co_u32 %1, s[24:25], %1, %1, s[20:21]",
        "v_addc_co_u32 %2, s[24:25], %2, %2, s[22:23]",
        "s_nop 0",
    ] for i in range(3)], []),
    # same step in the own/partner form with a tie-break mark: one DPP move, decisions fixed up on the scalar unit
    "step_mark": sum([[
        "v_xor_b32 %3, %7, %6",
        "v_xad_u32 %3, %7, %6, %3",
        "v_sub_u32 %4, %7, %3",
        "v_mov_b32_dpp %5, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf",
        "v_pk_add_u16 %0, %0, %3",
        "v_pk_add_u16 %5, %5, %4",
        "v_cmp_lt_u32_sdwa s[20:21], %5, %0 src0_sel:WORD_0 src1_sel:WORD_0",
        "v_cmp_lt_u32_sdwa s[22:23], %5, %0 src0_sel:WORD_1 src1_sel:WORD_1",
        "v_pk_min_u16 %0, %5, %0",
        "s_xor_b64 s[20:21], s[20:21], s[26:27]",
        "s_xor_b64 s[22:23], s[22:23], s[26:27]",
        "v_addc_co_u32 %1, s[24:25], %1, %1, s[20:21]",
        "v_addc_co_u32 %2, s[24:25], %2, %2, s[22:23]",
    ] for i in range(3)], []),
}

def count(lines):
    return sum(1 for l in lines if not l.startswith("s_waitcnt"))

print("// generated by tools/gen_probe_issue.py -- do not edit")
print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <vector>\n#include <algorithm>")
for name, lines in PROBES.items():
    body = "\\n\\t".join(lines)
    print("""
__global__ void __launch_bounds__(1024) k_%s(long long* out, int iters)
{
    unsigned v0 = threadIdx.x, v1 = v0 * 3 + 1, v2 = v0 ^ 0x55, v3 = v0 + 7, v4 = v0 * 5, v5 = v0 + 11, v6 = (threadIdx.x ^ 1) * 4, v7 = 0x00030003;
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++)
        asm volatile("%s"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5) : "v"(v6), "v"(v7)
                     : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "vcc", "memory");
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    if (v0 + v1 + v2 + v3 + v4 + v5 == 0x12345) out[0] = 0;
}""" % (name, body))

print("""
struct Probe { const char* name; void (*k)(long long*, int); int n; };
int main()
{
    const Probe probes[] = {""")
for name, lines in PROBES.items():
    print('        { "%s", k_%s, %d },' % (name, name, count(lines)))
print("""    };
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    long long* d; hipMalloc(&d, 8 * 65536);
    const int iters = 2000;
    printf("%-18s %8s %8s %8s   (core clocks per instruction per wave; %d CUs, clockRate %d kHz)\\n", "probe", "1w/SIMD", "2w/SIMD", "4w/SIMD", cus, prop.clockRate);
    for (const Probe& p : probes) {
        printf("%-18s", p.name);
        for (int wps : { 1, 2, 4 }) {
            const int threads = 256 * wps, blocks = cus;
            hipLaunchKernelGGL(p.k, dim3(blocks), dim3(threads), 0, 0, d, 10);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(p.k, dim3(blocks), dim3(threads), 0, 0, d, iters);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> h(blocks * threads / 64);
            hipMemcpy(h.data(), d, 8 * h.size(), hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            const double med = (double)h[h.size() / 2] / ((double)iters * p.n);
            printf(" %8.2f", med);
            (void)ms;
        }
        printf("\\n");
    }
    // clock64 tick rate against wall time
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_add_dep, dim3(cus), dim3(256), 0, 0, d, 200000);
        hipEventRecord(e1, 0); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long t; hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
        printf("clock64: %.1f MHz (ticks %lld in %.3f ms)\\n", t / (ms * 1e3), t, ms);
        for (int rep = 0; rep < 12; rep++) {                       // does the shader clock ramp under sustained load?
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_add_dep, dim3(cus), dim3(1024), 0, 0, d, 400000);
            hipEventRecord(e1, 0); hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost);
            printf("  sustained rep %d: %.1f MHz (%.3f ms)\\n", rep, t / (ms * 1e3), ms);
        }
    }
    return 0;
}""")
