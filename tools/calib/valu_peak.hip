// valu_peak.hip -- what the chip sustains, in wave-instructions per second, for the instruction kinds the receive path is made of
// (round 4, VERDICT r3 #2: "the achievable ceiling for this instruction mix ... from counters").  Every kernel is a long unrolled run of ONE
// instruction kind (or of the trellis step's mix) on independent registers; grid = all CUs x W waves per SIMD; the rate is
// instructions x waves / wall time (HIP events), i.e. it includes whatever clock the chip holds under that load.
//   hipcc --offload-arch=gfx950 -O3 -o valu_peak valu_peak.hip && ./valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND> __global__ void __launch_bounds__(256) k_probe(unsigned* out, int iters)
{
    unsigned a = threadIdx.x, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, e = a * 5 + 3, f = a + 11, g = 0x00030003u, h = a * 7;
    for (int i = 0; i < iters; i++) {
        if (KIND == 0)          // VOP2: v_add_u32
            asm volatile(REP64("v_add_u32 %0, %6, %0\n\tv_add_u32 %1, %6, %1\n\tv_add_u32 %2, %6, %2\n\tv_add_u32 %3, %6, %3\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "v"(g));
        else if (KIND == 1)     // VOP3P: v_pk_min_u16
            asm volatile(REP64("v_pk_min_u16 %0, %6, %0\n\tv_pk_min_u16 %1, %6, %1\n\tv_pk_min_u16 %2, %6, %2\n\tv_pk_min_u16 %3, %6, %3\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "v"(g));
        else if (KIND == 2)     // VOP2 + DPP: v_add_u32_dpp (row_ror:8)
            asm volatile(REP64("v_add_u32_dpp %0, %4, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %1, %5, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                               "v_add_u32_dpp %2, %4, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_add_u32_dpp %3, %5, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "v"(g));
        else if (KIND == 3)     // the trellis step of k_viterbi16: per register add, add_dpp, pk_min (x4) + xor, sub for the operands = 14
            asm volatile(REP64("v_xor_b32 %4, %6, %4\n\tv_sub_u32 %5, %6, %4\n\t"
                               "v_add_u32 %7, %4, %0\n\tv_add_u32_dpp %0, %1, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_pk_min_u16 %0, %7, %0\n\t"
                               "v_add_u32 %7, %5, %1\n\tv_add_u32_dpp %1, %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_pk_min_u16 %1, %7, %1\n\t"
                               "v_add_u32 %7, %4, %2\n\tv_add_u32_dpp %2, %3, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_pk_min_u16 %2, %7, %2\n\t"
                               "v_add_u32 %7, %5, %3\n\tv_add_u32_dpp %3, %2, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_pk_min_u16 %3, %7, %3\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "v"(g), "v"(h));
        else if (KIND == 4)     // VOP3 three-operand: v_add3_u32
            asm volatile(REP64("v_add3_u32 %0, %6, %0, %4\n\tv_add3_u32 %1, %6, %1, %5\n\tv_add3_u32 %2, %6, %2, %4\n\tv_add3_u32 %3, %6, %3, %5\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "v"(g));
        else                    // SALU beside nothing: s_add_u32 on four registers
            asm volatile(REP64("s_add_u32 s20, s20, s24\n\ts_add_u32 s21, s21, s24\n\ts_add_u32 s22, s22, s24\n\ts_add_u32 s23, s23, s24\n\t")
                         : "+v"(a) : : "s20", "s21", "s22", "s23", "s24", "scc");
    }
    if (a + b + c + d + e + f == 0x12345u) out[0] = a;
}

struct Probe { const char* name; void (*k)(unsigned*, int); int per_iter; };

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned* d; hipMalloc(&d, 4096);
    const Probe probes[] = {
        { "v_add_u32 (VOP2)", k_probe<0>, 256 }, { "v_pk_min_u16 (VOP3P)", k_probe<1>, 256 }, { "v_add_u32_dpp (DPP)", k_probe<2>, 256 },
        { "trellis step mix (4 add, 4 add_dpp, 4 pk_min, xor, sub)", k_probe<3>, 64 * 14 }, { "v_add3_u32 (VOP3)", k_probe<4>, 256 }, { "s_add_u32 (SALU)", k_probe<5>, 256 },
    };
    printf("# %s, %d CUs; G wave-instructions/s = instructions x waves / wall time (HIP events), ~25 ms per measurement after a 25 ms warm-up of the same kernel\n", prop.name, cus);
    printf("%-58s %12s %12s %12s %12s\n", "kind", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (const Probe& p : probes) {
        printf("%-58s", p.name);
        for (int wps : { 1, 2, 4, 8 }) {
            const int blocks = cus * wps;                                         // 256-thread blocks: one wave per SIMD each
            int iters = 2000;
            float ms = 0;
            for (int pass = 0; pass < 3; pass++) {                                // calibrate to ~25 ms, then warm, then measure
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(p.k, dim3(blocks), dim3(256), 0, 0, d, iters);
                hipEventRecord(e1, 0); hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
                if (pass == 0) iters = (int)(iters * 25.0f / (ms > 0.01f ? ms : 0.01f)) + 1;
            }
            const double insts = (double)iters * p.per_iter * (double)blocks * 4.0;
            printf(" %12.1f", insts / (ms * 1e-3) / 1e9);
        }
        printf("\n");
    }
    return 0;
}
