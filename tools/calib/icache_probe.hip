// icache_probe.hip -- does the code footprint of the kernels that share a CU bound the receive path?  (round 4: the same captures decode in
// 0.34 ms per 4096 when two large calls alternate on the chip and in 0.39-0.50 ms when four to eight calls' kernels -- k_scan 12 + 10 + ... KB,
// k_frame 8 KB, k_viterbi16's 21 KB window loop, k_finish 3.5 KB -- are resident together; profiles/r04_r_call_size.txt.)
// k_foot<KB> is a loop whose body is KB kilobytes of straight-line v_add_u32 on four independent registers (256 instructions per KB);
// (1) one kernel on every CU, rate against footprint; (2) two, three, four DIFFERENT kernels (distinct code) resident together, one wave of each
// per SIMD, rate against the summed footprint.  Rates in G wave-instructions/s of wall time, like valu_peak.hip.
//   hipcc --offload-arch=gfx950 -O3 -o icache_probe icache_probe.hip && ./icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP4(x) x x x x
#define REP64(x) REP4(REP4(REP4(x)))

// TAG makes otherwise identical kernels distinct functions at distinct addresses
template <int KB, int TAG> __global__ void __launch_bounds__(256) k_foot(unsigned* out, int iters)
{
    unsigned a = threadIdx.x + TAG, b = a * 3 + 1, c = a ^ 0x55, d = a + 7, g = 0x00030003u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < KB; k++)                                            // 64 x 4 instructions x 4 bytes = 1 KB
            asm volatile(REP64("v_add_u32 %0, %4, %0\n\tv_add_u32 %1, %4, %1\n\tv_add_u32 %2, %4, %2\n\tv_add_u32 %3, %4, %3\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(g));
    }
    if (a + b + c + d == 0x12345u) out[0] = a;
}

typedef void (*kern_t)(unsigned*, int);
struct K { kern_t k; int kb; };

static double run(const std::vector<K>& ks, int cus, int wps, unsigned* d, hipStream_t* st, hipEvent_t e0, hipEvent_t e1, double target_ms)
{
    // every kernel of the set on its own stream, cus x wps blocks of 256 threads each (one wave per SIMD per block); same instruction count each
    std::vector<int> iters(ks.size());
    double total_inst = 0;
    float ms = 0;
    double scale = 1.0;
    for (int pass = 0; pass < 3; pass++) {
        total_inst = 0;
        for (size_t i = 0; i < ks.size(); i++) { iters[i] = (int)(scale * 40000.0 / ks[i].kb) + 1; total_inst += (double)iters[i] * ks[i].kb * 256.0 * cus * wps * 4.0; }
        hipDeviceSynchronize();
        hipEventRecord(e0, st[0]);
        for (size_t i = 1; i < ks.size(); i++) hipStreamWaitEvent(st[i], e0, 0);
        for (size_t i = 0; i < ks.size(); i++) hipLaunchKernelGGL(ks[i].k, dim3(cus * wps), dim3(256), 0, st[i], d, iters[i]);
        hipDeviceSynchronize();
        hipEventRecord(e1, st[0]); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (pass == 0) scale = target_ms / (ms > 0.01f ? ms : 0.01f);
    }
    return total_inst / (ms * 1e-3) / 1e9;
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned* d; hipMalloc(&d, 4096);
    hipStream_t st[4]; for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("# %s, %d CUs; G wave-instructions/s of v_add_u32 (VOP2; 1166 is the chip's rate from 2 waves per SIMD, valu_peak.hip)\n", prop.name, cus);
    printf("# (1) ONE kernel, loop body of the given size\n%-10s %10s %10s %10s\n", "body KB", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
    const K singles[] = { { k_foot<4, 0>, 4 }, { k_foot<8, 0>, 8 }, { k_foot<16, 0>, 16 }, { k_foot<24, 0>, 24 }, { k_foot<32, 0>, 32 }, { k_foot<48, 0>, 48 },
                          { k_foot<56, 0>, 56 }, { k_foot<64, 0>, 64 }, { k_foot<80, 0>, 80 }, { k_foot<96, 0>, 96 }, { k_foot<128, 0>, 128 } };
    for (const K& k : singles) {
        printf("%-10d", k.kb);
        for (int wps : { 1, 2, 4 }) printf(" %10.1f", run({ k }, cus, wps, d, st, e0, e1, 20.0));
        printf("\n");
    }
    printf("# (2) SEVERAL different kernels resident together, one wave of each per SIMD (each on its own stream)\n%-28s %10s %10s\n", "bodies KB", "sum KB", "G inst/s");
    const std::vector<std::vector<K>> sets = {
        { { k_foot<8, 1>, 8 }, { k_foot<8, 2>, 8 } },
        { { k_foot<16, 1>, 16 }, { k_foot<16, 2>, 16 } },
        { { k_foot<24, 1>, 24 }, { k_foot<24, 2>, 24 } },
        { { k_foot<32, 1>, 32 }, { k_foot<32, 2>, 32 } },
        { { k_foot<24, 1>, 24 }, { k_foot<48, 1>, 48 } },
        { { k_foot<48, 1>, 48 }, { k_foot<48, 2>, 48 } },
        { { k_foot<8, 1>, 8 }, { k_foot<8, 2>, 8 }, { k_foot<8, 3>, 8 }, { k_foot<8, 4>, 8 } },
        { { k_foot<16, 1>, 16 }, { k_foot<16, 2>, 16 }, { k_foot<16, 3>, 16 }, { k_foot<16, 4>, 16 } },
        { { k_foot<24, 1>, 24 }, { k_foot<24, 2>, 24 }, { k_foot<24, 3>, 24 }, { k_foot<24, 4>, 24 } },
        { { k_foot<24, 1>, 24 }, { k_foot<32, 1>, 32 }, { k_foot<8, 1>, 8 }, { k_foot<4, 1>, 4 } },      // about the receive call's own kernels
    };
    for (const auto& s : sets) {
        char name[64]; int n = 0, sum = 0;
        for (const K& k : s) { n += snprintf(name + n, sizeof name - n, "%s%d", n ? " + " : "", k.kb); sum += k.kb; }
        printf("%-28s %10d %10.1f\n", name, sum, run(s, cus, 1, d, st, e0, e1, 20.0));
    }
    return 0;
}
