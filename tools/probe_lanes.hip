// probe_lanes.hip -- prints what each cross-lane primitive used by the Viterbi kernel delivers (lane -> source lane).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
    unsigned m = threadIdx.x;
    auto r32 = __builtin_amdgcn_permlane32_swap(m, m, false, false);
    auto r16 = __builtin_amdgcn_permlane16_swap(m, m, false, false);
    unsigned x1 = __builtin_amdgcn_update_dpp(0u, m, 0xB1, 0xF, 0xF, true);
    unsigned x2 = __builtin_amdgcn_update_dpp(0u, m, 0x4E, 0xF, 0xF, true);
    unsigned t = __builtin_amdgcn_update_dpp(m, m, 0x104, 0xF, 0x5, false);
    unsigned x4 = __builtin_amdgcn_update_dpp(t, m, 0x114, 0xF, 0xA, false);
    unsigned x8 = __builtin_amdgcn_update_dpp(0u, m, 0x128, 0xF, 0xF, true);
    unsigned* p = o + threadIdx.x * 8;
    p[0] = r32[0]; p[1] = r32[1]; p[2] = r16[0]; p[3] = r16[1]; p[4] = x1; p[5] = x2; p[6] = x4; p[7] = x8;
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 8 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[64 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[8] = {"p32[0]", "p32[1]", "p16[0]", "p16[1]", "xor1", "xor2", "xor4", "ror8"};
    for (int j = 0; j < 8; j++) { printf("%-7s", names[j]); for (int i = 0; i < 64; i++) printf(" %2u", h[i * 8 + j]); printf("\n"); }
    return 0;
}
