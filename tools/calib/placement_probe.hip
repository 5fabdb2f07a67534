// placement_probe.hip -- where does the dispatcher put the waves of a launch shaped like the trellis kernel's?  (round 4: two side-by-side launches of
// 1024 eight-frame waves finish 20 % sooner than one launch of 2048, profiles/r04_w_corun.txt.)  Workgroups of ONE wave with 20288 bytes of LDS (eight fit a
// CU) and, for comparison, workgroups of FOUR waves with 4 x 20288 bytes (two fit a CU, one wave per SIMD each by construction); every wave records the
// XCC / SE / CU / SIMD it runs on (s_getreg HW_ID, XCC_ID) and spins until all waves of the launch are resident, or a time-out.  Prints how many SIMDs
// hold 0, 1, 2, 3, 4 ... waves of the launch.
//   hipcc --offload-arch=gfx950 -O3 -o placement_probe placement_probe.hip && ./placement_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

constexpr int kLds = 20288;

template <int WAVES> __global__ void __launch_bounds__(64 * WAVES) k_place(unsigned* where, unsigned* arrived, unsigned total, long long spin_cycles)
{
    extern __shared__ unsigned char lds[];
    const unsigned w = blockIdx.x * WAVES + threadIdx.x / 64;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lds[threadIdx.x * 4] = (unsigned char)w;                                    // (the allocation is used)
    if ((threadIdx.x & 63) == 0) { where[2 * w] = hw; where[2 * w + 1] = xcc; atomicAdd(arrived, 1u); }
    const long long t0 = clock64();
    while (clock64() - t0 < spin_cycles) {                                      // stay resident for a while: the whole launch is on the chip together
        if (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= total && clock64() - t0 > spin_cycles / 4) break;
        __builtin_amdgcn_s_sleep(32);
    }
    if (lds[threadIdx.x * 4] == 255 && w == 0xFFFFFFFFu) where[0] = 0;
}

static void report(const char* name, const std::vector<unsigned>& h, unsigned nw)
{
    std::map<unsigned long long, int> per_simd, per_cu;
    for (unsigned w = 0; w < nw; w++) {
        const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xF;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned long long cuid = ((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu;
        per_cu[cuid]++; per_simd[(cuid << 2) | simd]++;
    }
    std::map<int, int> hs, hc;
    for (auto& kv : per_simd) hs[kv.second]++;
    for (auto& kv : per_cu) hc[kv.second]++;
    printf("%-44s %5u waves on %3zu CUs, %4zu SIMDs;  waves per SIMD:", name, nw, per_cu.size(), per_simd.size());
    for (auto& kv : hs) printf("  %d x%d", kv.first, kv.second);
    printf("   waves per CU:");
    for (auto& kv : hc) printf("  %d x%d", kv.first, kv.second);
    printf("\n");
}

int main()
{
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    printf("# %s, %d CUs (HW_ID: simd bits 5:4, cu 11:8, sh 12, se 15:13; XCC_ID 3:0)\n", prop.name, prop.multiProcessorCount);
    unsigned *d_where, *d_arr;
    hipMalloc(&d_where, 8 * 8192); hipMalloc(&d_arr, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_place<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kLds);
    for (unsigned nw : { 512u, 1024u, 1536u, 2048u }) {
        std::vector<unsigned> h(2 * nw);
        for (int shape = 0; shape < 2; shape++) {
            hipMemset(d_arr, 0, 4); hipMemset(d_where, 0xFF, 8 * nw);
            const unsigned total = nw <= 2048 ? nw : 2048;                      // more than 2048 cannot be resident together
            if (shape == 0) hipLaunchKernelGGL(k_place<1>, dim3(nw), dim3(64), kLds, 0, d_where, d_arr, total, 2000000LL);
            else            hipLaunchKernelGGL(k_place<4>, dim3(nw / 4), dim3(256), 4 * kLds, 0, d_where, d_arr, total, 2000000LL);
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); return 1; }
            hipMemcpy(h.data(), d_where, 8 * nw, hipMemcpyDeviceToHost);
            report(shape == 0 ? "one-wave workgroups, 20288 B LDS each" : "four-wave workgroups, 4 x 20288 B LDS each", h, nw);
        }
    }
    return 0;
}
