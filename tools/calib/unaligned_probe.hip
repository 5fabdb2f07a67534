// unaligned_probe.hip -- does a global_load_ushort / global_load_dword at an odd byte address return the bytes at that address on this
// device?  (The 3-bit soft stream is read with 16-bit loads at arbitrary byte offsets.)  Prints "unaligned ok" or the first mismatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
__global__ void k(const uint8_t* p, uint32_t* out16, uint32_t* out32, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out16[i] = *reinterpret_cast<const uint16_t*>(p + i);
    uint32_t v; __builtin_memcpy(&v, p + i, 4);
    out32[i] = v;
}
int main()
{
    const int n = 4096;
    std::vector<uint8_t> h(n + 8);
    for (int i = 0; i < n + 8; i++) h[i] = (uint8_t)(i * 37 + (i >> 3));
    uint8_t* d; uint32_t *o16, *o32;
    hipMalloc(&d, n + 8); hipMalloc(&o16, 4 * n); hipMalloc(&o32, 4 * n);
    hipMemcpy(d, h.data(), n + 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o16, o32, n);
    std::vector<uint32_t> r16(n), r32(n);
    hipMemcpy(r16.data(), o16, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(r32.data(), o32, 4 * n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; i++) {
        uint16_t w16; uint32_t w32; memcpy(&w16, &h[i], 2); memcpy(&w32, &h[i], 4);
        if (r16[i] != w16 || r32[i] != w32) { printf("mismatch at byte offset %d: u16 %x want %x, u32 %x want %x\n", i, r16[i], w16, r32[i], w32); return 1; }
    }
    printf("unaligned ok\n");
    return 0;
}
