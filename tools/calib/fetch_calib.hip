// fetch_calib.hip -- what rocprofv3's FETCH_SIZE / WRITE_SIZE report, per access shape, for a KNOWN number of bytes.
// MI355X_MICROARCH.md (HBM section): FETCH_SIZE is half the bytes of a wide (16 B per lane) coalesced read on gfx950 and "other
// access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access pattern".  The receive
// kernels use three read shapes (16 B per lane, 4 B per lane, scalar loads through the constant cache) and three write shapes
// (16 B, 4 B, 1 B per lane); each kernel below moves exactly BYTES bytes in one of them.
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- ./fetch_calib        (and a second run with --pmc WRITE_SIZE)
// tools/summarize_pmc.py reads the factors from profiles/*_calibration.json (tools/calib/summarize_calib.py writes it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

static const size_t BYTES = 1ull << 30;                                 // 1 GiB: four times the Infinity Cache

__global__ void __launch_bounds__(256) calib_read16(const uint4* __restrict__ p, size_t n, uint32_t* sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void __launch_bounds__(256) calib_read4(const uint32_t* __restrict__ p, size_t n, uint32_t* sink)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= p[i];
    if (acc == 0x12345678u) *sink = acc;
}
// scalar loads: every wave walks its own contiguous region 32 bytes (s_load_dwordx8) at a time -- k_viterbi's shape
__global__ void __launch_bounds__(256) calib_read_scalar(const uint32_t* __restrict__ p, size_t words_per_wave, uint32_t* sink)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (size_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t* q = p + wave * words_per_wave;
    uint32_t acc = 0;
    for (size_t i = 0; i < words_per_wave; i += 8) {
        uint32_t a, b, c, d, e, f, g, h;
        const uint32_t* r = q + i;
        asm volatile("s_load_dwordx8 s[20:27], %8, 0x0\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, s20\n\ts_mov_b32 %1, s21\n\ts_mov_b32 %2, s22\n\ts_mov_b32 %3, s23\n\t"
                     "s_mov_b32 %4, s24\n\ts_mov_b32 %5, s25\n\ts_mov_b32 %6, s26\n\ts_mov_b32 %7, s27"
                     : "=s"(a), "=s"(b), "=s"(c), "=s"(d), "=s"(e), "=s"(f), "=s"(g), "=s"(h) : "s"(r)
                     : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "memory");
        acc ^= a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void __launch_bounds__(256) calib_write16(uint4* p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ void __launch_bounds__(256) calib_write4(uint32_t* p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (uint32_t)i;
}
__global__ void __launch_bounds__(256) calib_write1(uint8_t* p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = (uint8_t)i;
}

int main()
{
    void* buf = nullptr; uint32_t* sink = nullptr;
    if (hipMalloc(&buf, BYTES) != hipSuccess || hipMalloc((void**)&sink, 4) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
    hipMemset(buf, 0, BYTES); hipDeviceSynchronize();
    const int grid = 256 * 8;
    hipLaunchKernelGGL(calib_read16, dim3(grid), dim3(256), 0, 0, (const uint4*)buf, BYTES / 16, sink);
    hipLaunchKernelGGL(calib_read4, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, BYTES / 4, sink);
    hipLaunchKernelGGL(calib_read_scalar, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, BYTES / 4 / ((size_t)grid * 4), sink);
    hipLaunchKernelGGL(calib_write16, dim3(grid), dim3(256), 0, 0, (uint4*)buf, BYTES / 16);
    hipLaunchKernelGGL(calib_write4, dim3(grid), dim3(256), 0, 0, (uint32_t*)buf, BYTES / 4);
    hipLaunchKernelGGL(calib_write1, dim3(grid), dim3(256), 0, 0, (uint8_t*)buf, BYTES / 4);         // a quarter: byte stores are slow
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    printf("{\"bytes\": %zu, \"bytes_write1\": %zu}\n", BYTES, BYTES / 4);
    return 0;
}
