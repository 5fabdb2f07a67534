// k_11n.hip -- 802.11n 2x2, stage level: the bricks of the reference's 11n receive graph that exist here
// so far, as batched per-stage entry points with the brick port shapes (the whole-path graph is not built yet):
//   sora_hip_demap11n         T11nDemap{BPSK,QPSK,QAM16,QAM64}        kernel/bb/Brick11/src/demapper11n.hpp:89-309 (dsp_demap.h)
//   sora_hip_deinterleave11n  T11nDeinterleave{...}_S0 / _S1           kernel/bb/Brick11/src/deinterleaver_11n.hpp:4-1618
// Both are pure gathers: one coalesced read of a symbol, table look-ups out of LDS, one coalesced write -- HBM-bound.
// The soft-value tables of dsp_demap.h are step functions of the limited coordinate v in [-128,127]; they are regenerated
// from their run lengths (value, count from v = -128 upward).  The de-interleavers are the standard HT interleaver
// (N_COL 13, N_ROW 4 N_BPSC, N_ROT 11) inverted, computed per element instead of the reference's unrolled tables.
#include "kernels.h"
#include "../../include/sora_hip.h"

namespace sora {
namespace {
struct Run { uint8_t v, n; };
__constant__ Run kRuns[83] = {
    {0,97},{1,10},{2,10},{3,11},{4,11},{5,10},{6,10},{7,97},                                                              // [0,8)   BPSK / QPSK
    {0,113},{1,7},{2,4},{3,4},{4,5},{5,4},{6,7},{7,112},                                                                   // [8,16)  16-QAM bit 0
    {0,58},{1,3},{2,2},{3,2},{4,2},{5,3},{6,3},{7,111},{6,3},{5,3},{4,2},{3,2},{2,2},{1,3},{0,57},                         // [16,31) 16-QAM bit 1
    {0,122},{1,3},{2,2},{3,1},{4,2},{5,2},{6,3},{7,121},                                                                   // [31,39) 64-QAM bit 0
    {0,52},{1,3},{2,2},{3,2},{4,1},{5,2},{6,3},{7,127},{6,3},{5,2},{4,1},{3,2},{2,2},{1,3},{0,51},                         // [39,54) 64-QAM bit 1
    {0,18},{1,2},{2,2},{3,2},{4,2},{5,1},{6,3},{7,57},{6,3},{5,2},{4,2},{3,1},{2,2},{1,3},{0,57},
    {1,3},{2,2},{3,1},{4,2},{5,2},{6,3},{7,57},{6,3},{5,1},{4,2},{3,2},{2,2},{1,2},{0,17} };                               // [54,83) 64-QAM bit 2
__constant__ int kRunFirst[7] = { 0, 8, 16, 31, 39, 54, 83 };

__device__ __forceinline__ int data_bin(int l)               // carrier walk of the 11n demappers: -28..-1 then 1..28, pilots at +-7, +-21 skipped
{
    if (l < 26) return l < 7 ? 36 + l : (l < 20 ? 37 + l : 38 + l);
    const int m = l - 26;
    return m < 6 ? 1 + m : (m < 19 ? 2 + m : 3 + m);
}
}  // namespace

// one wave per symbol, four symbols per block; lane l < 52 = data carrier l
__global__ void __launch_bounds__(256) k_demap11n_batch(const uint32_t* in, uint8_t* soft, int nb, uint32_t n)
{
    __shared__ uint8_t s_lut[6][256];
    {   // the six step tables, index v + 128
        const int t = threadIdx.x;
        for (int w = 0; w < 6; w++) {
            int acc = 0; uint8_t val = 0;
            for (int r = kRunFirst[w]; r < kRunFirst[w + 1]; r++) { if (t >= acc && t < acc + kRuns[r].n) val = kRuns[r].v; acc += kRuns[r].n; }
            s_lut[w][t] = val;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t sym = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sym >= n || lane >= 52) return;
    const cpx x = unpack(in[(size_t)sym * 64 + data_bin(lane)]);
    const int re = min(max(x.re, -128), 127) + 128, im = min(max(x.im, -128), 127) + 128;      // demap_limit
    uint8_t* o = soft + ((size_t)sym * 52 + lane) * nb;
    switch (nb) {
    case 1: o[0] = s_lut[0][re]; break;
    case 2: o[0] = s_lut[0][re]; o[1] = s_lut[0][im]; break;
    case 4: o[0] = s_lut[1][re]; o[1] = s_lut[2][re]; o[2] = s_lut[1][im]; o[3] = s_lut[2][im]; break;
    default: o[0] = s_lut[3][re]; o[1] = s_lut[4][re]; o[2] = s_lut[5][re]; o[3] = s_lut[3][im]; o[4] = s_lut[4][im]; o[5] = s_lut[5][im];
    }
}

// one thread per output soft value
__global__ void __launch_bounds__(256) k_deint11n_batch(const uint8_t* in, uint8_t* out, int nb, int iss, uint32_t n)
{
    const uint32_t per = 52u * nb;
    const uint64_t g = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (uint64_t)n * per) return;
    const uint32_t sym = (uint32_t)(g / per); const int k = (int)(g - (uint64_t)sym * per);
    const int s = nb / 2 > 1 ? nb / 2 : 1, nrow = 4 * nb, np = (int)per;
    const int i = nrow * (k % 13) + k / 13;
    int j = s * (i / s) + (i + np - (13 * i) / np) % s;
    if (iss > 0) j = ((j - ((iss * 2) % 3 + 3 * (iss / 3)) * 11 * nb) % np + np) % np;
    out[g] = in[(uint64_t)sym * per + j];
}

}  // namespace sora

using namespace sora;

int sora_hip_demap11n(const sora_complex16* d_in, uint8_t* d_soft, int n_bpsc, size_t n, void* stream)
{
    if (sora_hip_device_count() <= 0) return sora_internal_fail(SORA_ERR_NO_DEVICE, "no HIP device: this library has no CPU path", 0);
    if (!d_in || !d_soft || !(n_bpsc == 1 || n_bpsc == 2 || n_bpsc == 4 || n_bpsc == 6)) return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_hip_demap11n: bad argument", 0);
    if (n == 0) return SORA_OK;
    hipLaunchKernelGGL(k_demap11n_batch, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint32_t*>(d_in), d_soft, n_bpsc, (uint32_t)n);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SORA_OK : sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "k_demap11n_batch", (int)e);
}

int sora_hip_deinterleave11n(const uint8_t* d_in, uint8_t* d_out, int n_bpsc, int spatial_stream, size_t n, void* stream)
{
    if (sora_hip_device_count() <= 0) return sora_internal_fail(SORA_ERR_NO_DEVICE, "no HIP device: this library has no CPU path", 0);
    if (!d_in || !d_out || !(n_bpsc == 1 || n_bpsc == 2 || n_bpsc == 4 || n_bpsc == 6) || spatial_stream < 0 || spatial_stream > 1)
        return sora_internal_fail(SORA_ERR_INVALID_PARAM, "sora_hip_deinterleave11n: bad argument", 0);
    if (n == 0) return SORA_OK;
    const uint64_t total = (uint64_t)n * 52 * n_bpsc;
    hipLaunchKernelGGL(k_deint11n_batch, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, d_in, d_out, n_bpsc, spatial_stream, (uint32_t)n);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? SORA_OK : sora_internal_fail(SORA_ERR_HARDWARE_FAILED, "k_deint11n_batch", (int)e);
}
