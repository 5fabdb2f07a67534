// ref_graph_hip_shim.cpp -- TEST INFRASTRUCTURE ONLY (not shipped, not linked into libsora_hip.so).
//
// VERDICT r4 "missing" #1: a HIP brick inside the reference's OWN graph.  Three bricks written against the reference's real brick protocol -- TFilter<TFILTER_PARAMS>,
// DEFINE_IPORT / DEFINE_OPORT, STD_TFILTER_CONSTRUCTOR, BOOL_FUNC_PROCESS, REFERENCE_LOCAL_CONTEXT (kernel/brick/inc/brick.h:151-475), the deduced pin queues of
// pinqueue.h:104-246 -- whose Process() hands the burst to the product's C ABI (sora_hip_fft64, sora_hip_demap11a, sora_hip_deinterleave11a) and passes the result on
// with Next()->Process(opin()); and copies of CreateDemodGraph11a_40M (fb11ademod_config.hpp:168-233) that instantiate them through CREATE_BRICK_FILTER
// (brick.h:416-420) in place of TFFT64 (Brick11/src/fft.hpp:108-135), T11aDemap<N>::Filter (demapper11a.hpp:10-79) and T11aDeinterleave* (deinterleaver.hpp).  Those
// copies are GENERATED from the reference's text by oracle/ref_flatten.py ("hip" mode: fb11ademod_config_hip.hpp in the scratch tree) -- the substitution list there
// is the diff a maintainer applies.  The RxThread loop around the graph is ref_graph_shim.cpp's, unchanged.
//
// The product's entry points arrive as function pointers (ref_hip_bind) so that this library links nothing of the product: tests/test_gpu_hip_bricks_in_reference_graph.py
// binds the loaded libsora_hip.so and compares the events of this graph with libsora_refgraph.so's, capture by capture.
#include "MACStopwatch.h"
#include "stdbrick.hpp"
#include "fb11ademod_config.hpp"

struct HipStageApi {
    int (*fft64)(const void* d_in, void* d_out, size_t n, void* stream);
    int (*demap11a)(const void* d_in, void* d_soft, int n_bpsc, size_t n, void* stream);
    int (*deinterleave11a)(const void* d_in, void* d_out, int n_bpsc, size_t n, void* stream);
    void* (*dmalloc)(size_t);
    int (*h2d)(void*, const void*, size_t);
    int (*d2h)(void*, const void*, size_t);
    void* d_in; void* d_out;
    unsigned calls[3], errors;
};
static HipStageApi g_hip;

// burst in host memory -> device -> stage -> device -> host (the copies are synchronous with the null stream the stage runs on)
template <class CALL> static void hip_stage(int which, const void* in, size_t in_bytes, void* out, size_t out_bytes, CALL call)
{
    g_hip.calls[which]++;
    if (g_hip.h2d(g_hip.d_in, in, in_bytes) != 0 || call(g_hip.d_in, g_hip.d_out) != 0 || g_hip.d2h(out, g_hip.d_out, out_bytes) != 0) g_hip.errors++;
}

// ---- TFFT64's place (Brick11/src/fft.hpp:108-135)
DEFINE_LOCAL_CONTEXT(THipFFT64, CF_VOID);
template<TFILTER_ARGS>
class THipFFT64 : public TFilter<TFILTER_PARAMS>
{
public:
    DEFINE_IPORT(COMPLEX16, 64);
    DEFINE_OPORT(COMPLEX16, 64);
public:
    REFERENCE_LOCAL_CONTEXT(THipFFT64);
    STD_TFILTER_CONSTRUCTOR(THipFFT64) { }
    STD_TFILTER_RESET() { }
    BOOL_FUNC_PROCESS(ipin)
    {
        while (ipin.check_read())
        {
            const COMPLEX16* in = ipin.peek();
            COMPLEX16* out = opin().append();
            hip_stage(0, in, 256, out, 256, [](const void* di, void* d_o) { return g_hip.fft64(di, d_o, 1, NULL); });
            ipin.pop();
            Next()->Process(opin());
        }
        return true;
    }
};

// ---- T11aDemap<N_BPSC>::Filter's place (demapper11a.hpp:10-79)
DEFINE_LOCAL_CONTEXT(THip11aDemap, CF_VOID);
template<ushort N_BPSC>
class THip11aDemap
{
public:
template<TFILTER_ARGS>
class Filter : public TFilter<TFILTER_PARAMS>
{
    static const int NbPS = N_BPSC * 48;
public:
    DEFINE_IPORT(COMPLEX16, 64);
    DEFINE_OPORT(uchar, NbPS);
public:
    REFERENCE_LOCAL_CONTEXT(THip11aDemap);
    STD_TFILTER_CONSTRUCTOR(Filter) { }
    BOOL_FUNC_PROCESS(ipin)
    {
        while (ipin.check_read())
        {
            const COMPLEX16* in = ipin.peek();
            uchar* out = opin().append();
            hip_stage(1, in, 256, out, NbPS, [](const void* di, void* d_o) { return g_hip.demap11a(di, d_o, N_BPSC, 1, NULL); });
            ipin.pop();
            Next()->Process(opin());
        }
        return true;
    }
}; };

// ---- T11aDeinterleaveBPSK / QPSK / QAM16 / QAM64's place (deinterleaver.hpp)
DEFINE_LOCAL_CONTEXT(THip11aDeinterleave, CF_VOID);
template<ushort N_BPSC>
class THip11aDeinterleave
{
public:
template<TFILTER_ARGS>
class Filter : public TFilter<TFILTER_PARAMS>
{
    static const int NbPS = N_BPSC * 48;
public:
    DEFINE_IPORT(uchar, NbPS);
    DEFINE_OPORT(uchar, NbPS);
public:
    REFERENCE_LOCAL_CONTEXT(THip11aDeinterleave);
    STD_TFILTER_CONSTRUCTOR(Filter) { }
    BOOL_FUNC_PROCESS(ipin)
    {
        while (ipin.check_read())
        {
            const uchar* in = ipin.peek();
            uchar* out = opin().append();
            hip_stage(2, in, NbPS, out, NbPS, [](const void* di, void* d_o) { return g_hip.deinterleave11a(di, d_o, N_BPSC, 1, NULL); });
            ipin.pop();
            Next()->Process(opin());
        }
        return true;
    }
}; };

#include "fb11ademod_config_hip.hpp"         // generated by oracle/ref_flatten.py from the reference's own CreateDemodGraph11a_40M

static int g_hip_graph;                      // 0: the reference's bricks, 1: THipFFT64, 2: + THip11aDemap + THip11aDeinterleave
static inline void CreateDemodGraph11a_40M_Selected(ISource*& src, ISource*& vit, IControlPoint*& cs)
{
    if (g_hip_graph == 2) CreateDemodGraph11a_40M_HipFFTDemapDeint(src, vit, cs);
    else if (g_hip_graph == 1) CreateDemodGraph11a_40M_HipFFT(src, vit, cs);
    else CreateDemodGraph11a_40M(src, vit, cs);
}
#define REF_CREATE_40M CreateDemodGraph11a_40M_Selected
#include "ref_graph_shim.cpp"                // the RxThread loop and the exports (ref_rx11a_capture: a fresh graph per capture)

EXPORT int ref_hip_bind(void* fft64, void* demap11a, void* deinterleave11a, void* dmalloc, void* h2d, void* d2h)
{
    g_hip.fft64 = (int (*)(const void*, void*, size_t, void*))fft64;
    g_hip.demap11a = (int (*)(const void*, void*, int, size_t, void*))demap11a;
    g_hip.deinterleave11a = (int (*)(const void*, void*, int, size_t, void*))deinterleave11a;
    g_hip.dmalloc = (void* (*)(size_t))dmalloc; g_hip.h2d = (int (*)(void*, const void*, size_t))h2d; g_hip.d2h = (int (*)(void*, const void*, size_t))d2h;
    if (!g_hip.d_in) { g_hip.d_in = g_hip.dmalloc(4096); g_hip.d_out = g_hip.dmalloc(4096); }
    return g_hip.d_in && g_hip.d_out ? 0 : -1;
}
EXPORT void ref_hip_select(int graph) { g_hip_graph = graph; if (g_src) { IReferenceCounting::Release(g_src); g_src = NULL; } }
EXPORT void ref_hip_counters(unsigned out[4]) { out[0] = g_hip.calls[0]; out[1] = g_hip.calls[1]; out[2] = g_hip.calls[2]; out[3] = g_hip.errors; }
