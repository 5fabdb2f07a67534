"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/sora_hip.h
declares, keeps the struct layouts, and FAILS LOUDLY (never falls back) when no HIP device is present."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sora_hip.h")


@pytest.fixture(scope="module")
def lib():
    import sora_amd
    return sora_amd.load()


def declared_functions():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sora_[a-z0-9_]+)\s*\(", txt)) - {"sora_rx_t"})


def test_library_exports_every_declared_symbol(lib):
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libsora_hip.so does not export %s" % n


def test_binding_covers_the_header():
    from sora_amd import capi
    assert sorted(capi.EXPORTS) == declared_functions()


def test_abi_version_and_struct_layout(lib):
    from sora_amd import capi
    assert lib.sora_hip_abi_version() == 4
    assert ctypes.sizeof(capi.FrameResult) == 36          # 9 x 32-bit words: the unit of the multi-GPU gather
    assert ctypes.sizeof(capi.CaptureDesc) == 16
    assert ctypes.sizeof(capi.RxCfg) == 32


def test_no_silent_cpu_fallback(lib):
    """Without a GPU every compute entry point must refuse (SORA_ERR_NO_DEVICE), not compute on the host."""
    import sora_amd
    if sora_amd.device_count() > 0:
        pytest.skip("a HIP device is present")
    with pytest.raises(sora_amd.SoraError) as e:
        sora_amd.Rx(1, 2800)
    assert e.value.code == -5
    assert lib.sora_hip_fft64(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, None) == -5
    assert lib.sora_hip_demap11a(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 1, None) == -5
    assert b"no HIP device" in lib.sora_hip_last_error()


def test_product_does_not_link_or_import_the_oracle():
    """The shipped library and package must not reach into oracle/ (test infrastructure only)."""
    import sora_amd
    out = subprocess.run(["ldd", sora_amd.lib_path()], capture_output=True, text=True).stdout
    assert "sora_oracle" not in out and "sora_ref" not in out
    pkg = os.path.join(ROOT, "sora_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "pyoracle" not in src and "so_oracle" not in src and "libsora_oracle" not in src, f


def test_c_host_example_compiles():
    """Plain-C hosts (the reference's harness language: `demod11 -d` for 11a, 11b and 11n) build against the header and link the library."""
    import sora_amd
    out = os.path.join(ROOT, "examples", "_build")
    os.makedirs(out, exist_ok=True)
    libdir = os.path.dirname(sora_amd.lib_path())
    for name in ("demod11a", "demod11b", "demod11n", "shard11a"):
        src = os.path.join(ROOT, "examples", name + ".c")
        r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-L", libdir,
                            "-lsora_hip", "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined", "-o", os.path.join(out, name)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_brick_adapter_header_compiles(tmp_path):
    """The BRICK-shaped adapters instantiate into a sink-first graph exactly like CREATE_BRICK_* chains do."""
    hdr = os.path.join(ROOT, "include", "sora_brick.hpp")
    src = tmp_path / "graph.cpp"
    src.write_text("""
#include "sora_brick.hpp"
using namespace sora_brick;
int build_graph(sora_complex16* d_in, sora_complex16* d_fft, uint8_t* d_soft, uint8_t* d_de) {
    CF_Error ctx;
    TDrop<CF_Error> drop(ctx);
    THip11aDeinterleave<6, 16, CF_Error, TDrop<CF_Error>> deint(ctx, &drop, d_de);
    THip11aDemap<6, 16, CF_Error, decltype(deint)> demap(ctx, &deint, d_soft);
    THipFFT64<16, CF_Error, decltype(demap)> fft(ctx, &demap, d_fft);
    DevicePin<sora_complex16, 64 * 16> src(d_in);
    src.append();
    bool ok = fft.Process(src);
    fft.Reset(); fft.Flush();
    // TDownSample44_40 in front of a graph: 44 MHz samples in, 40 MHz out
    THipResample<28 * 40, SORA_INGEST_44TO40, CF_Error, TDrop<CF_Error>> down(ctx, &drop, d_fft);
    DevicePin<sora_complex16, 28 * 40> raw(d_in);
    raw.append();
    ok = ok && down.Process(raw) && down.produced() <= 28 * 40;
    // the 802.11b graph as a source over a batch of 44 MHz captures
    sora_rx_cfg cfg{}; cfg.struct_size = sizeof(cfg); cfg.sample_rate_mhz = 44; cfg.max_captures = 1; cfg.max_total_samples = 2800; cfg.max_frames_per_capture = 4;
    THipRx11bSource rx11b(ctx, cfg);
    sora_capture_desc cap{0, 2800, 0};
    rx11b.Bind(d_in, &cap, 1);
    ok = ok && (rx11b.handle() == nullptr || rx11b.Process());
    // the 802.11n 2x2 graph over two-chain 40 MHz captures
    cfg.sample_rate_mhz = 40;
    THipRx11nSource rx11n(ctx, cfg);
    rx11n.Bind(d_in, d_fft, &cap, 1);
    ok = ok && (rx11n.handle() == nullptr || rx11n.Process());
    // the 802.11n stage bricks and FFT<128>
    THip11nDeinterleave<4, 1, 8, CF_Error, TDrop<CF_Error>> d11n(ctx, &drop, d_de);
    THip11nDemap<4, 8, CF_Error, decltype(d11n)> m11n(ctx, &d11n, d_soft);
    THip11nMimoComp<4, CF_Error, TDrop<CF_Error>> zf(ctx, &drop, d_in, d_fft);
    THipFFT128<8, CF_Error, TDrop<CF_Error>> f128(ctx, &drop, d_fft);
    DevicePin<sora_complex16, 64 * 8> p64(d_in); p64.append(); ok = ok && m11n.Process(p64);
    DevicePin<sora_complex16, 128 * 4> p2(d_in); p2.append(); ok = ok && zf.Process(p2);
    DevicePin<sora_complex16, 128 * 8> p128(d_in); p128.append(); ok = ok && f128.Process(p128);
    return ok ? 0 : (int)ctx.error_code;
}
""")
    assert os.path.exists(hdr)
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for chain in ("brick_chain.cpp", "frame_chain.cpp"):                   # the graphs tests/test_gpu_hosts.py executes on the GPU box
        r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cxx", chain)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_every_bound_function_declares_its_argument_types():
    """ctypes passes undeclared pointer arguments as C ints (truncated to 32 bits): every export the binding lists has argtypes."""
    import sora_amd
    from sora_amd import capi
    L = sora_amd.load(build_if_missing=False)
    no_args = {"sora_hip_abi_version", "sora_hip_last_error", "sora_hip_device_count"}
    missing = [n for n in capi.EXPORTS if n not in no_args and getattr(L, n).argtypes is None]
    assert missing == ["sora_hip_memcpy_d2d"] or missing == [], missing      # (sora_hip_memcpy_d2d is for C hosts; the binding never calls it)


def test_the_library_on_disk_is_built_from_the_sources_in_the_tree():
    """The .so is a git-ignored artefact that travels with the snapshot: build() stamps it with a hash of every source and header, and build_info()
    (reported in bench.py's `build` object) says whether the file that gets loaded was built from the tree as it is now."""
    from sora_amd import build
    build.build()
    info = build.build_info()
    assert info["exists"] and info["stamp"] and info["built_from_this_tree"], info
    assert info["stamp"]["sources_sha256"] == build.sources_sha256()
