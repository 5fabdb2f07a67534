"""Hosts other than the Python binding, EXECUTED on the GPU box (-m gpu): the plain-C example hosts (the reference's
harness language, `demod11 -d`) and a BRICK-shaped C++ graph built from include/sora_brick.hpp.  They link
libsora_hip.so like any user would (gcc / g++, -lsora_hip) and run as separate processes."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "examples", "_build")


def build(cc, std, src, name, extra=()):
    import sora_amd
    sora_amd.load()
    if sora_amd.device_count() <= 0:
        pytest.skip("no HIP device")
    os.makedirs(OUT, exist_ok=True)
    libdir = os.path.dirname(sora_amd.lib_path())
    exe = os.path.join(OUT, name)
    r = subprocess.run([cc, std, "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-L", libdir, "-lsora_hip",
                        "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-o", exe] + list(extra),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def run(cmd):
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    return r.stdout


def make_dump(iq, raw14):
    """int16 [n,2] -> Sora dump bytes: 128-byte RX_BLOCKs = 16-byte descriptor + 28 COMPLEX16 (brickutil.h:40-55)."""
    n = len(iq) // 28 * 28
    x = iq[:n].astype(np.int16)
    if raw14:
        x = ((x.astype(np.int32) >> 2) & 0x3FFF).astype(np.uint16).view(np.int16)
    blocks = np.zeros((n // 28, 64), np.int16)
    blocks[:, 0] = 1                                                    # descriptor: valid flag, rest unused by the loader
    blocks[:, 8:] = x.reshape(-1, 56)
    return blocks.tobytes()


def test_plain_c_host_decodes_the_recorded_dump(tmp_path, golden_dir):
    """examples/demod11a.c (the reference's `demod11 --802.11a.brick -d -f fsample-6.dmp -p 40`) on kernel/test-data's
    fsample-6 re-framed as the dump it came from: 6 Mbps, 1392 bytes, FCS 0x80ef9b11, MPDU sha256 5a13a477..."""
    exe = build("gcc", "-std=c11", os.path.join(ROOT, "examples", "demod11a.c"), "demod11a")
    iq = np.load(os.path.join(golden_dir, "fsample6_40mhz_i8.npz"))["iq_i8"].astype(np.int16) << 8
    dump = tmp_path / "fsample6.dmp"; dump.write_bytes(make_dump(iq, raw14=True))
    mp = tmp_path / "mpdu.bin"
    out = run([exe, str(dump), "--raw14", "--rate", "40", "--out", str(mp)])
    assert "6000 kbps" in out and "length 1392" in out and "FCS 80ef9b11" in out and "FRAME_OK" in out and "good 1 / bad 0" in out, out
    assert hashlib.sha256(mp.read_bytes()).hexdigest() == "5a13a47743867e307040a009e1172b916c9015cd34fac586cafb2d0f1fd64b62"


def test_plain_c_host_keeps_calls_in_flight_and_takes_completions_as_they_happen(tmp_path, golden_dir):
    """examples/rxthread11a.c: RxThread's loop per call -- process_dev -> deliver_async into page-locked buffers -> (depth calls in flight) wait_any -> compare -- compiled
    with gcc -std=c11 -Wall -Werror against include/sora_hip.h only.  30 calls of fsample-6 with 5 in flight: every delivered table equals the first call's, whose frame is the
    recorded one; with nothing left in flight sora_rx_wait_any refuses instead of blocking."""
    exe = build("gcc", "-std=c11", os.path.join(ROOT, "examples", "rxthread11a.c"), "rxthread11a")
    iq = np.load(os.path.join(golden_dir, "fsample6_40mhz_i8.npz"))["iq_i8"].astype(np.int16) << 8
    dump = tmp_path / "fsample6.dmp"; dump.write_bytes(make_dump(iq, raw14=True))
    for depth, calls in ((5, 30), (1, 3), (16, 40)):
        out = run([exe, str(dump), "--raw14", "--calls", str(calls), "--depth", str(depth)])
        assert "6000 kbps" in out and "length 1392" in out and "FCS 80ef9b11" in out and "FRAME_OK" in out, out
        assert "calls %d, in flight %d, collected %d, tables differing from the first call's 0, idle wait_any refused" % (calls, depth, calls) in out, out


def test_plain_c_hosts_of_the_11b_and_11n_graphs_run(tmp_path, golden_dir, oracle):
    """examples/demod11b.c / demod11n.c on recorded modulator output (tests/golden): every frame the oracle reports."""
    from test_oracle_11b import channel_11b
    from gpu_util import capture_11n
    z = np.load(os.path.join(golden_dir, "refgraph_11b.npz"))
    c = channel_11b(z["tx_1"], 101)
    want = oracle.rx11b_capture(c)
    f = tmp_path / "b.dmp"; f.write_bytes(make_dump(c, raw14=False))
    exe = build("gcc", "-std=c11", os.path.join(ROOT, "examples", "demod11b.c"), "demod11b")
    out = run([exe, str(f)])
    assert out.count("FRAME_OK") == sum(1 for r in want if r["error_code"] == 1) >= 1, out
    zn = np.load(os.path.join(golden_dir, "refgraph_11n.npz"))
    a, b = capture_11n(np.random.default_rng(3), [(zn["tx1_0"], zn["tx1_1"])], sigma=20.0)
    wn = oracle.rx11n_capture(a, b)
    (tmp_path / "n_0.dmp").write_bytes(make_dump(a, raw14=False)); (tmp_path / "n_1.dmp").write_bytes(make_dump(b, raw14=False))
    exe = build("gcc", "-std=c11", os.path.join(ROOT, "examples", "demod11n.c"), "demod11n")
    out = run([exe, str(tmp_path / "n")])
    assert out.count("FRAME_OK") == sum(1 for r in wn if r["error_code"] == 1) >= 1, out


@pytest.mark.parametrize("nb", [1, 2, 4, 6])
def test_brick_adapter_chain_runs_on_the_gpu(tmp_path, oracle, nb):
    """THipFFT64 -> THip11aDemap<N> -> THip11aDeinterleave<N> -> sink, built and driven like a CREATE_BRICK_* chain
    (tests/cxx/brick_chain.cpp): what comes out of the sink is deinterleave(demap(FFT<64>(x))) of the oracle, bit for bit."""
    exe = build("g++", "-std=c++17", os.path.join(ROOT, "tests", "cxx", "brick_chain.cpp"), "brick_chain")
    rng = np.random.default_rng(40 + nb)
    n = 64
    x = rng.integers(-9000, 9000, size=(n, 64, 2)).astype(np.int16)
    fin = tmp_path / "in.bin"; fout = tmp_path / "out.bin"
    fin.write_bytes(x.tobytes())
    out = run([exe, str(nb), str(fin), str(fout)])
    assert "%d symbols" % n in out
    got = np.frombuffer(fout.read_bytes(), np.uint8).reshape(n, 48 * nb)
    for i in range(n):
        want = oracle.deinterleave(nb, oracle.demap(nb, oracle.fft(x[i], 64)))
        assert np.array_equal(got[i], want), i


@pytest.mark.parametrize("rate", [6000, 24000, 48000, 54000])
def test_brick_graph_decodes_a_whole_frame_on_the_gpu(tmp_path, oracle, rate):
    """tests/cxx/frame_chain.cpp: T11aLTS, then per symbol T11aDataSymbol..TChannelEqualization -> TPilotTrack -> T11aDemap<N> ->
    T11aDeinterleave<N> -> T11aViterbi as BRICK adapters (one symbol per Process(), like the reference's bricks), with the timing and RX
    vector the oracle's receiver found.  What the Viterbi brick emits descrambles to the transmitted MPDU with a good FCS."""
    from gpu_util import awgn
    from oracle.pyoracle import rate_params
    exe = build("g++", "-std=c++17", os.path.join(ROOT, "tests", "cxx", "frame_chain.cpp"), "frame_chain")
    rng = np.random.default_rng(rate)
    mp = rng.integers(0, 256, 333).astype(np.uint8).tobytes()
    cap = awgn(oracle.tx_capture(mp, rate, lead=40), 200, rate)[::2].copy()
    res = oracle.rx_capture(cap, 20)
    assert len(res) == 1 and res[0]["error_code"] == 1
    nb, cr, _ = rate_params(rate)
    fin = tmp_path / "cap.bin"; fout = tmp_path / "dec.bin"; fin.write_bytes(cap.tobytes())
    out = run([exe, str(fin), str(res[0]["start_sample"]), str(res[0]["nsym"]), str(nb), str(cr), str(res[0]["length"]), str(fout)])
    assert "%d symbols" % res[0]["nsym"] in out
    dec = np.frombuffer(fout.read_bytes(), np.uint8)
    e, mpdu, crc = oracle.desc_sink(dec, res[0]["length"])
    assert e == 1 and mpdu.tobytes()[:len(mp)] == mp and crc == res[0]["crc32"]


def test_c_host_shards_captures_and_gathers_over_rccl(tmp_path, golden_dir):
    """examples/shard11a.c: the C-level multi-GPU entry points (sora_shard_*: partition, ncclAllGather of the result rows over
    RCCL).  This box has one GPU, so the world is one rank -- the collectives run all the same; the 8-GPU run is the same
    command once per rank."""
    exe = build("gcc", "-std=c11", os.path.join(ROOT, "examples", "shard11a.c"), "shard11a")
    iq = np.load(os.path.join(golden_dir, "fsample6_40mhz_i8.npz"))["iq_i8"].astype(np.int16) << 8
    dump = tmp_path / "fsample6.dmp"; dump.write_bytes(make_dump(iq, raw14=True))
    out = run([exe, str(dump), "--raw14", "--rate", "40", "--captures", "5", "--world", "1", "--rank", "0", "--id-file", str(tmp_path / "sora.id")])
    assert "world 1: 5 captures, 5 frames gathered (5), good 5 / bad 0" in out and "6000 kbps length 1392 FCS 80ef9b11, last: capture 4" in out, out
    # the gathered MPDUs: five copies of the fixture's MPDU (sha256 pinned in SURVEY.md 8c), byte for byte -- the host prints their FNV-1a
    from oracle.pyoracle import Oracle
    mp = Oracle().rx_capture(iq[:len(iq) // 28 * 28], 40)[0]["mpdu"]
    assert hashlib.sha256(mp).hexdigest() == "5a13a47743867e307040a009e1172b916c9015cd34fac586cafb2d0f1fd64b62"
    h = 2166136261
    for b in mp * 5:
        h = ((h ^ b) * 16777619) & 0xFFFFFFFF
    assert "%d MPDU bytes gathered, fnv1a %08x" % (5 * len(mp), h) in out, out


def test_shard_api_from_python_matches_results(golden_dir):
    """sora_shard_gather_results through the binding (world of one): the gathered table is sora_rx_results' table."""
    import ctypes
    import torch
    import sora_amd
    from sora_amd.capi import FrameResult
    L = sora_amd.load()
    iq = (np.load(os.path.join(golden_dir, "fsample6_40mhz_i8.npz"))["iq_i8"].astype(np.int16) << 8)
    n = len(iq) // 28 * 28
    rx = sora_amd.Rx(3, 3 * n, sample_rate_mhz=40, max_frames_per_capture=4)
    d = torch.from_numpy(np.concatenate([iq[:n]] * 3)).cuda()
    rx.process_dev(d, [(0, n, 10), (n, n, 11), (2 * n, n, 12)])
    want = rx.results(with_mpdu=False)
    ident = (ctypes.c_uint8 * 128)(); assert L.sora_shard_unique_id(ident) == 0
    sh = ctypes.c_void_p(); assert L.sora_shard_create(ident, 1, 0, 0, ctypes.byref(sh)) == 0, L.sora_hip_last_error()
    rows = (FrameResult * 12)(); counts = (ctypes.c_uint32 * 1)(); total = ctypes.c_size_t(0)
    assert L.sora_shard_gather_results(sh, rx._h, 0, 12, rows, counts, ctypes.byref(total)) == 0, L.sora_hip_last_error()
    assert total.value == len(want) == 3 and counts[0] == 3
    for r, w in zip(rows, want):
        assert (r.capture_id, r.end_sample, r.error_code, r.rate_kbps, r.length, r.crc32) == (w["capture_id"], w["end_sample"], w["error_code"], w["rate_kbps"], w["length"], w["crc32"])
    # ... and with the MPDUs: dense, in row order, mpdu_offset pointing into the gathered buffer
    full = rx.results()
    rows2 = (FrameResult * 12)(); mp = (ctypes.c_uint8 * (12 * 2504))(); mtot = ctypes.c_size_t(0)
    assert L.sora_shard_gather_results_mpdu(sh, rx._h, 0, 12, rows2, counts, ctypes.byref(total), 12 * 2504, mp, ctypes.byref(mtot)) == 0, L.sora_hip_last_error()
    assert total.value == 3 and mtot.value == sum(len(w["mpdu"]) for w in full)
    for r, w in zip(rows2, full):
        assert bytes(mp[r.mpdu_offset:r.mpdu_offset + r.length]) == w["mpdu"]
    # a buffer too small for this rank's MPDUs is refused (after the exchange, on every rank)
    assert L.sora_shard_gather_results_mpdu(sh, rx._h, 0, 12, rows2, counts, ctypes.byref(total), 64, mp, ctypes.byref(mtot)) != 0
    first = ctypes.c_size_t(); cnt = ctypes.c_size_t(); got = []
    for r in range(3):
        L.sora_shard_partition(256, 3, r, ctypes.byref(first), ctypes.byref(cnt)); got.append((first.value, cnt.value))
    assert got == [(0, 86), (86, 85), (171, 85)]
    L.sora_shard_destroy(sh); rx.close()


def test_bench_result_exchange_over_rccl_with_a_world_of_one(oracle):
    """bench.py --gpus N (BASELINE configs[4]: 256 captures over 8 GPUs = 32 per rank) ends with ONE exchange: RCCL all-gathers of the
    device-packed rows and the dense MPDU blocks of every rank.  The 8-GPU run is the driver's; this runs the very same function on the
    one GPU of the test box, over an initialised `nccl` process group of one rank, with the per-rank batch of configs[4]."""
    import torch
    import torch.distributed as dist
    import sora_amd
    sys.path.insert(0, ROOT)
    import bench
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        nfr = 32
        iq, descs, payloads = bench.make_workload(oracle, nfr, seed0=0)
        dev = torch.device("cuda", 0)
        d_iq = torch.from_numpy(iq).to(dev)
        rx = sora_amd.Rx(max_captures=nfr, max_total_samples=len(iq), sample_rate_mhz=20, max_frames_per_capture=2)
        g = bench.exchange_results(torch, rx, d_iq, sora_amd.Rx.captures(descs), dev, nfr, 2)
        assert g["rows"] == nfr and g["rows_per_rank"] == [nfr], g
        assert g["mpdus_equal_to_the_transmitted_payloads"] >= nfr - 2 and g["mpdu_bytes"] == bench.MPDU_LEN * nfr, g
        rx.close()
    finally:
        dist.destroy_process_group()
