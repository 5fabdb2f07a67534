#!/usr/bin/env python3
"""bench.py -- headline benchmark: IQ Msamples/s through the 802.11a 54 Mbps RX PHY on MI355X.

Workload (BASELINE.json configs[2]): 4096 independent 20 MHz captures per GPU, each holding one 54 Mbps
frame with a 1500-byte MPDU (PLCP LENGTH 1500 -> 56 data symbols, 4880 samples) followed by 160 samples of
silence (capture = 5040 samples = 360 source bursts).  Synthetic IQ: fixed-seed payloads through the
restated reference transmitter (oracle/so_tx11a.c), AWGN at ~30/27 dB SNR on 3 of every 4 captures.
A "step" = one sora_rx_process_dev call over the whole batch, inputs resident in HBM.  `value` counts the
4880 frame samples per capture (19.99 Msamples per 4096 frames).

Multi-GPU (--gpus N): captures are the natural shard -- each rank runs its own batch end to end (weak scaling), no
data-path collective; the RCCL all-gather of the result rows and MPDUs and one all-reduce of the frame counters
after the timed region are the only exchanges.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment)
this process is one rank; a plain `python bench.py --gpus N` starts the N ranks itself (one process per GPU,
rendezvous on 127.0.0.1) and rank 0 prints the line.  --shape shard is BASELINE configs[4] literally: 32 captures
of sixteen back-to-back frames per GPU (256 concurrent captures over 8 GPUs) instead of 4096 captures of one frame.

The timed region: the K-step block (--steps) is repeated until at least --min-seconds have passed; every step is a
process call whose dense result rows AND MPDU array are delivered to page-locked host memory behind the kernels, and the
oldest call in flight is waited for and its rows compared with the verified ones.  Before anything is timed, the first
call's rows are compared, capture by capture, with what the compiled reference graph reports (--check 0 = all captures).

Prints ONE JSON line on rank 0 (see the contract in the task description), with `roofline` for the dominant kernel (HIP
events on the library's own streams, one call in flight) and `cpu_baseline` (the reference's SSE graph compiled from its
sources, on all usable host cores; the scalar restatement where that library is absent).
"""
import argparse
import json
import os
import sys
import time

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues per stream-priority level (default 4).  Round 3's bench
# set GPU_MAX_HW_QUEUES=16 before HIP started, because eight pipelines of one priority ran three at a time; since round 4 the library spreads a
# handle's pipelines over the three priority levels (sora_internal_stream_create) and gets a hardware queue per pipeline by itself: the default
# run sets NO environment variable (config.hw_queues = null; profiles/r04_m_stream_priorities.txt).  --hw-queues N still sets it, for A/B runs.


def _early_hw_queues(argv):
    """--hw-queues N, read before the HIP runtime starts: N > 0 sets GPU_MAX_HW_QUEUES (unless the environment already does),
    0 leaves the runtime's default alone (`config.hw_queues` is then null)."""
    for i, a in enumerate(argv):
        if a == "--hw-queues" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--hw-queues="):
            return int(a.split("=", 1)[1])
    return None


_HWQ = _early_hw_queues(sys.argv)
if _HWQ is None:
    _HWQ = 0
if _HWQ > 0:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(_HWQ))

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# the sections live in benchlib/ (round 5); everything tests and tools used to reach as bench.X is re-exported here
from benchlib.common import *  # noqa: E402,F401,F403
from benchlib.common import _traffic_profile  # noqa: E402,F401
from benchlib.cpu import _cpu_worker, _cpu_worker_11b, _cpu_worker_11n, cpu_baseline, host_cores  # noqa: E402,F401
from benchlib.stages import bench_ingest, bench_stages, bench_tx  # noqa: E402,F401
from benchlib.latency import bench_e2e, bench_large_call, bench_latency  # noqa: E402,F401
from benchlib.rows import bench_11b, bench_11n, bench_ht40, bench_shard_shape  # noqa: E402,F401


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n, argv, extra_env=None):
    """`python bench.py --gpus N` without torch.distributed.run: start the N ranks (one process per GPU, the environment torch.distributed.run would set,
    rendezvous on 127.0.0.1), pass rank 0's output through, return the first non-zero exit code."""
    import subprocess
    port = str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        while any(pr.poll() is None for pr in procs):
            time.sleep(0.2)
            rc = next((pr.returncode for pr in procs if pr.returncode not in (None, 0)), 0)
            if rc:                              # a rank that died leaves the others in a rendezvous or a collective: end exactly the processes started here, now
                break
        rc = rc or next((pr.returncode for pr in procs if pr.returncode not in (None, 0)), 0)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    return rc


def launcher_selftest():
    """What tests/test_bench_launcher.py runs on CPU: the ranks launch_ranks started find each other (gloo) and rank 0 prints one line."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([rank + 1], dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"launcher": "ok", "n_gpus": world, "sum_of_rank_plus_one": int(t.item()), "master": os.environ["MASTER_ADDR"] + ":" + os.environ["MASTER_PORT"]}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-mt-worker":                 # one instance of the reference's two-thread harness (cpu_baseline.two_thread): never returns normally
        p_, nf, first, stride, secs = sys.argv[2].split(",")
        print(json.dumps(list(_cpu_worker((p_, int(nf), int(first), int(stride), float(secs), "reference_mt")))), flush=True)
        os._exit(0)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--headline-only", action="store_true", help="stop after the timed region and print its step time only (kernel-trace runs: tools/trace_timeline.sh)")
    ap.add_argument("--wait-oldest", action="store_true", help="the host waits for its OLDEST ticket (round-3 loop) instead of taking completions as they come (sora_rx_wait_any)")
    ap.add_argument("--extra-launches", type=int, default=0, help="experiment: empty kernel launches appended to every call (tool hook sora_internal_rx_extra: needs the TOOLS variant of the library, SORA_HIP_LIB=sora_amd/lib/variants/tools.so from sora_amd.build.build_variant('tools', ['SORA_TOOLS']))")
    ap.add_argument("--no-plain", action="store_true", help="skip the plain_host section (experiments)")
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="captures per GPU (default: the BASELINE config; "
                    "--gpus 8 --frames 32 is BASELINE configs[4] literally: 256 captures over 8 GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the stage / ingest / tx / 11b / 11n sections")
    ap.add_argument("--depth", type=int, default=0, help="process calls in flight on the handle's internal pipelines (0 = library default)")
    ap.add_argument("--trellis", type=int, default=-1, help="trellis kernel: 64 = k_viterbi, 16 = k_viterbi16, 0 = the library's choice from the depth (default: leave the handle as created)")
    ap.add_argument("--check", type=int, default=0, help="captures compared with the reference after the timed region (0 = all)")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="the timed region repeats the K-step block until it has lasted this long")
    ap.add_argument("--no-deliver", action="store_true", help="do not deliver rows + MPDUs to the host inside the timed region (round-1 behaviour)")
    ap.add_argument("--hw-queues", type=int, default=0, help="GPU_MAX_HW_QUEUES for this process (read before HIP starts); 0 = leave the runtime default")
    ap.add_argument("--shape", default="batch", choices=["batch", "shard"], help="batch: --frames captures of one frame per GPU (BASELINE configs[2], the headline); "
                    "shard: 32 captures of 16 back-to-back frames per GPU (BASELINE configs[4]: 256 concurrent captures over 8 GPUs)")
    ap.add_argument("--launcher-selftest", action="store_true", help="only start the ranks, let them find each other (gloo) and print one line: the CPU test of --gpus N's own launcher")
    ap.add_argument("--only", default="", help="run just one of the extra sections (stages, ingest, tx, rx11b, rx11b_cck, rx11n, rx11n_40, shard_32x16, latency) and print its object: for profiling that section alone")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:                               # not under torch.distributed.run: this process becomes the launcher of the N ranks
        if not args.launcher_selftest:
            import torch
            if torch.cuda.device_count() < args.gpus:
                sys.exit("bench.py: --gpus %d, but this node shows %d device(s)" % (args.gpus, torch.cuda.device_count()))
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.launcher_selftest:
        return launcher_selftest()

    import torch
    import sora_amd
    from oracle.pyoracle import Oracle

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE is %d: start it as `python bench.py --gpus N` or under torch.distributed.run --nproc-per-node N with the same N" % (args.gpus, world))
    if torch.cuda.device_count() < min(world, local_rank + 1):
        sys.exit("bench.py: rank %d needs device %d, this node shows %d" % (rank, local_rank, torch.cuda.device_count()))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if args.only:
        sections = {"stages": lambda: bench_stages(torch, sora_amd, dev), "ingest": lambda: bench_ingest(torch, sora_amd, dev), "tx": lambda: bench_tx(torch, sora_amd),
                    "rx11b": lambda: bench_11b(torch, sora_amd, dev), "rx11b_cck": lambda: bench_11b(torch, sora_amd, dev, cpu=False, rate_kbps=11000),
                    "rx11n": lambda: bench_11n(torch, sora_amd, dev),
                    "rx11n_40": lambda: bench_ht40(torch, sora_amd, dev), "shard_32x16": lambda: bench_shard_shape(torch, sora_amd, dev, Oracle()),
                    "latency": lambda: bench_latency(torch, sora_amd, dev, None, None, None, 0)}
        print(json.dumps({args.only: sections[args.only]()}))
        return
    oracle = Oracle()
    shard = args.shape == "shard"
    # captures per rank x frames per capture: 4096 x 1 (configs[2]) | 32 x 16 (configs[4]: k_scan walks a capture's frames one after the other, the other kernels see 512 frames)
    nfr = 32 if shard and args.frames == FRAMES_PER_GPU else args.frames
    fpc = 16 if shard else 1
    MAXF = fpc + 2 if shard else 2
    iq, descs, payloads = make_workload(oracle, nfr * fpc, seed0=rank * 100003)
    if shard:
        descs = [(i * fpc * CAPTURE_SAMPLES, fpc * CAPTURE_SAMPLES, i) for i in range(nfr)]
    d_iq = torch.from_numpy(iq).to(dev)
    # one 80 MB input re-read every step sits in the 256 MiB Infinity Cache.  The timed steps rotate through NCOPIES device
    # copies of the batch at different addresses (same samples, so every call's table can be compared with the verified one): 320 MB of input in
    # play, consecutive calls share no line.
    NCOPIES = 4 if world == 1 else 2
    d_iqs = [d_iq] + [d_iq.clone() for _ in range(NCOPIES - 1)]
    descs = sora_amd.Rx.captures(descs)           # packed sora_capture_desc[]: built once, submitted every step
    rx = sora_amd.Rx(max_captures=nfr, max_total_samples=len(iq), sample_rate_mhz=20, device=local_rank, max_frames_per_capture=MAXF)

    if args.depth:
        rx.set_depth(args.depth)
    depth = rx.set_depth(0)
    if args.trellis >= 0:
        rx.set_trellis(args.trellis)
    if args.extra_launches:
        import ctypes
        _L = sora_amd.capi.load(); _L.sora_internal_rx_extra.argtypes = [ctypes.c_void_p, ctypes.c_uint]
        assert _L.sora_internal_rx_extra(rx._h, args.extra_launches) == 0
    torch.cuda.synchronize()
    rx.wait_for_producer = False                  # the inputs are resident in HBM from here on: no per-call wait for torch's stream

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the first call is checked before anything is timed: every capture against the reference (or --check of them)
    t = rx.process_dev(d_iq, descs)
    res = rx.results(ticket=t)
    kind, want = reference_rows(iq, nfr, oracle, fpc)
    idx = list(range(nfr)) if args.check <= 0 or args.check >= nfr else list(range(0, nfr, max(1, nfr // args.check)))[:args.check]
    parity_ok, why = check_against_reference(res, kind, want, idx)
    if not parity_ok:
        print("PARITY MISMATCH vs %s: %s" % (kind, why), file=sys.stderr)
    n_ok = sum(1 for r in res if r["error_code"] == sora_amd.E_FRAME_OK)
    seen = {}                                                       # (rows come in capture, time order: the k-th row of a capture is its k-th frame)
    n_payload_ok = 0
    for r in res:
        k = seen.get(r["capture_id"], 0); seen[r["capture_id"]] = k + 1
        n_payload_ok += r["error_code"] == sora_amd.E_FRAME_OK and k < fpc and r["mpdu"][:-4] == payloads[r["capture_id"] * fpc + k]

    # ---- result delivery inside the timed region: after every process call its dense rows and its MPDU array are copied
    # to page-locked host memory behind the kernels (sora_rx_deliver_async), and the oldest call in flight is waited for
    # and its rows compared with the verified ones -- what RxThread does per frame (fb11a_demod.cpp:37-71), per call.
    nb = depth + TableChecker.EXTRA
    bufs = [sora_amd.HostResults(nfr * MAXF, rx.mpdu_bytes(t)) for _ in range(nb)]
    rx.deliver_async(t, bufs[0]); rx.wait(t)
    exp_n = int(bufs[0].nrows[0]); exp_rows = bufs[0].rows[:exp_n].copy(); exp_mpdu = bufs[0].mpdu.copy()
    exp_bytes = exp_rows.tobytes()
    host_rows_ok = exp_n == len(res) and all(int(exp_rows[k]["crc32"]) == res[k]["crc32"] and int(exp_rows[k]["error_code"]) == res[k]["error_code"] for k in range(exp_n)) \
        and all(bytes(exp_mpdu[int(r["mpdu_offset"]):int(r["mpdu_offset"]) + int(r["length"])]) == res[k]["mpdu"] for k, r in enumerate(exp_rows) if int(r["error_code"]) == 1)
    stats = {"t_submit": 0.0, "t_wait": 0.0, "t_check": 0.0}
    share = pin_rank_threads(local_rank, world)                     # this rank's cores: the submit thread on the first, the checker's threads on the others
    chk = TableChecker(exp_bytes, exp_mpdu, cores=share[1:] if len(share) > 1 else None)   # rows AND MPDU bytes of every delivered call, compared by a few host threads

    # Completion order (round 4): calls in flight overtake one another (their streams sit on different dispatch priorities and share the chip), so
    # the loop takes whichever delivered call has finished (sora_rx_wait_any) and the next process call reuses THAT pipeline; --wait-oldest is the
    # round-3 loop (always wait for the oldest ticket), reported beside the headline as in_order_host.
    import collections
    inflight = {}                                                   # ticket -> index of the buffer it is delivered into
    free = collections.deque(range(nb))
    order = {"any": not args.wait_oldest}

    def consume_one():
        ta = time.perf_counter()
        if order["any"]:
            tk = rx.wait_any()
        else:
            tk = min(inflight); rx.wait(tk)
        tb = time.perf_counter()
        stats["t_wait"] += tb - ta
        bi = inflight.pop(tk); b = bufs[bi]
        # every call's rows AND MPDU bytes, whatever the number of ranks: the comparison runs on the checker's threads, which (like the submit
        # thread) are pinned to this rank's own share of the host's cores (pin_rank_threads) -- round 3 sampled every world-th call instead
        chk.check(bi, int(b.nrows[0]) == exp_n, b.rows[:exp_n], b.mpdu)
        free.append(bi)
        stats["t_check"] += time.perf_counter() - tb

    def run_block(k, deliver, dep=None, bound=False):
        dep = dep or depth                                          # calls in flight on the handle right now (<= len(bufs) - EXTRA)
        for _ in range(k):
            ta = time.perf_counter()
            if deliver:
                bi = free.popleft()
                chk.release(bi)                                     # the buffer the next call will be delivered into: its comparison must be over
                if bound:
                    rx.bind_mpdu(bufs[bi])                          # sora_rx_bind_mpdu: the call's frame sink writes its MPDUs into this buffer itself; deliver_async copies rows only
            tk = rx.process_dev(d_iqs[(rx.ticket() + 1) % NCOPIES], descs)
            if deliver:
                rx.deliver_async(tk, bufs[bi]); inflight[tk] = bi
            stats["t_submit"] += time.perf_counter() - ta
            if deliver and len(inflight) >= dep:
                consume_one()
        while inflight:
            consume_one()

    deliver = not args.no_deliver
    run_block(args.warmup, deliver)
    rx.flush()
    barrier()
    run_block(args.steps, deliver); rx.flush()                       # (the first block after the warm-up still pays first-use costs: size the region from the second)
    t0 = time.perf_counter()
    run_block(args.steps, deliver); rx.flush()
    probe = time.perf_counter() - t0
    repeats = max(1, int(np.ceil(args.min_seconds / max(probe, 1e-6))))
    if world > 1:                                  # every rank runs the same number of blocks
        r_t = torch.tensor([repeats], device=dev, dtype=torch.int64); dist.all_reduce(r_t, op=dist.ReduceOp.MAX); repeats = int(r_t.item())
    chk.drain(); chk.compared = chk.bad = 0; chk.mpdu_compared = 0
    barrier()
    for k_ in ("t_submit", "t_wait", "t_check"):
        stats[k_] = 0.0
    t0 = time.perf_counter()
    run_block(args.steps * repeats, deliver)                        # ONE continuous run of K x repeats steps: the calls in flight are collected at its end only
    rx.flush()                                                      # (round 3 drained the handle after every K steps: with K = 20 and eight calls in flight a fifth of the region was fill and drain)
    chk.drain()                                                     # every delivered table has been compared when the clock stops
    barrier()
    t1 = time.perf_counter()
    host_ms = {k_[2:]: round(1e3 * stats[k_] / (args.steps * repeats), 4) for k_ in ("t_submit", "t_wait", "t_check")}
    if args.headline_only:
        if rank == 0:
            print(json.dumps({"ms_per_step": round(1e3 * (t1 - t0) / (args.steps * repeats), 4), "steps": args.steps * repeats, "calls_in_flight": depth, "captures_per_call": nfr,
                              "calls_with_wrong_rows": chk.bad, "host_ms_per_step": host_ms}))
        return
    timed_steps = args.steps * repeats
    stats["delivered"], stats["bad"] = chk.compared, chk.bad
    stats["mpdu_compared_timed"] = getattr(chk, "mpdu_compared", 0)
    mpdu_ok = bool(deliver) and all((b.mpdu == exp_mpdu).all() for b in bufs)     # the last calls' MPDU arrays once more, byte for byte
    # the same K steps once more with HIP events around every kernel launch (on the streams the kernels run on): the
    # roofline's launch durations are means over this region; `value` comes from the un-instrumented region above
    # (recording 6 events per call costs a few percent, reported as ms_per_step_profiled)
    rx.set_profiling(True)
    barrier()
    t2 = time.perf_counter()
    run_block(args.steps, deliver)
    rx.flush()
    barrier()
    t3 = time.perf_counter()
    ktimes = rx.kernel_times()
    rx.set_profiling(False)
    # and with ONE call in flight: each kernel alone on the chip (the per-kernel roofline without the CU sharing of overlapped calls).
    # The trellis kernel is pinned to the one the timed region used (left to itself the library picks k_viterbi for a single call in
    # flight and k_viterbi16 from depth 4); the other one is measured alone as well, for the record.
    lanes = rx.trellis(); trellis_setting = rx.set_trellis(-1)
    tname = TRELLIS_NAMES

    def alone(l):
        rx.set_trellis(l); rx.set_depth(1); rx.flush()
        rx.set_profiling(True)
        for _ in range(max(10, args.steps // 2)):
            rx.process_dev(d_iq, descs)
        rx.flush()
        kt = rx.kernel_times()
        rx.set_profiling(False)
        return {(tname[l] if k == "k_viterbi" else k): v for k, v in kt.items()}
    ktimes1 = alone(lanes)
    ktimes1_others = {l: alone(l) for l in TRELLIS_NAMES if l != lanes}
    ktimes = {(tname[lanes] if k == "k_viterbi" else k): v for k, v in ktimes.items()}

    # ---- what a plain host gets: process -> deliver -> wait with ONE or TWO calls in flight, i.e. at most two of the
    # handle's streams in use -- fewer than the runtime's default four hardware queues, so GPU_MAX_HW_QUEUES plays no part -- same timed-region
    # protocol (delivery + comparison inside), each trellis kernel pinned in turn.
    plain = {}
    if world == 1 and not args.no_plain and not shard:
        if order["any"] and deliver:                                 # the headline's calls in flight with the round-3 loop: always wait for the OLDEST ticket
            rx.set_trellis(lanes); rx.set_depth(depth); rx.flush(); order["any"] = False
            run_block(args.warmup, deliver); rx.flush(); chk.drain(); bad0 = chk.bad
            nblk = max(1, repeats // 4)
            tp0 = time.perf_counter()
            run_block(args.steps * nblk, deliver)
            rx.flush(); chk.drain()
            ms_p = (time.perf_counter() - tp0) / (nblk * args.steps) * 1e3
            plain["calls_in_flight_%d_waiting_for_the_oldest_ticket" % depth] = {"ms_per_step": round(ms_p, 4), "msamples_per_s": round(nfr * FRAME_SAMPLES / ms_p / 1e3, 1),
                                                                                 "calls_with_wrong_rows": chk.bad - bad0}
            order["any"] = True
        for dval in (1, 2):
            for l in (64, 16, 1):
                rx.set_trellis(l); rx.set_depth(dval); rx.flush()
                run_block(args.warmup, deliver, dval); rx.flush(); chk.drain(); bad0 = chk.bad
                nblk = max(1, repeats // 6)
                tp0 = time.perf_counter()
                run_block(args.steps * nblk, deliver, dval)
                rx.flush(); chk.drain()
                ms_p = (time.perf_counter() - tp0) / (nblk * args.steps) * 1e3
                plain["calls_in_flight_%d_%s" % (dval, tname[l])] = {"ms_per_step": round(ms_p, 4), "msamples_per_s": round(nfr * FRAME_SAMPLES / ms_p / 1e3, 1),
                                                                      "calls_with_wrong_rows": chk.bad - bad0}
        # ... and with the MPDUs written to the host's buffer by the frame sink itself (sora_rx_bind_mpdu, round 6) instead of copied behind the call
        for dval in (1, 2):
            rx.set_trellis(1); rx.set_depth(dval); rx.flush()
            run_block(args.warmup, deliver, dval, bound=True); rx.flush(); chk.drain(); bad0 = chk.bad
            nblk = max(1, repeats // 6)
            tp0 = time.perf_counter()
            run_block(args.steps * nblk, deliver, dval, bound=True)
            rx.flush(); chk.drain()
            ms_p = (time.perf_counter() - tp0) / (nblk * args.steps) * 1e3
            plain["calls_in_flight_%d_%s_bound_mpdu" % (dval, tname[1])] = {"ms_per_step": round(ms_p, 4), "msamples_per_s": round(nfr * FRAME_SAMPLES / ms_p / 1e3, 1),
                                                                           "calls_with_wrong_rows": chk.bad - bad0, "delivery": "sora_rx_bind_mpdu + sora_rx_deliver_async (rows and count)"}
        best1 = min((k for k in plain if k.startswith("calls_in_flight_1_")), key=lambda k: plain[k]["ms_per_step"])
        plain["calls_in_flight_1"] = dict(plain[best1], config=best1)           # (the library's automatic choice for one lone call of this size is the window-parallel trellis)
        best2 = min((k for k in plain if k.startswith("calls_in_flight_2")), key=lambda k: plain[k]["ms_per_step"])
        plain["best_with_at_most_two_calls_in_flight"] = dict(plain[best2], config=best2)
        try:
            plain["two_calls_in_flight_of_%d_captures" % (8 * nfr)] = bench_large_call(torch, sora_amd, local_rank, d_iqs, nfr, MAXF, exp_rows, exp_mpdu, args.min_seconds,
                                                                                           share[1:] if len(share) > 1 else None)
        except Exception as e:                                      # (a second 320 MB input and its workspace: report, do not lose the line)
            plain["two_calls_in_flight_of_%d_captures" % (8 * nfr)] = {"error": repr(e)}
        plain["note"] = ("the headline keeps %d calls in flight%s; with at most two, the chip holds at most 8192 frames = 1024 waves of k_viterbi16 (one per SIMD, each bound by its own "
                         "issue rate) or 4096 of k_viterbi (1.7x the instructions): DESIGN.md section 3.6, profiles/r04_a_depth_table.txt" % (depth, " on %s hardware queues" % os.environ["GPU_MAX_HW_QUEUES"] if os.environ.get("GPU_MAX_HW_QUEUES") else ""))
    rx.set_trellis(trellis_setting); rx.set_depth(depth); rx.flush()
    latency = e2e = None
    if world == 1 and not args.no_extras and not shard:
        latency = bench_latency(torch, sora_amd, dev, rx, d_iq, descs, nfr)
        e2e = bench_e2e(torch, sora_amd, dev, rx, iq, nfr, exp_rows, exp_mpdu)

    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    counters = torch.tensor([len(res), n_ok, n_payload_ok, stats["delivered"], stats["bad"]], device=dev, dtype=torch.int64)
    gathered_rows = None; gathered = None
    if world > 1:
        # the path's one exchange step: RCCL all-gathers of the device-packed result rows AND the MPDUs (every MPDU
        # reaches the one host buffer, fb11a_demod.cpp:64-70) + all-reduce of counters.  Per rank and exchange: 8 bytes of counts,
        # 36 bytes x 2 x captures of rows, MPDU_LEN x captures of MPDU bytes.
        gathered = exchange_results(torch, rx, d_iq, descs, dev, nfr, MAXF, fpc)
        gathered_rows = gathered["rows"]
        dist.all_reduce(counters)
    tot_frames, tot_ok, tot_payload_ok, tot_delivered, tot_bad = [int(v) for v in counters.tolist()]

    total_samples = float(nfr * fpc) * FRAME_SAMPLES * world * timed_steps
    msps = total_samples / elapsed / 1e6
    ms_per_step = elapsed / timed_steps * 1e3
    if rank == 0:
        dom = max((k for k in ktimes if k.startswith("k_")), key=lambda k: ktimes[k])
        launch_bytes = nfr * fpc * FRAME_SAMPLES * ALG_BYTES_PER_SAMPLE
        ach = launch_bytes / (ktimes[dom] * 1e-3)
        ach1 = launch_bytes / (ktimes1[dom] * 1e-3)
        air_s = nfr * fpc * CAPTURE_SAMPLES / 20e6                                   # what one step's captures last on the air
        out = {
            "metric": "IQ Msamples/s through 802.11a 54 Mbps RX PHY",
            "value": round(msps, 3), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "timed_steps": timed_steps, "timed_seconds": round(elapsed, 4),
            "ms_per_step": round(ms_per_step, 4), "ms_per_step_profiled": round((t3 - t2) / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int16 IQ / u8 path metrics", "data": "synthetic",
            "config": {"workload": ("802.11a 54 Mbps (64-QAM r=3/4) RX, %d concurrent captures/GPU x %d back-to-back 1500-byte frames (4880 samples @20 MHz, +160 silence each), AWGN 30/27 dB on 3 of 4: BASELINE configs[4]" % (nfr, fpc)) if shard else
                                   "802.11a 54 Mbps (64-QAM r=3/4) RX, %d captures/GPU x one 1500-byte frame (4880 samples @20 MHz, +160 silence), AWGN 30/27 dB on 3 of 4" % nfr,
                       "shape": args.shape, "captures_per_gpu": nfr, "frames_per_capture": fpc,
                       "frames_per_gpu": nfr * fpc, "samples_per_frame": FRAME_SAMPLES, "capture_samples": CAPTURE_SAMPLES, "input_copies_rotated": NCOPIES, "input_bytes_in_play": int(NCOPIES * iq.nbytes), "calls_in_flight": depth, "completions": "as they happen (sora_rx_wait_any)" if order["any"] else "oldest ticket first (sora_rx_wait)", "trellis_kernel": tname[lanes], "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "sharding": "captures per rank, no data-path collective",
                       "timed_region": "%d x %d steps in one continuous run; every step = process call + pack + async delivery of rows and MPDUs to pinned host memory + wait for %s, whose rows and MPDU bytes are compared with the verified ones by %d host threads%s" % (repeats, args.steps, "whichever call in flight finishes first (sora_rx_wait_any)" if order["any"] else "the oldest call in flight", TableChecker.EXTRA, "" if world == 1 else " (every rank pins its submit thread and its checker threads to its own slice of the host's cores)")
                                       if deliver else "%d x %d process calls, nothing delivered" % (repeats, args.steps)},
            "decoded_mbit_per_s": round(msps * (MPDU_LEN * 8.0 / FRAME_SAMPLES), 2),
            "frames": tot_frames, "gathered_rows": gathered_rows, "gathered": gathered, "frames_crc_ok": tot_ok, "frames_payload_ok": tot_payload_ok,
            "parity": {"against": kind, "captures_checked": len(idx), "ok": parity_ok, "host_rows_ok": host_rows_ok},
            "host_ms_per_step": host_ms,
            "delivery": {"enabled": deliver, "calls_delivered_and_compared": tot_delivered, "calls_with_wrong_rows": tot_bad, "rows_per_call": exp_n,
                         "row_bytes_per_call": 36 * nfr * MAXF, "mpdu_bytes_per_call": int(exp_mpdu.size), "last_calls_mpdu_ok": mpdu_ok, "calls_with_mpdu_bytes_compared": stats.get("mpdu_compared_timed", 0)},
            # the reference's own figure of merit (MACStopwatch.h:84-128): cost / required time, < 1 = faster than real time
            "realtime": {"factor": round(ms_per_step * 1e-3 / air_s, 7), "channels_20mhz_in_real_time": round(air_s / (ms_per_step * 1e-3), 1),
                         "call_latency_ms_one_in_flight": round(sum(v for k, v in ktimes1.items()), 4)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach1 / 1e9, 2), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                         "frac": round(ach1 / HBM_PEAK, 5), "traffic": measured_traffic(dom) if nfr == FRAMES_PER_GPU and not shard else None,
                         "traffic_source": _traffic_profile()[1],
                         "algorithmic_bytes_per_launch": launch_bytes, "kernel_ms": round(ktimes1[dom], 4),
                         "kernel_ms_note": "mean launch duration with ONE call in flight (the kernel alone on the chip); with %d calls overlapped the same launch lasts %.4f ms (frac %.5f) because it shares the CUs" % (depth, ktimes[dom], ach / HBM_PEAK),
                         "whole_path_frac": round(msps * 1e6 / world * ALG_BYTES_PER_SAMPLE / HBM_PEAK, 5),
                         "other_trellis_kernels": {tname[l]: {"kernel_ms": round(kt[tname[l]], 4), "frac": round(launch_bytes / (kt[tname[l]] * 1e-3) / HBM_PEAK, 5)} for l, kt in ktimes1_others.items()},
                         "other_trellis_kernels_note": "sora_rx_set_trellis, each alone on the chip: k_viterbi = two frames per wave, k_viterbi16 = eight per wave (the one for 32768 and more captures in flight), k_viterbi16w = the frames' "
                                                       "trace-back windows decoded side by side and proven afterwards (the one below that; the automatic choice follows depth x max_captures; the proof runs in the waves of the finishing kernel, k_win_redo_finish)",
                         "valu": valu_roofline(nfr, ms_per_step) if not shard else None},
            "kernel_ms": {k: round(v, 4) for k, v in ktimes.items()},
            "kernel_ms_one_call_in_flight": {k: round(v, 4) for k, v in ktimes1.items()},
        }
        try:                                                        # where the library this run loaded came from
            from sora_amd import build as _b
            out["build"] = dict(_b.build_info(), loaded=os.environ.get("SORA_HIP_LIB") or "sora_amd/lib/libsora_hip.so")
        except Exception as e:
            out["build"] = {"error": repr(e)}
        if plain:
            out["plain_host"] = plain
        if latency is not None:
            out["latency"] = latency
        if e2e is not None:
            out["e2e"] = e2e
        if world == 1 and not args.no_extras and not shard:
            out["stages"] = bench_stages(torch, sora_amd, dev)
            out["ingest"] = bench_ingest(torch, sora_amd, dev)
            out["tx"] = bench_tx(torch, sora_amd)
            out["rx11b"] = bench_11b(torch, sora_amd, dev)
            out["rx11b_cck"] = bench_11b(torch, sora_amd, dev, cpu=False, rate_kbps=11000)
            out["rx11n"] = bench_11n(torch, sora_amd, dev)
            out["rx11n_40"] = bench_ht40(torch, sora_amd, dev)
            out["shard_32x16"] = bench_shard_shape(torch, sora_amd, dev, oracle)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(iq, nfr * fpc)
            out["realtime"]["cpu_reference_factor_one_core"] = round(20.0 / out["cpu_baseline"]["single_core_value"], 4) if out["cpu_baseline"].get("single_core_value") else None
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
