"""ctypes binding of the C ABI in include/sora_hip.h (libsora_hip.so).

This is plumbing only: device memory comes from torch (or sora_hip_malloc), all compute is the HIP library.
The wrapper mirrors the reference harness: `Rx.process(...)` = RxThread over a batch of captures,
`Rx.results()` = the frames TBB11aFrameSink reported.  Nothing here falls back to a CPU implementation:
if the library or a GPU is missing, calls raise.
"""
import collections
import ctypes
import os

import numpy as np

from . import build as _build

SORA_OK = 0
E_FRAME_OK = 0x1
TRELLIS_WINDOWED = 1          # sora_rx_set_trellis: the window-parallel trellis (include/sora_hip.h SORA_TRELLIS_WINDOWED)
E_PLCP_HEADER_FAIL = 0x80000005
E_CRC32_FAIL = 0x80000006
ERR_NO_DEVICE = -5
ERR_CAPACITY = -6


class SoraError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("sora_hip error %d (0x%08X): %s" % (code, code & 0xFFFFFFFF, msg))
        self.code = code


class RxCfg(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("device", ctypes.c_int32), ("sample_rate_mhz", ctypes.c_uint32),
                ("max_captures", ctypes.c_uint32), ("max_total_samples", ctypes.c_uint64),
                ("max_frames_per_capture", ctypes.c_uint32), ("cca_pwr_threshold", ctypes.c_uint32)]


class CaptureDesc(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_uint64), ("nsamples", ctypes.c_uint32), ("capture_id", ctypes.c_uint32)]


class FrameResult(ctypes.Structure):
    _fields_ = [("capture_id", ctypes.c_uint32), ("start_sample", ctypes.c_uint32), ("end_sample", ctypes.c_uint32),
                ("error_code", ctypes.c_uint32), ("rate_kbps", ctypes.c_uint32), ("length", ctypes.c_uint16),
                ("nsym", ctypes.c_uint16), ("crc32", ctypes.c_uint32), ("cfo_est", ctypes.c_int16),
                ("flags", ctypes.c_uint16), ("mpdu_offset", ctypes.c_uint32)]


class Ht40Frame(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_uint64), ("n_bpsc", ctypes.c_uint32), ("code_rate", ctypes.c_uint32), ("length", ctypes.c_uint32 * 2),
                ("cfo", ctypes.c_int32), ("noise_var", ctypes.c_float), ("frame_id", ctypes.c_uint32)]


EXPORTS = ["sora_hip_abi_version", "sora_hip_last_error", "sora_hip_device_count", "sora_hip_malloc", "sora_hip_free",
           "sora_hip_memcpy_h2d", "sora_hip_memcpy_d2h", "sora_hip_memcpy_d2d", "sora_hip_stream_synchronize", "sora_rx_create", "sora_rx_destroy", "sora_rx_reset",
           "sora_rx_flush", "sora_rx_stream", "sora_rx_process_dev", "sora_rx_process_dump", "sora_rx_set_stream_mode", "sora_rx_stream_consumed", "sora_rx_process", "sora_rx_results",
           "sora_rx_results_dev", "sora_rx_ticket", "sora_rx_wait", "sora_rx_wait_any", "sora_rx_results_of", "sora_rx_results_dev_of", "sora_rx_stream_of",
           "sora_rx_mpdu_bytes", "sora_rx_deliver_async", "sora_hip_host_alloc", "sora_hip_host_free", "sora_rx_set_profiling", "sora_rx_kernel_times", "sora_rx_kernel_name",
                      "sora_rx_set_depth", "sora_rx_set_fused", "sora_rx_set_trellis", "sora_rx_trellis", "sora_rx_window_stats", "sora_rx_set_front", "sora_rx_front",
                      "sora_rx_call_front", "sora_rx_set_ordered", "sora_hip_set_share_window_us", "sora_rx_set_pipe_wait_us", "sora_rx_pipe_stats", "sora_rx_bind_mpdu",
                      "sora_hip_table_count", "sora_hip_table_name", "sora_hip_table_pin", "sora_hip_table_digest", "sora_hip_table_read", "sora_rx_set_graph",
                      "sora_rx_kernel_name_fused", "sora_hip_fft64", "sora_hip_fft128", "sora_hip_lts11a", "sora_hip_symfront11a", "sora_hip_pilot_track11a", "sora_hip_pilot11a",
                      "sora_hip_freq_comp11a", "sora_hip_equalize11a", "sora_hip_phase_comp11a", "sora_hip_demap11a", "sora_hip_deinterleave11a", "sora_hip_viterbi11a",
                      "sora_hip_viterbi11a_ws", "sora_hip_viterbi11a_workspace_bytes",
           "sora_hip_ingest", "sora_hip_ingest_count", "sora_hip_tx11a", "sora_hip_tx11a_samples",
           "sora_hip_demap11n", "sora_hip_deinterleave11n", "sora_hip_mimo_est11n", "sora_hip_mimo_comp11n", "sora_hip_cfo_est11n", "sora_hip_freq_comp11n",
                      "sora_hip_pilot_track11n", "sora_hip_siso_est11n", "sora_hip_siso_comp11n", "sora_hip_sig_demap11n", "sora_hip_sig_decode11n", "sora_rx11b_create",
                      "sora_rx11b_destroy", "sora_rx11b_stream", "sora_rx11b_synchronize", "sora_rx11b_process_dev", "sora_rx11b_process", "sora_rx11b_results", "sora_rx11b_ticket",
                      "sora_rx11b_calls_in_flight", "sora_rx11b_set_single_pass", "sora_rx11b_wait", "sora_rx11b_wait_any", "sora_rx11b_stream_of", "sora_rx11b_results_of",
                      "sora_rx11b_deliver_async", "sora_rx11n_deliver_async", "sora_ht40_deliver_async",
           "sora_rx11n_create", "sora_rx11n_destroy", "sora_rx11n_stream", "sora_rx11n_process_dev", "sora_rx11n_process", "sora_rx11n_results",
           "sora_rx11n_set_depth", "sora_rx11n_set_trellis", "sora_rx11n_trellis", "sora_rx11n_window_stats", "sora_rx11n_synchronize", "sora_rx11n_ticket", "sora_rx11n_wait",
                      "sora_rx11n_wait_any", "sora_rx11n_results_of",
           "sora_ht40_symbols", "sora_ht40_create", "sora_ht40_destroy", "sora_ht40_stream", "sora_ht40_synchronize", "sora_ht40_set_trellis", "sora_ht40_process_dev",
                      "sora_ht40_process_captures_dev", "sora_ht40_results", "sora_ht40_ticket", "sora_ht40_calls_in_flight", "sora_ht40_wait", "sora_ht40_wait_any",
                      "sora_ht40_stream_of", "sora_ht40_results_of",
           "sora_shard_unique_id", "sora_shard_create", "sora_shard_destroy", "sora_shard_world", "sora_shard_partition", "sora_shard_gather_rows",
           "sora_shard_reduce_counters", "sora_shard_gather_results", "sora_shard_gather_results_mpdu"]

_lib = None


def table_names():
    """the library's pinned look-up tables (include/sora_hip.h: sora_hip_table_*)"""
    L = load()
    return [L.sora_hip_table_name(i).decode() for i in range(L.sora_hip_table_count())]


def table_pin(name):
    r = load().sora_hip_table_pin(name.encode())
    return r.decode() if r else None


def table_digest(name):
    """sha256 of the table as this build on this host generates it (no device needed)"""
    buf = ctypes.create_string_buffer(65)
    _check(load().sora_hip_table_digest(name.encode(), buf))
    return buf.value.decode()


def table_read(name):
    """the device-resident table of the current device, as bytes"""
    L = load(); n = ctypes.c_size_t(0)
    _check(L.sora_hip_table_read(name.encode(), None, 0, ctypes.byref(n)))
    buf = ctypes.create_string_buffer(n.value)
    _check(L.sora_hip_table_read(name.encode(), buf, n.value, ctypes.byref(n)))
    return buf.raw[:n.value]


def lib_path():
    return _build.LIB


def load(build_if_missing=True):
    """Load libsora_hip.so (building it with hipcc if absent).  Raises if it cannot be had."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64 (same soname as /opt/rocm's).  Whichever copy is mapped first
    # serves the whole process, and torch only initialises against its own: import torch BEFORE dlopen-ing
    # libsora_hip.so so both share torch's runtime.  (A pure C host has no such concern.)
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    path = os.environ.get("SORA_HIP_LIB") or _build.LIB             # SORA_HIP_LIB: an experimental build of the same library (tools/ab_*.sh)
    if path == _build.LIB and build_if_missing and _build.needs_build():
        _build.build()
    if not os.path.exists(path):
        raise SoraError(-1, "libsora_hip.so is missing and could not be built; there is no CPU fallback")
    L = ctypes.CDLL(path)
    L.sora_hip_last_error.restype = ctypes.c_char_p
    L.sora_hip_malloc.restype = ctypes.c_void_p
    L.sora_hip_malloc.argtypes = [ctypes.c_size_t]
    L.sora_hip_free.argtypes = [ctypes.c_void_p]
    L.sora_hip_memcpy_h2d.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.sora_hip_memcpy_d2h.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.sora_rx_create.argtypes = [ctypes.POINTER(RxCfg), ctypes.POINTER(ctypes.c_void_p)]
    L.sora_rx_destroy.argtypes = [ctypes.c_void_p]; L.sora_rx_destroy.restype = None
    L.sora_rx_reset.argtypes = [ctypes.c_void_p]
    L.sora_rx_flush.argtypes = [ctypes.c_void_p]
    L.sora_rx_stream.argtypes = [ctypes.c_void_p]; L.sora_rx_stream.restype = ctypes.c_void_p
    L.sora_rx_process_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(CaptureDesc), ctypes.c_size_t]
    L.sora_rx_process.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(CaptureDesc), ctypes.c_size_t]
    L.sora_rx_results.argtypes = [ctypes.c_void_p, ctypes.POINTER(FrameResult), ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t),
                                  ctypes.c_void_p, ctypes.c_size_t]
    L.sora_rx_results_dev.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]
    L.sora_rx_ticket.argtypes = [ctypes.c_void_p]
    L.sora_rx_wait.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sora_rx_wait_any.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    L.sora_rx_results_of.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(FrameResult), ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t),
                                     ctypes.c_void_p, ctypes.c_size_t]
    L.sora_rx_results_dev_of.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]
    L.sora_rx_stream_of.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.sora_rx_stream_of.restype = ctypes.c_void_p
    L.sora_rx_mpdu_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int]; L.sora_rx_mpdu_bytes.restype = ctypes.c_size_t
    L.sora_rx_deliver_async.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.sora_hip_host_alloc.argtypes = [ctypes.c_size_t]; L.sora_hip_host_alloc.restype = ctypes.c_void_p
    L.sora_hip_host_free.argtypes = [ctypes.c_void_p]; L.sora_hip_host_free.restype = None
    L.sora_rx_set_profiling.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sora_rx_kernel_times.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.sora_rx_kernel_name.argtypes = [ctypes.c_size_t]; L.sora_rx_kernel_name.restype = ctypes.c_char_p
    L.sora_rx_set_depth.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sora_rx_set_stream_mode.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sora_rx_stream_consumed.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
    L.sora_rx_process_dump.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.POINTER(CaptureDesc), ctypes.c_size_t]
    L.sora_rx_set_fused.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sora_rx_set_trellis.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sora_rx_set_graph.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sora_rx_trellis.argtypes = [ctypes.c_void_p]
    L.sora_rx_window_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
    L.sora_rx_set_front.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sora_rx_front.argtypes = [ctypes.c_void_p]
    L.sora_rx_call_front.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sora_rx_set_ordered.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sora_hip_set_share_window_us.argtypes = [ctypes.c_uint]; L.sora_hip_set_share_window_us.restype = ctypes.c_uint
    L.sora_rx_bind_mpdu.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.sora_rx_set_pipe_wait_us.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    L.sora_rx_pipe_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
    L.sora_hip_table_count.argtypes = []
    L.sora_hip_table_name.argtypes = [ctypes.c_int]; L.sora_hip_table_name.restype = ctypes.c_char_p
    L.sora_hip_table_pin.argtypes = [ctypes.c_char_p]; L.sora_hip_table_pin.restype = ctypes.c_char_p
    L.sora_hip_table_digest.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    L.sora_hip_table_read.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    L.sora_rx11n_set_trellis.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sora_rx11n_trellis.argtypes = [ctypes.c_void_p]
    L.sora_rx11n_window_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]
    L.sora_ht40_set_trellis.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sora_rx_kernel_name_fused.argtypes = [ctypes.c_size_t]; L.sora_rx_kernel_name_fused.restype = ctypes.c_char_p
    L.sora_ht40_symbols.argtypes = [ctypes.c_uint32] * 4; L.sora_ht40_symbols.restype = ctypes.c_uint32
    L.sora_ht40_create.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p)]
    L.sora_ht40_destroy.argtypes = [ctypes.c_void_p]; L.sora_ht40_destroy.restype = None
    L.sora_ht40_stream.argtypes = [ctypes.c_void_p]; L.sora_ht40_stream.restype = ctypes.c_void_p
    L.sora_ht40_synchronize.argtypes = [ctypes.c_void_p]
    L.sora_ht40_process_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(Ht40Frame), ctypes.c_size_t, ctypes.c_void_p]
    L.sora_ht40_results.argtypes = [ctypes.c_void_p, ctypes.POINTER(FrameResult), ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p, ctypes.c_size_t]
    L.sora_shard_unique_id.argtypes = [ctypes.c_void_p]
    L.sora_shard_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    L.sora_shard_destroy.argtypes = [ctypes.c_void_p]; L.sora_shard_destroy.restype = None
    L.sora_shard_world.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    L.sora_shard_partition.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    L.sora_shard_partition.restype = None
    L.sora_shard_gather_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.sora_shard_reduce_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.sora_shard_gather_results.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.POINTER(FrameResult),
                                            ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_size_t)]
    L.sora_shard_gather_results_mpdu.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.POINTER(FrameResult),
                                                 ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_size_t), ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_size_t)]
    L.sora_hip_fft64.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_fft128.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_lts11a.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_symfront11a.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    for fn in ("sora_hip_freq_comp11a", "sora_hip_equalize11a", "sora_hip_phase_comp11a"):
        getattr(L, fn).argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_pilot_track11a.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_pilot11a.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_demap11a.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_deinterleave11a.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_viterbi11a.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_viterbi11a_workspace_bytes.argtypes = [ctypes.c_size_t, ctypes.c_size_t]; L.sora_hip_viterbi11a_workspace_bytes.restype = ctypes.c_size_t
    L.sora_hip_viterbi11a_ws.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    L.sora_hip_stream_synchronize.argtypes = [ctypes.c_void_p]
    L.sora_hip_tx11a_samples.argtypes = [ctypes.c_uint32, ctypes.c_uint32]; L.sora_hip_tx11a_samples.restype = ctypes.c_size_t
    L.sora_hip_tx11a.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.sora_hip_ingest_count.argtypes = [ctypes.c_size_t, ctypes.c_uint]; L.sora_hip_ingest_count.restype = ctypes.c_size_t
    L.sora_hip_ingest.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p, ctypes.c_size_t,
                                  ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
    L.sora_hip_demap11n.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_deinterleave11n.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_cfo_est11n.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_freq_comp11n.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_pilot_track11n.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_mimo_est11n.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_mimo_comp11n.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_siso_est11n.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_siso_comp11n.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_sig_demap11n.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_size_t, ctypes.c_void_p]
    L.sora_hip_sig_decode11n.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_size_t, ctypes.c_void_p]
    L.sora_rx11n_create.argtypes = [ctypes.POINTER(RxCfg), ctypes.POINTER(ctypes.c_void_p)]
    L.sora_rx11n_destroy.argtypes = [ctypes.c_void_p]; L.sora_rx11n_destroy.restype = None
    L.sora_rx11n_stream.argtypes = [ctypes.c_void_p]; L.sora_rx11n_stream.restype = ctypes.c_void_p
    L.sora_rx11n_process_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(CaptureDesc), ctypes.c_size_t]
    L.sora_rx11n_process.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(CaptureDesc), ctypes.c_size_t]
    L.sora_rx11n_set_depth.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sora_rx11n_ticket.argtypes = [ctypes.c_void_p]
    L.sora_rx11n_wait.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sora_rx11n_wait_any.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    L.sora_rx11n_results_of.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(FrameResult), ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p, ctypes.c_size_t]
    L.sora_rx11n_results.argtypes = [ctypes.c_void_p, ctypes.POINTER(FrameResult), ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t),
                                     ctypes.c_void_p, ctypes.c_size_t]
    L.sora_rx11b_create.argtypes = [ctypes.POINTER(RxCfg), ctypes.POINTER(ctypes.c_void_p)]
    L.sora_rx11b_set_single_pass.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.sora_rx11b_destroy.argtypes = [ctypes.c_void_p]; L.sora_rx11b_destroy.restype = None
    L.sora_rx11b_stream.argtypes = [ctypes.c_void_p]; L.sora_rx11b_stream.restype = ctypes.c_void_p
    L.sora_rx11b_synchronize.argtypes = [ctypes.c_void_p]
    L.sora_rx11b_process_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(CaptureDesc), ctypes.c_size_t]
    L.sora_rx11b_process.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(CaptureDesc), ctypes.c_size_t]
    _res_of = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(FrameResult), ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p, ctypes.c_size_t]
    _deliver = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.sora_ht40_process_captures_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32]
    for pre in ("sora_rx11b", "sora_ht40"):
        getattr(L, pre + "_ticket").argtypes = [ctypes.c_void_p]
        getattr(L, pre + "_calls_in_flight").argtypes = [ctypes.c_void_p]
        getattr(L, pre + "_wait").argtypes = [ctypes.c_void_p, ctypes.c_int]
        getattr(L, pre + "_wait_any").argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        getattr(L, pre + "_stream_of").argtypes = [ctypes.c_void_p, ctypes.c_int]; getattr(L, pre + "_stream_of").restype = ctypes.c_void_p
        getattr(L, pre + "_results_of").argtypes = _res_of
        getattr(L, pre + "_deliver_async").argtypes = _deliver
    L.sora_rx11n_deliver_async.argtypes = _deliver
    L.sora_rx11n_synchronize.argtypes = [ctypes.c_void_p]
    L.sora_rx11b_results.argtypes = [ctypes.c_void_p, ctypes.POINTER(FrameResult), ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t),
                                     ctypes.c_void_p, ctypes.c_size_t]
    _lib = L
    return L


def _check(rc):
    if rc != SORA_OK:
        raise SoraError(rc, (load().sora_hip_last_error() or b"").decode())


def device_count():
    return load().sora_hip_device_count()


def _dev_ptr(x):
    """Device pointer of a torch CUDA tensor (or an int address)."""
    if isinstance(x, int):
        return x
    if hasattr(x, "data_ptr"):
        if not x.is_cuda:
            raise ValueError("expected a device (HBM) tensor")
        if not x.is_contiguous():
            raise ValueError("expected a contiguous tensor")
        return x.data_ptr()
    raise TypeError("expected a torch CUDA tensor or an integer device address")


class Rx:
    """sora_rx_t: the 802.11a demod graph over a batch of captures."""

    def __init__(self, max_captures, max_total_samples, sample_rate_mhz=20, device=0, max_frames_per_capture=2,
                 cca_pwr_threshold=0):
        L = load()
        cfg = RxCfg(ctypes.sizeof(RxCfg), device, sample_rate_mhz, max_captures, max_total_samples,
                    max_frames_per_capture, cca_pwr_threshold)
        h = ctypes.c_void_p()
        _check(L.sora_rx_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h; self._L = L; self.cfg = cfg; self._keep = None
        self.wait_for_producer = True      # process_dev first waits for torch's current stream (bench.py turns it off: its inputs are resident)
        # harness-side overrides (the library itself reads no environment variable): A/B runs of the tools and tests
        if os.environ.get("SORA_HIP_DEPTH"): self.set_depth(int(os.environ["SORA_HIP_DEPTH"]))
        if os.environ.get("SORA_HIP_FUSED"): self.set_fused(int(os.environ["SORA_HIP_FUSED"]))
        if os.environ.get("SORA_HIP_GRAPH"): self.set_graph(int(os.environ["SORA_HIP_GRAPH"]))
        if os.environ.get("SORA_HIP_TRELLIS"): self.set_trellis(int(os.environ["SORA_HIP_TRELLIS"]))
        if os.environ.get("SORA_HIP_FRONT"): self.set_front(int(os.environ["SORA_HIP_FRONT"]))

    def close(self):
        if self._h:
            self._L.sora_rx_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return self._L.sora_rx_stream(self._h)

    CAPTURE_DTYPE = np.dtype([("offset", "<u8"), ("nsamples", "<u4"), ("capture_id", "<u4")])   # = sora_capture_desc

    @classmethod
    def captures(cls, captures):
        """[(offset, nsamples[, id])] -> packed sora_capture_desc array (build it once when the same set is submitted repeatedly)."""
        if isinstance(captures, np.ndarray) and captures.dtype == cls.CAPTURE_DTYPE:
            return np.ascontiguousarray(captures)
        arr = np.zeros(len(captures), cls.CAPTURE_DTYPE)
        if len(captures):
            if all(len(c) > 2 for c in captures):
                a = np.asarray(captures, dtype=np.uint64).reshape(len(captures), -1)
                arr["offset"] = a[:, 0]; arr["nsamples"] = a[:, 1]; arr["capture_id"] = a[:, 2]
            else:
                for i, c in enumerate(captures):
                    arr[i] = (c[0], c[1], c[2] if len(c) > 2 else i)
        return arr

    @classmethod
    def _caps(cls, captures):
        arr = cls.captures(captures)
        return arr, arr.ctypes.data_as(ctypes.POINTER(CaptureDesc))

    def process_dev(self, d_iq, captures):
        """d_iq: int16 torch CUDA tensor [N,2] (resident in HBM); captures: [(offset, nsamples[, id])]."""
        arr, ptr = self._caps(captures)
        _hold(self, d_iq)
        _check(self._L.sora_rx_process_dev(self._h, _dev_ptr(d_iq), ptr, len(arr)))
        return self._L.sora_rx_ticket(self._h)

    def process(self, h_iq, captures):
        a = np.ascontiguousarray(h_iq, np.int16).reshape(-1, 2)
        arr, ptr = self._caps(captures)
        _check(self._L.sora_rx_process(self._h, a.ctypes.data, len(a), ptr, len(arr)))
        return self._L.sora_rx_ticket(self._h)

    def set_stream_mode(self, enable=-1):
        """1: capture k of a call continues capture k of the call before it (sora_hip.h: stream continuation); returns the previous mode"""
        r = int(self._L.sora_rx_set_stream_mode(self._h, int(enable)))
        if r not in (0, 1):
            raise SoraError(r, (self._L.sora_hip_last_error() or b"").decode())
        return r

    def stream_consumed(self, ticket, ncaps):
        """per capture of the most recent call: input-rate samples that are final = where the next call's capture must start in the stream"""
        out = np.zeros(ncaps, np.uint32)
        _check(self._L.sora_rx_stream_consumed(self._h, int(ticket), out.ctypes.data, int(ncaps)))
        return out

    def process_dump(self, h_dump, flags, captures):
        """h_dump: the raw dump bytes in host memory -- a numpy uint8 array or a (pinned) torch CPU uint8 tensor, untouched until the call has
        completed; flags: INGEST_*; captures address the ingested stream.  Copy, ingest and the receive chain run on the call's own stream."""
        arr, ptr = self._caps(captures)
        if hasattr(h_dump, "data_ptr"):
            addr, nbytes = h_dump.data_ptr(), h_dump.numel() * h_dump.element_size()
        else:
            a = np.ascontiguousarray(h_dump).view(np.uint8).reshape(-1); addr, nbytes = a.ctypes.data, a.size; h_dump = a
        if self._keep is None:
            self._keep = collections.deque(maxlen=17)
        self._keep.append(h_dump)
        _check(self._L.sora_rx_process_dump(self._h, ctypes.c_void_p(addr), nbytes, int(flags), ptr, len(arr)))
        return self._L.sora_rx_ticket(self._h)

    def ticket(self):
        """ticket of the most recent process call (0: none)"""
        return self._L.sora_rx_ticket(self._h)

    def wait(self, ticket):
        _check(self._L.sora_rx_wait(self._h, int(ticket)))

    def wait_any(self):
        """Block until some call with an enqueued delivery (deliver_async) has finished -> its ticket (the oldest finished one).  The call is then
        released: the next process call may reuse its pipeline ahead of older calls still in flight."""
        t = ctypes.c_int(0)
        _check(self._L.sora_rx_wait_any(self._h, ctypes.byref(t)))
        return t.value

    def mpdu_bytes(self, ticket):
        return int(self._L.sora_rx_mpdu_bytes(self._h, int(ticket)))

    def deliver_async(self, ticket, buf):
        """Enqueue the delivery of the call's rows (+ MPDU array when buf.mpdu is not None) into the page-locked HostResults `buf`."""
        _check(self._L.sora_rx_deliver_async(self._h, int(ticket), buf.rows.ctypes.data, len(buf.rows), buf.nrows.ctypes.data,
                                             buf.mpdu.ctypes.data if buf.mpdu is not None else None, buf.mpdu.size if buf.mpdu is not None else 0))

    def results_dev(self, ticket=None):
        """Device-resident results of the last call (or of the call `ticket` names): (rows int32 tensor [cap_rows, 9] (36-byte
        sora_frame_result rows), nrows int32 tensor [1], mpdu uint8 base address).  The tensors alias library memory: valid
        until that pipeline's next call."""
        import torch
        rows = ctypes.c_void_p(); nrows = ctypes.c_void_p(); mpdu = ctypes.c_void_p()
        if ticket is None:
            _check(self._L.sora_rx_results_dev(self._h, ctypes.byref(rows), ctypes.byref(nrows), ctypes.byref(mpdu)))
        else:
            _check(self._L.sora_rx_results_dev_of(self._h, int(ticket), ctypes.byref(rows), ctypes.byref(nrows), ctypes.byref(mpdu)))
        cap = self.cfg.max_captures * self.cfg.max_frames_per_capture
        dev = torch.device("cuda", self.cfg.device)

        class _Arr:                                        # __cuda_array_interface__ view over library memory
            def __init__(self, ptr, shape, typestr):
                self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2}
        r = torch.as_tensor(_Arr(rows.value, (cap, 9), "<i4"), device=dev)
        n = torch.as_tensor(_Arr(nrows.value, (1,), "<i4"), device=dev)
        return r, n, mpdu.value

    def set_profiling(self, enable=True):
        _check(self._L.sora_rx_set_profiling(self._h, 1 if enable else 0))

    def set_depth(self, depth=0):
        """number of process calls kept in flight on internal pipelines (1..8); returns the previous value"""
        return int(self._L.sora_rx_set_depth(self._h, int(depth)))

    def set_trellis(self, lanes_per_pair=-1):
        """64: k_viterbi, 16: k_viterbi16, 1 (TRELLIS_WINDOWED): k_viterbi16w, 0: the library chooses from the capacity in flight; returns the previous setting"""
        r = int(self._L.sora_rx_set_trellis(self._h, int(lanes_per_pair)))
        if r < 0: _check(r)
        return r

    def set_front(self, kernels=-1):
        """1: k_frame (one wave per frame), 3: k_sym_front -> k_track_lds -> k_sym_back, 4: k_pipe (those and the window-parallel trellis as one launch),
        0: the library chooses; returns the previous setting"""
        r = int(self._L.sora_rx_set_front(self._h, int(kernels)))
        if r < 0: _check(r)
        return r

    def front(self):
        return int(self._L.sora_rx_front(self._h))

    def set_ordered(self, on=-1):
        """1: calls in flight complete in submission order (a call's trellis kernel starts behind the previous call's); returns the previous setting"""
        r = int(self._L.sora_rx_set_ordered(self._h, int(on)))
        if r < 0: _check(r)
        return r

    def call_front(self, ticket=0):
        """the kernels call `ticket` (0: the most recent one) WAS launched with -- latched at its process call, where front() is a forecast"""
        r = int(self._L.sora_rx_call_front(self._h, int(ticket)))
        if r < 0: _check(r)
        return r

    def trellis(self):
        """lanes per frame pair of the trellis kernel the next call uses (64: k_viterbi, 16: k_viterbi16, 1 = TRELLIS_WINDOWED: k_viterbi16w)"""
        return int(self._L.sora_rx_trellis(self._h))

    def window_stats(self):
        """the window-parallel trellis's proof record since the handle was created (waits for the calls in flight)"""
        v = (ctypes.c_ulonglong * 4)()
        _check(self._L.sora_rx_window_stats(self._h, v))
        return {"boundaries": int(v[0]), "boundaries_failed": int(v[1]), "frames_decoded_again": int(v[2]), "units": int(v[3])}

    def bind_mpdu(self, buf):
        """the NEXT process call writes its MPDUs straight into the page-locked HostResults `buf` (deliver_async(ticket, buf) then copies rows and count only); None cancels"""
        if buf is None:
            _check(self._L.sora_rx_bind_mpdu(self._h, None, 0))
        else:
            _check(self._L.sora_rx_bind_mpdu(self._h, buf.mpdu.ctypes.data, buf.mpdu.size))

    def set_pipe_wait_us(self, us=-1):
        """bound of the waits inside a k_pipe launch (0: every wait that is not satisfied at once gives up -> the call is made again by the finishing kernel)"""
        return int(self._L.sora_rx_set_pipe_wait_us(self._h, int(us)))

    def pipe_stats(self):
        """k_pipe's safety net since the handle was created (waits for the calls in flight)"""
        v = (ctypes.c_ulonglong * 2)()
        _check(self._L.sora_rx_pipe_stats(self._h, v))
        return {"calls_made_again": int(v[0]), "backoffs": int(v[1])}

    def set_graph(self, enable=-1):
        return int(self._L.sora_rx_set_graph(self._h, int(enable)))

    def set_fused(self, enable=-1):
        """1: decode the data field with the fused kernel (k_decode -- a build variant since round 4: SoraError SORA_E_NOT_SUPPORTED in the
        default library), 0: k_frame + k_viterbi; returns the previous setting"""
        r = int(self._L.sora_rx_set_fused(self._h, int(enable)))
        if r not in (0, 1):
            raise SoraError(r, (self._L.sora_hip_last_error() or b"").decode())
        return r

    def kernel_times(self):
        """{kernel name: ms} of the profiled process calls (HIP events on the handle's streams)."""
        ms = (ctypes.c_float * 8)(); n = ctypes.c_size_t(0)
        _check(self._L.sora_rx_kernel_times(self._h, ms, 8, ctypes.byref(n)))
        name = self._L.sora_rx_kernel_name_fused if self.set_fused(-1) else self._L.sora_rx_kernel_name
        return {name(i).decode(): ms[i] for i in range(n.value) if name(i)}

    def flush(self):
        _check(self._L.sora_rx_flush(self._h))

    def reset(self):
        _check(self._L.sora_rx_reset(self._h))

    def results(self, with_mpdu=True, max_frames=None, ticket=None):
        """Frames of the most recent process call, or of the call `ticket` names."""
        if max_frames is None:
            max_frames = self.cfg.max_captures * self.cfg.max_frames_per_capture
        res = (FrameResult * max(1, max_frames))()
        n = ctypes.c_size_t(0)
        mp = np.zeros(max_frames * 2504 if with_mpdu else 1, np.uint8)
        if ticket is None:
            _check(self._L.sora_rx_results(self._h, res, max_frames, ctypes.byref(n), mp.ctypes.data if with_mpdu else None, mp.size))
        else:
            _check(self._L.sora_rx_results_of(self._h, int(ticket), res, max_frames, ctypes.byref(n), mp.ctypes.data if with_mpdu else None, mp.size))
        out = []
        for i in range(n.value):
            r = res[i]
            d = {f: getattr(r, f) for f, _ in FrameResult._fields_}
            if with_mpdu:
                d["mpdu"] = mp[r.mpdu_offset:r.mpdu_offset + r.length].tobytes() if r.error_code in (E_FRAME_OK, E_CRC32_FAIL) else b""
            out.append(d)
        return out


ROW_DTYPE = np.dtype([("capture_id", "<u4"), ("start_sample", "<u4"), ("end_sample", "<u4"), ("error_code", "<u4"), ("rate_kbps", "<u4"),
                      ("length", "<u2"), ("nsym", "<u2"), ("crc32", "<u4"), ("cfo_est", "<i2"), ("flags", "<u2"), ("mpdu_offset", "<u4")])   # = sora_frame_result
ROW_TRUNCATED = 1


class HostResults:
    """Page-locked host buffers for Rx.deliver_async: rows (ROW_DTYPE), nrows (uint32[1]) and, optionally, the MPDU array
    (row["mpdu_offset"] indexes it).  Memory comes from sora_hip_host_alloc; close() releases it."""

    def __init__(self, max_rows, mpdu_bytes=0):
        L = load()
        self._L = L
        assert ROW_DTYPE.itemsize == ctypes.sizeof(FrameResult) == 36
        sizes = [max(1, max_rows) * 36, 64, max(0, int(mpdu_bytes))]
        self._ptrs = []
        arrs = []
        for nbytes in sizes:
            if nbytes == 0:
                arrs.append(None); continue
            p = L.sora_hip_host_alloc(nbytes)
            if not p:
                raise SoraError(-1, "sora_hip_host_alloc failed")
            self._ptrs.append(p)
            arrs.append(np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(p)))
        self.rows = arrs[0].view(ROW_DTYPE)
        self.nrows = arrs[1][:4].view(np.uint32)
        self.counts = arrs[1][:8].view(np.uint32)       # the 11b / 11n / HT40 handles deliver {rows, MPDU bytes}
        self.mpdu = arrs[2]

    def results(self):
        """the delivered table as the list of dicts Rx*.results() returns (after wait(ticket)): dense MPDUs, mpdu_offset indexes self.mpdu"""
        out = []
        for r in self.rows[:int(self.counts[0])]:
            d = {k: int(r[k]) for k in ROW_DTYPE.names}
            has = d["error_code"] in (E_FRAME_OK, E_CRC32_FAIL) and self.mpdu is not None
            d["mpdu"] = self.mpdu[d["mpdu_offset"]:d["mpdu_offset"] + min(d["length"], 4096)].tobytes() if has else b""
            d["stream"] = d["start_sample"]
            out.append(d)
        return out

    def close(self):
        for p in self._ptrs:
            self._L.sora_hip_host_free(p)
        self._ptrs = []
        self.rows = self.nrows = self.mpdu = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Rx11b:
    """sora_rx11b_t: the 802.11b demod graph (44 MHz samples) over a batch of captures."""

    def __init__(self, max_captures, max_total_samples, device=0, max_frames_per_capture=8, cca_pwr_threshold=0):
        L = load()
        cfg = RxCfg(ctypes.sizeof(RxCfg), device, 44, max_captures, max_total_samples, max_frames_per_capture, cca_pwr_threshold)
        h = ctypes.c_void_p()
        _check(L.sora_rx11b_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h; self._L = L; self.cfg = cfg; self._keep = None

    def close(self):
        if self._h:
            self._L.sora_rx11b_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(self._L.sora_rx11b_synchronize(self._h))

    def process_dev(self, d_iq, captures):
        arr, ptr = Rx._caps(captures)
        _hold(self, d_iq)
        _check(self._L.sora_rx11b_process_dev(self._h, _dev_ptr(d_iq), ptr, len(arr)))
        return self._L.sora_rx11b_ticket(self._h)

    def ticket(self):
        return self._L.sora_rx11b_ticket(self._h)

    def calls_in_flight(self):
        return int(self._L.sora_rx11b_calls_in_flight(self._h))

    def set_single_pass(self, enable=-1):
        """the pass plan: 2 automatic (default), 1 every capture straight through the CCK-capable kernel, 0 always two passes; returns the previous plan"""
        return int(self._L.sora_rx11b_set_single_pass(self._h, int(enable)))

    def wait(self, ticket):
        _check(self._L.sora_rx11b_wait(self._h, int(ticket)))

    def wait_any(self):
        """-> the ticket of the oldest finished call with an enqueued delivery (blocks until there is one); that call is released"""
        t = ctypes.c_int(0)
        _check(self._L.sora_rx11b_wait_any(self._h, ctypes.byref(t)))
        return t.value

    def deliver_async(self, ticket, buf):
        """rows + dense MPDUs of the call into the page-locked HostResults `buf`, behind the call's kernels; valid after wait(ticket)"""
        _check(self._L.sora_rx11b_deliver_async(self._h, int(ticket), buf.rows.ctypes.data, len(buf.rows), buf.counts.ctypes.data,
                          buf.mpdu.ctypes.data if buf.mpdu is not None else None, buf.mpdu.size if buf.mpdu is not None else 0))

    def process(self, h_iq, captures):
        a = np.ascontiguousarray(h_iq, np.int16).reshape(-1, 2)
        arr, ptr = Rx._caps(captures)
        _check(self._L.sora_rx11b_process(self._h, a.ctypes.data, len(a), ptr, len(arr)))

    def results(self, ticket=None, with_mpdu=True):
        max_frames = self.cfg.max_captures * self.cfg.max_frames_per_capture
        res = (FrameResult * max(1, max_frames))()
        n = ctypes.c_size_t(0)
        mp = np.zeros(max_frames * 4096 if with_mpdu else 1, np.uint8)
        if ticket is None:
            _check(self._L.sora_rx11b_results(self._h, res, max_frames, ctypes.byref(n), mp.ctypes.data if with_mpdu else None, mp.size))
        else:
            _check(self._L.sora_rx11b_results_of(self._h, int(ticket), res, max_frames, ctypes.byref(n), mp.ctypes.data if with_mpdu else None, mp.size))
        out = []
        for r in res[:n.value]:
            d = {f: getattr(r, f) for f, _ in FrameResult._fields_}
            d["mpdu"] = mp[r.mpdu_offset:r.mpdu_offset + min(r.length, 4096)].tobytes() if with_mpdu and r.error_code in (E_FRAME_OK, E_CRC32_FAIL) else b""
            out.append(d)
        return out


class Rx11n:
    """sora_rx11n_t: the 802.11n 2x2 demod graph (two RX chains at 40 MHz) over a batch of captures; rate_kbps = MCS index."""

    def __init__(self, max_captures, max_total_samples, device=0, max_frames_per_capture=8):
        L = load()
        cfg = RxCfg(ctypes.sizeof(RxCfg), device, 40, max_captures, max_total_samples, max_frames_per_capture, 0)
        h = ctypes.c_void_p()
        _check(L.sora_rx11n_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h; self._L = L; self.cfg = cfg; self._keep = None

    def close(self):
        if self._h:
            self._L.sora_rx11n_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        """waits for every call in flight"""
        t = self._L.sora_rx11n_ticket(self._h)
        for k in range(max(1, t - 3), t + 1):
            if self._L.sora_rx11n_wait(self._h, k) != SORA_OK:
                pass                                                     # a ticket whose pipeline has been reused: that call finished long ago

    def set_depth(self, depth=0):
        r = self._L.sora_rx11n_set_depth(self._h, depth)
        if r < 0: _check(r)
        return r

    def set_trellis(self, lanes_per_pair=-1):
        r = int(self._L.sora_rx11n_set_trellis(self._h, int(lanes_per_pair)))
        if r < 0: _check(r)
        return r

    def trellis(self):
        """the trellis kernel the next call uses: 64, 16 or 1 (TRELLIS_WINDOWED)"""
        return int(self._L.sora_rx11n_trellis(self._h))

    def window_stats(self):
        v = (ctypes.c_ulonglong * 4)()
        _check(self._L.sora_rx11n_window_stats(self._h, v))
        return {"boundaries": int(v[0]), "boundaries_failed": int(v[1]), "frames_decoded_again": int(v[2]), "units": int(v[3])}

    def wait(self, ticket):
        _check(self._L.sora_rx11n_wait(self._h, int(ticket)))

    def wait_any(self):
        """-> the ticket of the oldest finished call with an enqueued delivery (blocks until there is one); that call is released"""
        t = ctypes.c_int(0)
        _check(self._L.sora_rx11n_wait_any(self._h, ctypes.byref(t)))
        return t.value

    def deliver_async(self, ticket, buf):
        """rows + dense MPDUs of the call into the page-locked HostResults `buf`, behind the call's kernels; valid after wait(ticket)"""
        _check(self._L.sora_rx11n_deliver_async(self._h, int(ticket), buf.rows.ctypes.data, len(buf.rows), buf.counts.ctypes.data,
                          buf.mpdu.ctypes.data if buf.mpdu is not None else None, buf.mpdu.size if buf.mpdu is not None else 0))

    def process_dev(self, d_iq0, d_iq1, captures):
        arr, ptr = Rx._caps(captures)
        _hold(self, (d_iq0, d_iq1))
        _check(self._L.sora_rx11n_process_dev(self._h, _dev_ptr(d_iq0), _dev_ptr(d_iq1), ptr, len(arr)))
        return self._L.sora_rx11n_ticket(self._h)

    def process(self, h_iq0, h_iq1, captures):
        a = np.ascontiguousarray(h_iq0, np.int16).reshape(-1, 2); b = np.ascontiguousarray(h_iq1, np.int16).reshape(-1, 2)
        assert len(a) == len(b)
        arr, ptr = Rx._caps(captures)
        _check(self._L.sora_rx11n_process(self._h, a.ctypes.data, b.ctypes.data, len(a), ptr, len(arr)))

    def results(self, ticket=None):
        max_frames = self.cfg.max_captures * self.cfg.max_frames_per_capture
        res = (FrameResult * max(1, max_frames))()
        n = ctypes.c_size_t(0)
        mp = np.zeros(max_frames * 4096, np.uint8)
        if ticket is None:
            _check(self._L.sora_rx11n_results(self._h, res, max_frames, ctypes.byref(n), mp.ctypes.data, mp.size))
        else:
            _check(self._L.sora_rx11n_results_of(self._h, int(ticket), res, max_frames, ctypes.byref(n), mp.ctypes.data, mp.size))
        out = []
        for r in res[:n.value]:
            d = {f: getattr(r, f) for f, _ in FrameResult._fields_}
            d["mpdu"] = mp[r.mpdu_offset:r.mpdu_offset + min(r.length, 4096)].tobytes() if r.error_code in (E_FRAME_OK, E_CRC32_FAIL) else b""
            out.append(d)
        return out


class RxHt40:
    """sora_ht40_t: the data field of HT-mixed 40 MHz two-stream frames (BASELINE configs[3]; parity unpinned, see include/sora_hip.h)."""

    def __init__(self, max_frames, max_soft_values, device=0):
        L = load()
        h = ctypes.c_void_p()
        _check(L.sora_ht40_create(device, max_frames, max_soft_values, ctypes.byref(h)))
        self._h = h; self._L = L; self.max_frames = max_frames; self._keep = None

    def close(self):
        if self._h:
            self._L.sora_ht40_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(self._L.sora_ht40_synchronize(self._h))

    @staticmethod
    def frames(descs):
        """[(offset, n_bpsc, code_rate, length0, length1, cfo, noise_var[, id])] -> packed descriptor array (reusable)"""
        if isinstance(descs, ctypes.Array):
            return descs
        arr = (Ht40Frame * max(1, len(descs)))()
        for i, d in enumerate(descs):
            arr[i].offset = d[0]; arr[i].n_bpsc = d[1]; arr[i].code_rate = d[2]; arr[i].length[0] = d[3]; arr[i].length[1] = d[4]
            arr[i].cfo = d[5]; arr[i].noise_var = d[6]; arr[i].frame_id = d[7] if len(d) > 7 else i
        arr._n = len(descs)
        return arr

    def set_trellis(self, lanes_per_pair=-1):
        r = int(self._L.sora_ht40_set_trellis(self._h, int(lanes_per_pair)))
        if r < 0: _check(r)
        return r

    def process_dev(self, d_iq0, d_iq1, descs, d_weights=None):
        arr = self.frames(descs); n = getattr(arr, "_n", len(arr))
        _hold(self, (d_iq0, d_iq1, d_weights)); self._n = n
        _check(self._L.sora_ht40_process_dev(self._h, _dev_ptr(d_iq0), _dev_ptr(d_iq1), arr, n, _dev_ptr(d_weights) if d_weights is not None else None))
        t = self._L.sora_ht40_ticket(self._h)
        self._nof = getattr(self, "_nof", {}); self._nof[t] = n
        self._prune_nof()
        return t

    def _prune_nof(self):
        """forget the row bounds of tickets the LIBRARY no longer knows (a released slot is reused ahead of older calls still in flight, so ticket arithmetic
        does not tell: ADVICE r4)"""
        for old in [k for k in self._nof if not self._L.sora_ht40_stream_of(self._h, int(k))]:
            del self._nof[old]

    def process_captures_dev(self, d_iq0, d_iq1, captures, max_frames_per_capture=4):
        """raw two-chain 40 MHz captures [(offset, nsamples[, id])]: the front end finds, parses and measures the frames"""
        arr, ptr = Rx._caps(captures)
        _hold(self, (d_iq0, d_iq1))
        _check(self._L.sora_ht40_process_captures_dev(self._h, _dev_ptr(d_iq0), _dev_ptr(d_iq1), ptr, len(arr), int(max_frames_per_capture)))
        t = self._L.sora_ht40_ticket(self._h)
        self._n = len(arr) * int(max_frames_per_capture)
        self._nof = getattr(self, "_nof", {}); self._nof[t] = self._n
        self._prune_nof()
        return t

    def ticket(self):
        return self._L.sora_ht40_ticket(self._h)

    def calls_in_flight(self):
        return int(self._L.sora_ht40_calls_in_flight(self._h))

    def wait(self, ticket):
        _check(self._L.sora_ht40_wait(self._h, int(ticket)))

    def wait_any(self):
        """-> the ticket of the oldest finished call with an enqueued delivery (blocks until there is one); that call is released"""
        t = ctypes.c_int(0)
        _check(self._L.sora_ht40_wait_any(self._h, ctypes.byref(t)))
        return t.value

    def deliver_async(self, ticket, buf):
        """rows + dense MPDUs of the call into the page-locked HostResults `buf`, behind the call's kernels; valid after wait(ticket)"""
        _check(self._L.sora_ht40_deliver_async(self._h, int(ticket), buf.rows.ctypes.data, len(buf.rows), buf.counts.ctypes.data,
                          buf.mpdu.ctypes.data if buf.mpdu is not None else None, buf.mpdu.size if buf.mpdu is not None else 0))

    def results(self, with_mpdu=True, ticket=None):
        n2 = 2 * (self._n if ticket is None else self._nof.get(int(ticket), self.max_frames))
        res = (FrameResult * max(1, n2))(); n = ctypes.c_size_t(0)
        mp = np.zeros(n2 * 4096 if with_mpdu else 1, np.uint8)
        if ticket is None:
            _check(self._L.sora_ht40_results(self._h, res, n2, ctypes.byref(n), mp.ctypes.data if with_mpdu else None, mp.size))
        else:
            _check(self._L.sora_ht40_results_of(self._h, int(ticket), res, n2, ctypes.byref(n), mp.ctypes.data if with_mpdu else None, mp.size))
        out = []
        for r in res[:n.value]:
            d = {f: getattr(r, f) for f, _ in FrameResult._fields_}
            d["stream"] = d["start_sample"]
            if with_mpdu:
                d["mpdu"] = mp[r.mpdu_offset:r.mpdu_offset + r.length].tobytes()
            out.append(d)
        return out


def ht40_symbols(length0, length1, n_bpsc, code_rate):
    return load().sora_ht40_symbols(length0, length1, n_bpsc, code_rate)


# ---- per-stage entry points on torch CUDA tensors ---------------------------------------------------
def _hold(obj, item, sync=True):
    """The handles keep several calls in flight on their own non-blocking streams (up to 16: sora_rx_set_depth): (i) the inputs of the last 17 calls stay referenced
    (torch's caching allocator must not hand a block to the next tensor while a kernel still reads it), (ii) whatever produced the
    tensors on torch's current stream has finished before the library's stream reads them."""
    import collections
    if getattr(obj, "_keep", None) is None:
        obj._keep = collections.deque(maxlen=17)
    obj._keep.append(item)
    if sync and getattr(obj, "wait_for_producer", True):
        import torch
        torch.cuda.current_stream().synchronize()


def _stream_ptr(stream):
    if stream is None:
        import torch
        return torch.cuda.current_stream().cuda_stream
    return stream


def fft64(x, stream=None):
    """TFFT64: x int16 CUDA tensor [n,64,2] -> same shape."""
    import torch
    out = torch.empty_like(x)
    _check(load().sora_hip_fft64(_dev_ptr(x), _dev_ptr(out), x.shape[0], _stream_ptr(stream)))
    return out


def fft128(x, stream=None):
    """FFT<128>: x int16 CUDA tensor [n,128,2] -> same shape."""
    import torch
    out = torch.empty_like(x)
    _check(load().sora_hip_fft128(_dev_ptr(x), _dev_ptr(out), x.shape[0], _stream_ptr(stream)))
    return out


def lts11a(x, stream=None):
    """T11aLTS: x int16 CUDA [n,144,2] -> int16 CUDA [n,258]: (cfo_est, reserved, freq[64][2], chan[64][2])."""
    import torch
    out = torch.zeros((x.shape[0], 258), dtype=torch.int16, device=x.device)
    _check(load().sora_hip_lts11a(_dev_ptr(x), _dev_ptr(out), x.shape[0], _stream_ptr(stream)))
    return out


def symfront11a(x, ctx, ctx_index=None, stream=None):
    """x int16 CUDA [n,80,2], ctx = lts11a() output [m,258], ctx_index int32 CUDA [n] or None -> eq int16 [n,64,2]."""
    import torch
    out = torch.empty((x.shape[0], 64, 2), dtype=torch.int16, device=x.device)
    _check(load().sora_hip_symfront11a(_dev_ptr(x), _dev_ptr(ctx), _dev_ptr(ctx_index) if ctx_index is not None else None,
                                       _dev_ptr(out), x.shape[0], _stream_ptr(stream)))
    return out


def _cmul64(fn, x, coef, index, stream):
    import torch
    out = torch.empty_like(x)
    _check(getattr(load(), fn)(_dev_ptr(x), _dev_ptr(coef), _dev_ptr(index) if index is not None else None, _dev_ptr(out), x.shape[0], _stream_ptr(stream)))
    return out


def freq_comp11a(x, ctx, ctx_index=None, stream=None):
    """TFreqCompensation alone: x int16 CUDA [n,64,2], ctx = lts11a() output [m,258], ctx_index int32 CUDA [n] or None -> [n,64,2]"""
    return _cmul64("sora_hip_freq_comp11a", x, ctx, ctx_index, stream)


def equalize11a(x, ctx, ctx_index=None, stream=None):
    """TChannelEqualization alone (same arguments)"""
    return _cmul64("sora_hip_equalize11a", x, ctx, ctx_index, stream)


def phase_comp11a(x, state, state_index=None, stream=None):
    """TPhaseCompensate alone: state int16 CUDA [m,134] (sora_track11a_state), state_index int32 CUDA [n] or None"""
    return _cmul64("sora_hip_phase_comp11a", x, state, state_index, stream)


def pilot_track11a(eq, first, nsym, state, stream=None):
    """eq int16 CUDA [total,64,2]; first/nsym int32 CUDA [f]; state int16 CUDA [f,134] (in/out) -> tracked [total,64,2]."""
    import torch
    out = torch.zeros_like(eq)
    _check(load().sora_hip_pilot_track11a(_dev_ptr(eq), _dev_ptr(first), _dev_ptr(nsym), _dev_ptr(state), _dev_ptr(out), first.shape[0], _stream_ptr(stream)))
    return out


def set_share_window_us(us):
    """the window in which another batch-sized handle's call makes a small handle's automatic choice leave k_pipe alone (process-wide; returns the old value)"""
    return int(load().sora_hip_set_share_window_us(int(us)))


def pilot11a(x, first, nsym, state, stream=None):
    """TPilotTrack alone: x = TPhaseCompensate's output, int16 CUDA [total,64,2]; the frame tables and the state of pilot_track11a."""
    import torch
    out = torch.zeros_like(x)
    _check(load().sora_hip_pilot11a(_dev_ptr(x), _dev_ptr(first), _dev_ptr(nsym), _dev_ptr(state), _dev_ptr(out), first.shape[0], _stream_ptr(stream)))
    return out


def demap11a(x, n_bpsc, stream=None):
    import torch
    out = torch.empty((x.shape[0], 48 * n_bpsc), dtype=torch.uint8, device=x.device)
    _check(load().sora_hip_demap11a(_dev_ptr(x), _dev_ptr(out), n_bpsc, x.shape[0], _stream_ptr(stream)))
    return out


def demap11n(x, n_bpsc, stream=None):
    """x: int16 CUDA tensor [n,64,2] (pilot-tracked symbols of one spatial stream) -> uint8 [n, 52*n_bpsc]."""
    import torch
    out = torch.empty((x.shape[0], 52 * n_bpsc), dtype=torch.uint8, device=x.device)
    _check(load().sora_hip_demap11n(_dev_ptr(x), _dev_ptr(out), n_bpsc, x.shape[0], _stream_ptr(stream)))
    return out


def deinterleave11n(s, n_bpsc, spatial_stream, stream=None):
    """s: uint8 CUDA tensor [n, 52*n_bpsc] of spatial stream 0 or 1 -> de-interleaved, same shape."""
    import torch
    out = torch.empty_like(s)
    _check(load().sora_hip_deinterleave11n(_dev_ptr(s), _dev_ptr(out), n_bpsc, spatial_stream, s.shape[0], _stream_ptr(stream)))
    return out


def cfo_est11n(lltf0, lltf1, stream=None):
    """lltf0/lltf1: int16 CUDA [n,128,2] -> state int16 CUDA [n,24] (vfo_delta_i | vfo_step_i | vfo_theta_i)."""
    import torch
    st = torch.empty((lltf0.shape[0], 24), dtype=torch.int16, device=lltf0.device)
    _check(load().sora_hip_cfo_est11n(_dev_ptr(lltf0), _dev_ptr(lltf1), _dev_ptr(st), lltf0.shape[0], _stream_ptr(stream)))
    return st


def freq_comp11n(in0, in1, first, nbursts, state, stream=None):
    """in0/in1: int16 CUDA [N,2]; first/nbursts: int32 CUDA [nframes]; state: int16 CUDA [nframes,24] (updated in place) -> (out0, out1)."""
    import torch
    o0 = in0.clone(); o1 = in1.clone()
    mx = int(nbursts.max().item()) if nbursts.numel() else 0
    _check(load().sora_hip_freq_comp11n(_dev_ptr(in0), _dev_ptr(in1), _dev_ptr(o0), _dev_ptr(o1), _dev_ptr(first), _dev_ptr(nbursts), _dev_ptr(state),
                                        state.shape[0], mx, _stream_ptr(stream)))
    return o0, o1


def pilot_track11n(x0, x1, first, nsym, state, want_theta=True, stream=None):
    """x0/x1: int16 CUDA [nsym_total,64,2]; first/nsym: int32 CUDA [nframes]; state updated in place -> theta int16 [nsym_total,8] or None."""
    import torch
    th = torch.zeros((x0.shape[0], 8), dtype=torch.int16, device=x0.device) if want_theta else None
    _check(load().sora_hip_pilot_track11n(_dev_ptr(x0), _dev_ptr(x1), _dev_ptr(first), _dev_ptr(nsym), _dev_ptr(state),
                                          _dev_ptr(th) if want_theta else None, state.shape[0], _stream_ptr(stream)))
    return th


def mimo_est11n(ltf0, ltf1, stream=None):
    """ltf0/ltf1: int16 CUDA tensors [n,128,2] (the two HT-LTF symbols of RX chain 0 / 1 after the FFT) -> (h, hinv) int16 [n,2,128,2]."""
    import torch
    n = ltf0.shape[0]
    h = torch.empty((n, 2, 128, 2), dtype=torch.int16, device=ltf0.device); hinv = torch.empty_like(h)
    _check(load().sora_hip_mimo_est11n(_dev_ptr(ltf0), _dev_ptr(ltf1), _dev_ptr(h), _dev_ptr(hinv), n, _stream_ptr(stream)))
    return h, hinv


def mimo_comp11n(hinv, y0, y1, frame_index=None, stream=None):
    """hinv: int16 [nframes,2,128,2]; y0/y1: int16 [nsym,64,2] (RX chain 0 / 1 after the FFT); frame_index: int32/uint32 [nsym] or None
    -> (x0, x1) int16 [nsym,64,2], the two spatial streams."""
    import torch
    x0 = torch.empty_like(y0); x1 = torch.empty_like(y1)
    _check(load().sora_hip_mimo_comp11n(_dev_ptr(hinv), _dev_ptr(frame_index) if frame_index is not None else None, _dev_ptr(y0), _dev_ptr(y1),
                                        _dev_ptr(x0), _dev_ptr(x1), y0.shape[0], _stream_ptr(stream)))
    return x0, x1


def siso_est11n(lltf0, lltf1, stream=None):
    """lltf0/lltf1: int16 CUDA tensors [n,128,2] (the two L-LTF symbols of RX chain 0 / 1 after the FFT) -> ch int16 [n,2,64,2]."""
    import torch
    n = lltf0.shape[0]
    ch = torch.empty((n, 2, 64, 2), dtype=torch.int16, device=lltf0.device)
    _check(load().sora_hip_siso_est11n(_dev_ptr(lltf0), _dev_ptr(lltf1), _dev_ptr(ch), n, _stream_ptr(stream)))
    return ch


def siso_comp11n(ch, y0, y1, frame_index=None, stream=None):
    """ch: int16 [nframes,2,64,2]; y0/y1: int16 [nsym,64,2] -> (x0, x1, mrc) int16 [nsym,64,2] (TSisoChannelComp then TMrcCombine)."""
    import torch
    x0 = torch.empty_like(y0); x1 = torch.empty_like(y1); m = torch.empty_like(y0)
    _check(load().sora_hip_siso_comp11n(_dev_ptr(ch), _dev_ptr(frame_index) if frame_index is not None else None, _dev_ptr(y0), _dev_ptr(y1),
                                        _dev_ptr(x0), _dev_ptr(x1), _dev_ptr(m), y0.shape[0], _stream_ptr(stream)))
    return x0, x1, m


def sig_demap11n(sym, stream=None):
    """sym: int16 [n,3,64,2] (L-SIG, HT-SIG1, HT-SIG2 after MRC) -> soft uint8 [n,144]."""
    import torch
    soft = torch.empty((sym.shape[0], 144), dtype=torch.uint8, device=sym.device)
    _check(load().sora_hip_sig_demap11n(_dev_ptr(sym), _dev_ptr(soft), sym.shape[0], _stream_ptr(stream)))
    return soft


def sig_decode11n(soft, stream=None):
    """soft: uint8 [n,144] (sig_demap11n's output) -> int32 [n,12]: error_code, data_rate_kbps, frame_length, ht_frame_mcs, ht_frame_length,
    code_rate, total_symbols, remain_symbols, symbol_type, L-SIG, HT-SIG bits 0..31, HT-SIG bits 32..41."""
    import torch
    rec = torch.empty((soft.shape[0], 12), dtype=torch.int32, device=soft.device)
    _check(load().sora_hip_sig_decode11n(_dev_ptr(soft), _dev_ptr(rec), soft.shape[0], _stream_ptr(stream)))
    return rec


def deinterleave11a(s, n_bpsc, stream=None):
    import torch
    out = torch.empty_like(s)
    _check(load().sora_hip_deinterleave11a(_dev_ptr(s), _dev_ptr(out), n_bpsc, s.shape[0], _stream_ptr(stream)))
    return out


def viterbi11a(soft, soft_off, nsoft, frame_len, code_rate, out_stride=2560, stream=None):
    """soft: uint8 CUDA tensor; soft_off/nsoft: int32 CUDA tensors [n]; frame_len: int16 CUDA [n]."""
    import torch
    n = soft_off.shape[0]
    out = torch.zeros((n, out_stride), dtype=torch.uint8, device=soft.device)
    out_off = (torch.arange(n, device=soft.device, dtype=torch.int32) * out_stride).contiguous()
    _check(load().sora_hip_viterbi11a(_dev_ptr(soft), _dev_ptr(soft_off), _dev_ptr(nsoft), _dev_ptr(frame_len), code_rate,
                                      _dev_ptr(out), _dev_ptr(out_off), n, _stream_ptr(stream)))
    return out


def viterbi11a_workspace_bytes(soft_span_bytes, n):
    return int(load().sora_hip_viterbi11a_workspace_bytes(int(soft_span_bytes), int(n)))


def viterbi11a_ws(soft, soft_off, nsoft, frame_len, code_rate, workspace, out=None, out_off=None, out_stride=2560, lanes_per_pair=0, stream=None):
    """The Viterbi brick out of a caller-owned workspace (uint8 CUDA tensor of >= viterbi11a_workspace_bytes(soft.numel(), n)):
    no allocation and no host wait inside the call; the caller synchronises the stream before reading `out`."""
    import torch
    n = soft_off.shape[0]
    if out is None:
        out = torch.zeros((n, out_stride), dtype=torch.uint8, device=soft.device)
        out_off = (torch.arange(n, device=soft.device, dtype=torch.int32) * out_stride).contiguous()
    _check(load().sora_hip_viterbi11a_ws(_dev_ptr(soft), soft.numel(), _dev_ptr(soft_off), _dev_ptr(nsoft), _dev_ptr(frame_len), code_rate,
                                         _dev_ptr(out), _dev_ptr(out_off), n, _dev_ptr(workspace), workspace.numel(), int(lanes_per_pair), _stream_ptr(stream)))
    return out


INGEST_RXBLOCK, INGEST_RAW14, INGEST_44TO40, INGEST_DECIMATE2 = 1, 2, 4, 8


def ingest_count(raw_bytes, flags):
    return int(load().sora_hip_ingest_count(int(raw_bytes), int(flags)))


def ingest(raw, flags, stream=None, sync=True):
    """raw: uint8 (dump bytes) or int16 [n,2] CUDA tensor -> int16 [m,2] CUDA tensor (de-framed / sign-fixed / 44->40 / decimated).
    sync: wait for the kernel, so that the result can go straight to Rx.process_dev (whose streams do not follow the null stream)."""
    import torch
    nbytes = raw.numel() * raw.element_size()
    n = ingest_count(nbytes, flags)
    out = torch.empty((max(n, 1), 2), dtype=torch.int16, device=raw.device)
    got = ctypes.c_size_t(0)
    _check(load().sora_hip_ingest(_dev_ptr(raw), nbytes, int(flags), _dev_ptr(out), n, ctypes.byref(got), _stream_ptr(stream)))
    if sync:
        _check(load().sora_hip_stream_synchronize(_stream_ptr(stream)))
    return out[:got.value]


def tx11a_samples(mpdu_len_nofcs, rate_kbps):
    return int(load().sora_hip_tx11a_samples(int(mpdu_len_nofcs), int(rate_kbps)))


def tx11a(mpdus, rates_kbps, seeds=None, device=0, stream=None, sync=True, gaps=None):
    """Modulate a batch of MPDUs (bytes WITHOUT FCS) on the GPU.  -> (int8 CUDA tensor [total,2] COMPLEX8 @40 MHz, offsets list).
    Frame f occupies samples offsets[f] .. offsets[f+1] (gaps[f] zero samples in front of frame f, if given, included at its start)."""
    import torch
    n = len(mpdus)
    seeds = [0xFF] * n if seeds is None else list(seeds)
    lens = [len(m) for m in mpdus]
    off = np.zeros(n + 1, np.int64); np.cumsum([(l + 3) // 4 * 4 for l in lens], out=off[1:])
    blob = np.zeros(max(int(off[-1]), 4), np.uint8)
    for f, m in enumerate(mpdus):
        blob[off[f]:off[f] + lens[f]] = np.frombuffer(bytes(m), np.uint8)
    ns = [tx11a_samples(l, r) for l, r in zip(lens, rates_kbps)]
    if any(v == 0 for v in ns):
        raise SoraError(-1, "tx11a: unsupported rate or length")
    gaps = [0] * n if gaps is None else [int(v) for v in gaps]
    ooff = np.zeros(n + 1, np.uint64); np.cumsum([a + b for a, b in zip(ns, gaps)], out=ooff[1:])
    first = ooff[:-1] + np.asarray(gaps, np.uint64)
    dev = torch.device("cuda", device)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
    d_blob = t(blob, np.uint8); d_off = torch.from_numpy(off[:-1].astype(np.int32)).to(dev)
    d_len = torch.from_numpy(np.asarray(lens, np.int32)).to(dev); d_rate = torch.from_numpy(np.asarray(rates_kbps, np.int32)).to(dev)
    d_seed = t(np.asarray(seeds, np.uint8), np.uint8); d_ooff = torch.from_numpy(first.astype(np.int64)).to(dev)
    out = torch.zeros((int(ooff[-1]), 2), dtype=torch.int8, device=dev)
    _check(load().sora_hip_tx11a(_dev_ptr(d_blob), _dev_ptr(d_off), _dev_ptr(d_len), _dev_ptr(d_rate), _dev_ptr(d_seed), n,
                                 _dev_ptr(out), _dev_ptr(d_ooff), _stream_ptr(stream)))
    if sync:
        _check(load().sora_hip_stream_synchronize(_stream_ptr(stream)))
    return out, [int(v) for v in ooff]
