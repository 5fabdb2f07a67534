"""sora_amd -- MI355X-native 802.11a receive PHY behind Sora's BRICK operator shapes.

The product is sora_amd/lib/libsora_hip.so (hand-written HIP for gfx950, C ABI in include/sora_hip.h);
this package is the thin Python binding used by the tests and bench.py.  There is no CPU compute path.
"""
from .capi import (Rx, Rx11b, Rx11n, RxHt40, ht40_symbols, HostResults, ROW_DTYPE, ROW_TRUNCATED, SoraError, device_count, load, lib_path, fft64, fft128, lts11a, symfront11a,
    pilot_track11a, pilot11a, set_share_window_us, freq_comp11a, equalize11a, phase_comp11a, demap11a, deinterleave11a, demap11n, deinterleave11n, mimo_est11n, mimo_comp11n,
    cfo_est11n, freq_comp11n, pilot_track11n, siso_est11n, siso_comp11n, sig_demap11n, sig_decode11n, viterbi11a, viterbi11a_ws, viterbi11a_workspace_bytes,  # noqa: F401
                   ingest, ingest_count, tx11a, tx11a_samples, INGEST_RXBLOCK, INGEST_RAW14, INGEST_44TO40, INGEST_DECIMATE2,
                   E_FRAME_OK, E_CRC32_FAIL, E_PLCP_HEADER_FAIL, TRELLIS_WINDOWED, table_names, table_pin, table_digest, table_read)
