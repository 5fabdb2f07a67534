/* ref_legacy_pre.h -- TEST INFRASTRUCTURE ONLY.  Force-included (after ref_compat.h) when oracle/build_ref.sh compiles the reference's LEGACY
 * 802.11a receiver (kernel/bb/dot11a: BB11ARxCarrierSense / BB11ARxFrameDemod, the C path of demod11 -d without --802.11a.brick) as a second
 * cross-check oracle.  The few driver-side names that path touches; nothing here is reference code. */
#pragma once
typedef char FLAG, *PFLAG;
typedef struct _TIMINGINFO { long long a, b; } TIMINGINFO;                 /* soratime stop-watch of the offline statistics: not measured here */
static inline void TimerStart(TIMINGINFO*) {}
static inline void TimerStop(TIMINGINFO*) {}
static inline double TimerRead(TIMINGINFO*) { return 0; }
#define KdPrint(x)
#define E_FAIL                       ((HRESULT)0x80004005L)
#define E_INVALIDARG                 ((HRESULT)0x80070057L)
#define E_FETCH_SIGNAL_HW_TIMEOUT    ((HRESULT)0x80050004L)
#define E_FETCH_SIGNAL_FORCE_STOPPED ((HRESULT)0x80050005L)
#define KeInitializeSpinLock(p)      (void)(p)
/* A dump walked block by block: what SoraGenRadioRxStreamOffline + SoraRadioReadRxStream (kernel/core/inc/rxstream.h:8-37) give the offline
 * harness (kernel/bb/demod11/demod11a.cpp:38-45): the next 28 samples, or the hardware time-out when the dump is exhausted. */
struct _SORA_RADIO_RX_STREAM { const void* base; unsigned long long nblocks, pos; };
