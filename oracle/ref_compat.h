/* ref_compat.h -- TEST INFRASTRUCTURE ONLY.  Force-included (-include) when oracle/build_ref.sh compiles the
 * reference's BRICK graphs with clang in MSVC-compatibility mode on Linux: the Windows integer model (LLP64: long and
 * ULONG are 32 bits -- the reference's arithmetic assumes it), the handful of SDK names the offline graphs touch and
 * the min/max macros of windef.h.  Nothing here is reference code. */
#pragma once
#define _MSC_VER 1900                       /* selects the decltype-based TYPEOF and static_assert-based CCASSERT */
#include <type_traits>
#include <typeinfo>
#include <cxxabi.h>
#include <new>
#include <pthread.h>                        /* (system headers come before the SAL macros below: libc / libstdc++ use names like __in) */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <stdio.h>
#include <assert.h>
#include <limits.h>
#include <sys/types.h>
#include <immintrin.h>
#define _UI64_MAX UINT64_MAX
#define _UI32_MAX UINT32_MAX
#define IN
#define OUT
#define __in
#define __out
#define __inout
#define __stdcall
#define __cdecl
#define TRUE 1
#define FALSE 0
#define S_OK 0
#define FAILED(hr) ((hr) < 0)
#define SUCCEEDED(hr) ((hr) >= 0)
#define DEFINE_GUID(...)
#define UNREFERENCED_PARAMETER(x) (void)(x)
#define ASSERT(x) assert(x)
typedef unsigned char  uchar, UCHAR, *PUCHAR, BYTE, BOOLEAN;
typedef char           CHAR, *PCHAR;
typedef unsigned short ushort, USHORT, WORD, *PUSHORT;
typedef short          SHORT;
typedef unsigned int   uint, UINT, DWORD, *PUINT;
typedef int            INT, BOOL, LONG, HRESULT;
typedef uint32_t       sora_ulong32, ULONG, *PULONG;
#define ulong sora_ulong32                  /* glibc's <sys/types.h> already has a 64-bit ulong */
typedef int64_t        LONGLONG;
typedef uint64_t       ULONGLONG;
typedef void           VOID, *PVOID, *HANDLE, *PPACKET_BASE, *SORA_ETHREAD;
typedef uintptr_t      UPOINTER, ULONG_PTR;
typedef long           KSPIN_LOCK;
typedef union _LARGE_INTEGER { struct { uint32_t LowPart; int32_t HighPart; } u; int64_t QuadPart; } LARGE_INTEGER, *PLARGE_INTEGER;
static inline int QueryPerformanceCounter(LARGE_INTEGER* p) { p->QuadPart = 0; return 1; }
static inline int QueryPerformanceFrequency(LARGE_INTEGER* p) { p->QuadPart = 1; return 1; }
#define SORA_RX_SIGNAL_UNIT_NUM_PER_DESC  7 /* _rx_manager.h:79-81: an RX_BLOCK carries 7 x 16-byte units = 28 COMPLEX16 */
#define SORA_RX_SIGNAL_UNIT_COMPLEX16_NUM 4
#define SORA_RX_SIGNAL_UNIT_SIZE          16
#define M128_WORD_NUM                     8
static inline void* _aligned_malloc(size_t size, size_t align) { return aligned_alloc(align, (size + align - 1) / align * align); }
static inline void  _aligned_free(void* p) { free(p); }
/* the sources pass "unsigned long*" (32 bits on Windows, 64 here) to the intrinsic */
template<class T> static inline unsigned char sora_bit_scan_reverse(T* index, uint32_t mask)
{ if (!mask) return 0; *index = (T)(31 - __builtin_clz(mask)); return 1; }
#define _BitScanReverse sora_bit_scan_reverse
#define min(a, b) (((a) < (b)) ? (a) : (b))
#define max(a, b) (((a) > (b)) ? (a) : (b))
