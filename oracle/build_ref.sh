#!/bin/bash
# build_ref.sh -- build oracle/_ref/libsora_ref.so from the REFERENCE's own arithmetic headers.
#
# TEST INFRASTRUCTURE.  Compiles the reference sources where they lie under $SORA_REFERENCE
# (default /root/reference); nothing from the reference is copied into this repository: the
# MSVC-only constructs are patched in a scratch directory that is removed after the compile and the
# only output is oracle/_ref/libsora_ref.so (git-ignored).  On a box without the reference tree the
# script is a no-op (the prebuilt .so, if any, is used as it is).
#
# Patches:
#   * vector128.h: drop the hand re-declared intrinsics (lines 26-81), include <immintrin.h>, drop the
#     two non-existent *_epi64 wrappers
#   * stub const.h / sora.h providing FINL/SELECTANY/A16 and Windows LLP64 integer types
#     (ULONG/ulong = 32 bits -- the reference assumes sizeof(long)==4)
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${SORA_REFERENCE:-/root/reference}"
OUT="$HERE/_ref"
if [ ! -d "$REF/kernel/core/inc" ]; then
    echo "build_ref.sh: reference tree not found at $REF -- skipping (using prebuilt $OUT if present)"
    exit 0
fi
CXX="${SORA_REF_CXX:-/opt/rocm/lib/llvm/bin/clang++}"
TMP="$(mktemp -d /tmp/sora_ref_build.XXXXXX)"
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$OUT"

CI="$REF/kernel/core/inc"; BS="$REF/kernel/bb/Brick11/src"
for f in complex.h fft_r4dif.h ifft_r4dif.h fft_lut_twiddle.h fft_lut_bitreversal.h intalg.h intalglut.h CRC32.h \
         operator_repeater.h tpltrick.h unroll.h; do
    cp "$CI/$f" "$TMP/$f"
done
for f in viterbicore.h viterbilut.h demapper.h 44MTo40M.hpp; do cp "$BS/$f" "$TMP/$f"; done
mkdir -p "$TMP/bb"; printf '#pragma once\n#include "const.h"\n' > "$TMP/bb/bba.h"      # 44MTo40M.hpp only needs the integer types from it
sed -e '26,81d' "$CI/vector128.h" \
  | sed -e 's|#include <emmintrin.h>|#include <immintrin.h>|' \
        -e '/_mm_sign_epi64/d' -e '/_mm_abs_epi64/d' > "$TMP/vector128.h"

cat > "$TMP/const.h" <<'EOF'
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <sys/types.h>
#include <new>
#define _UI64_MAX UINT64_MAX
#define _UI32_MAX UINT32_MAX
#define FINL      __forceinline
#define SELECTANY __declspec(selectany)
#define A16       __declspec(align(16))
#define IN
#define OUT
#define SORA_EXTERN_C extern "C"
typedef unsigned char  uchar, UCHAR, *PUCHAR;
typedef unsigned short ushort, USHORT;
typedef unsigned int   uint, UINT;
typedef uint32_t       sora_ulong32, ULONG, *PULONG;   /* Windows LLP64: long is 32 bits */
#define ulong sora_ulong32                              /* glibc already typedefs a 64-bit ulong */
#define MEM_ALIGN(n) __declspec(align(n))
#define SORA_RX_SIGNAL_UNIT_NUM_PER_DESC  7               /* _rx_manager.h:79-81: 7 x 16-byte units = 28 COMPLEX16 per RX_BLOCK */
#define SORA_RX_SIGNAL_UNIT_COMPLEX16_NUM 4
EOF
cat > "$TMP/sora.h" <<'EOF'
#pragma once
#include "const.h"
EOF

"$CXX" -std=c++14 -O2 -U__OPTIMIZE__ -fPIC -shared -fvisibility=hidden \
    -fms-extensions -fms-compatibility -fdelayed-template-parsing -fno-operator-names \
    -msse4.1 -mssse3 -Wno-everything \
    -I"$TMP" "$HERE/ref_shim.cpp" -o "$OUT/libsora_ref.so" &
PID_KERNELS=$!

# ---- the reference's BRICK graphs themselves (oracle/ref_flatten.py patches a scratch copy; oracle/ref_compat.h
#      supplies the Windows integer model) -> oracle/_ref/libsora_refgraph.so
python3 "$HERE/ref_flatten.py" "$TMP/flat"
"$CXX" -std=c++14 -O2 -U__OPTIMIZE__ -fPIC -shared -fvisibility=hidden \
    -fms-extensions -fms-compatibility -fms-compatibility-version=19.00 -fdelayed-template-parsing -fno-operator-names \
    -msse4.1 -mssse3 -Wno-everything -DUSER_MODE -D__XSAVEINTRIN_H -include "$HERE/ref_compat.h" \
    -I"$TMP/flat" "$HERE/ref_graph_shim.cpp" -o "$OUT/libsora_refgraph.so" &
PID_GRAPH=$!

# ---- the 11a receive graph once more WITH the reference's thread boundary (TThreadSeparator + a joined Viterbi thread):
#      only there to show that the two-thread harness reports what the same-thread build reports (tests/test_oracle_vs_refgraph.py)
python3 "$HERE/ref_flatten.py" "$TMP/flat_mt" mt
"$CXX" -std=c++14 -O2 -U__OPTIMIZE__ -fPIC -shared -fvisibility=hidden -pthread \
    -fms-extensions -fms-compatibility -fms-compatibility-version=19.00 -fdelayed-template-parsing -fno-operator-names \
    -msse4.1 -mssse3 -Wno-everything -DUSER_MODE -D__XSAVEINTRIN_H -include "$HERE/ref_compat.h" \
    -I"$TMP/flat_mt" "$HERE/ref_graph_mt_shim.cpp" -o "$OUT/libsora_refgraph_mt.so" &
PID_MT=$!
# ---- the LEGACY 802.11a receiver (kernel/bb/dot11a: the C path behind BB11ARxFrameDemod), the second cross-check oracle of SURVEY section 8 f4
python3 "$HERE/ref_flatten.py" "$TMP/flat_legacy" legacy
cp "$HERE/ref_legacy_rxstream.h" "$TMP/flat_legacy/ref_legacy_rxstream.h"
"$CXX" -std=c++14 -O2 -U__OPTIMIZE__ -fPIC -shared -fvisibility=hidden -pthread \
    -fms-extensions -fms-compatibility -fms-compatibility-version=19.00 -fdelayed-template-parsing -fno-operator-names \
    -msse4.1 -mssse3 -Wno-everything -DUSER_MODE -DSTATIC_LUT -D__XSAVEINTRIN_H -include "$HERE/ref_compat.h" -include "$HERE/ref_legacy_pre.h" \
    -I"$TMP/flat_legacy" -I"$TMP/flat_legacy/bb" -I"$TMP/flat_legacy/inc" "$HERE/ref_legacy_shim.cpp" -o "$OUT/libsora_reflegacy.so" &
PID_LEGACY=$!
# ---- the 11a receive graph with HIP bricks plugged into it through the reference's own CREATE_BRICK_FILTER (oracle/ref_graph_hip_shim.cpp; the product's entry points
#      are bound at run time, nothing of it is linked); the generated substitution list is kept beside the library for INTEGRATION.md / the test to quote
python3 "$HERE/ref_flatten.py" "$TMP/flat_hip" hip
"$CXX" -std=c++14 -O2 -U__OPTIMIZE__ -fPIC -shared -fvisibility=hidden \
    -fms-extensions -fms-compatibility -fms-compatibility-version=19.00 -fdelayed-template-parsing -fno-operator-names \
    -msse4.1 -mssse3 -Wno-everything -DUSER_MODE -D__XSAVEINTRIN_H -include "$HERE/ref_compat.h" \
    -I"$TMP/flat_hip" -I"$HERE" "$HERE/ref_graph_hip_shim.cpp" -o "$OUT/libsora_refgraph_hip.so" &
PID_HIP=$!
cp "$TMP/flat_hip/fb11ademod_config_hip.diff" "$OUT/fb11ademod_config_hip.diff"
wait $PID_KERNELS; echo "build_ref.sh: built $OUT/libsora_ref.so"          # the three compiles run side by side; set -e stops on the first failure
wait $PID_GRAPH;   echo "build_ref.sh: built $OUT/libsora_refgraph.so"
wait $PID_MT;      echo "build_ref.sh: built $OUT/libsora_refgraph_mt.so"
wait $PID_LEGACY;  echo "build_ref.sh: built $OUT/libsora_reflegacy.so"
wait $PID_HIP;     echo "build_ref.sh: built $OUT/libsora_refgraph_hip.so"
