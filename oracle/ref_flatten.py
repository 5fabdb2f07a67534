#!/usr/bin/env python3
"""ref_flatten.py <scratch-dir> -- TEST INFRASTRUCTURE ONLY (used by oracle/build_ref.sh).

Lays the reference's BRICK headers (kernel/core/inc, kernel/brick/inc, kernel/bb/Brick11/src, kernel/bb/demod11,
kernel/inc) out flat in a SCRATCH directory and patches, there, the constructs that only MSVC accepts, so that clang in
-fms-compatibility mode can compile the reference's own demodulation / modulation graphs on Linux.  Nothing is written
into this repository: the scratch directory is deleted by build_ref.sh after the compile, the only product is
oracle/_ref/libsora_refgraph.so (git-ignored).  Every patch below names what it works around; none changes arithmetic.
"""
import glob
import os
import re
import shutil
import sys

REF = os.environ.get("SORA_REFERENCE", "/root/reference") + "/kernel"
OUT = sys.argv[1]
KEEP_SEPARATOR_11A = len(sys.argv) > 2 and sys.argv[2] == "mt"     # libsora_refgraph_mt.so: the 11a graph with its real thread boundary
LEGACY = len(sys.argv) > 2 and sys.argv[2] == "legacy"            # libsora_reflegacy.so: the legacy dot11a C receiver (kernel/bb/dot11a), SURVEY section 8 f4
HIPBRICKS = len(sys.argv) > 2 and sys.argv[2] == "hip"            # libsora_refgraph_hip.so: the reference's 11a receive graph with HIP bricks plugged into it (oracle/ref_graph_hip_shim.cpp)

shutil.rmtree(OUT, ignore_errors=True)
os.makedirs(OUT + "/bb")
for d in ("core/inc", "brick/inc", "bb/Brick11/src", "bb/demod11", "inc"):
    for f in glob.glob(f"{REF}/{d}/*.h") + glob.glob(f"{REF}/{d}/*.hpp"):
        shutil.copy(f, OUT + "/" + os.path.basename(f))
for f in glob.glob(f"{REF}/inc/bb/*.h"):
    shutil.copy(f, OUT + "/bb/" + os.path.basename(f))
if LEGACY:
    # kernel/bb/dot11a: the sources of the legacy receiver (dot11/*.c, mod/*.c), its headers (inc/bb/mod.h, inc/bb/mod/*.h, inc/lut.h) and the
    # static look-up tables it links (lutst/*.c), laid out so that both spellings of its includes resolve ("bb/mod/x.h", "../inc/bb/mod.h")
    D = REF + "/bb/dot11a"
    for sub in ("bb/mod", "inc/bb/mod", "dot11", "lutst"):
        os.makedirs(OUT + "/" + sub, exist_ok=True)
    for f in glob.glob(D + "/inc/bb/mod/*.h"):
        shutil.copy(f, OUT + "/bb/mod/" + os.path.basename(f)); shutil.copy(f, OUT + "/inc/bb/mod/" + os.path.basename(f))
    shutil.copy(D + "/inc/bb/mod.h", OUT + "/bb/mod.h"); shutil.copy(D + "/inc/bb/mod.h", OUT + "/inc/bb/mod.h"); shutil.copy(D + "/inc/lut.h", OUT + "/lut.h")
    for f in glob.glob(D + "/dot11/*.c") + glob.glob(D + "/dot11/*.h") + glob.glob(D + "/mod/*.c"):
        shutil.copy(f, OUT + "/dot11/" + os.path.basename(f))
    for f in glob.glob(D + "/lutst/*.c"):
        shutil.copy(f, OUT + "/lutst/" + os.path.basename(f))


# ---- the Windows integer model (LLP64): the keyword `long` is 32 bits for the compiler the reference was written for, 64 here.
#      The typedef'd spellings (ULONG, ulong, LONG) come from ref_compat.h; the literal keyword is respelt in the scratch copy
#      (`long long` stays).  It matters: demap_dqpsk_bits (soradsp.h:190-198) computes `(UCHAR)((unsigned long)(re+im) >> 31) << pos`,
#      which is 1 << pos with a 32-bit long and 0xFF << pos with a 64-bit one -- the CCK decoders (cck.hpp) decode garbage otherwise.
_LONG = re.compile(r"(?<![A-Za-z0-9_])long(?![A-Za-z0-9_])")
def llp64(text):
    out = []
    for line in text.split("\n"):
        if "long" in line and "long long" not in line and not line.lstrip().startswith(("//", "*", "/*")):
            line = _LONG.sub("int", line)
        out.append(line)
    return "\n".join(out)
for f in sorted(glob.glob(OUT + "/*.h") + glob.glob(OUT + "/*.hpp") + glob.glob(OUT + "/bb/*.h")):
    s = open(f, encoding="latin-1").read()
    t = llp64(s)
    if t != s:
        with open(f, "w", encoding="latin-1") as fh:
            fh.write(t)

def write(name, text, mode="w"):
    with open(OUT + "/" + name, mode, encoding="latin-1") as fh:
        fh.write(text)


def edit(name, fn):
    p = OUT + "/" + name
    s = open(p, encoding="latin-1").read()
    t = fn(s)
    if t == s:
        raise SystemExit("ref_flatten.py: patch had no effect on " + name + " (reference layout changed?)")
    write(name, t)


# ---- Windows SDK / WDK headers the sources include by name: empty stand-ins (types come from ref_compat.h)
for h in ("windows.h", "Windows.h", "winerror.h", "guiddef.h", "process.h", "windef.h"):
    write(h, "#pragma once\n")
write("typeinfo.h", "#pragma once\n#include <typeinfo>\n")
write("new.h", "#pragma once\n#include <new>\n")

# ---- vector128.h: hand re-declared intrinsics (lines 26-81) and two wrappers of intrinsics that do not exist
def v128(s):
    lines = s.split("\n")
    del lines[25:81]
    s = "\n".join(lines).replace("#include <emmintrin.h>", "#include <immintrin.h>")
    return "\n".join(l for l in s.split("\n") if "_mm_sign_epi64" not in l and "_mm_abs_epi64" not in l)
edit("vector128.h", v128)

# ---- driver-side umbrella headers: the offline graphs need only the arithmetic headers behind them
write("sora.h", '#pragma once\n#include "const.h"\n#include "complex.h"\n#include "vector128.h"\n#include "func.h"\n'
                "typedef struct _SORA_RADIO_RX_STREAM SORA_RADIO_RX_STREAM, *PSORA_RADIO_RX_STREAM;\n")
write("soratypes.h", '#pragma once\n#include "const.h"\n#include "complex.h"\n'
                     "typedef COMPLEX16 SAMPLE, *PSAMPLE, RXSAMPLE, *PRXSAMPLE; typedef COMPLEX8 TXSAMPLE, *PTXSAMPLE;\n"
                     "typedef char FLAG, *PFLAG;\n#define MAX_RADIO_NUMBER 8\n")
write("brickutil.h", "#pragma once\n")           # dump-file loader (the shim gets samples from its caller)
write("soratime.h", "#pragma once\nstruct SoraStopwatch { SoraStopwatch(bool = false) {} void Restart() {} void Stop() {} void Start() {} void Reset() {} };\n")
write("MACStopwatch.h", "#pragma once\nstruct MACStopwatch { template<class... A> void LeaveRX(A...) {} template<class... A> void EnterRX(A...) {}"
                        " template<class... A> void LeaveCS(A...) {} template<class... A> void EnterCS(A...) {} void OutputStats() {} };\n")
edit("dspcomm.h", lambda s: s.replace("typedef unsigned long\tulong;", "").replace("typedef COMPLEX8        TXSAMPLE;", "")
     .replace("typedef COMPLEX16       RXSAMPLE;", ""))
edit("stdbrick.hpp", lambda s: s.replace("#include <rxstream.hpp>", ""))      # live radio source
edit("bb/bba.h", lambda s: s.replace('"../../brick/inc/bb_debug.h"', '"bb_debug.h"'))

# ---- brick.h: MSVC lets trailing macro arguments be omitted, accepts "(TYPE) (&NAME)" declarators and binds the
#      base-class names of templates late (IControlPoint / DummyBrickInstance are used before they are declared)
def brick(s):
    for m in ("FACADE_FIELD", "REFERENCE_SHARED_VAR", "DEFINE_SHARED_VAR", "CTX_VAR_RW", "CTX_VAR_RO"):
        s = s.replace("#define %s(TYPE, NAME, DIMENSIONS)" % m, "#define %s(TYPE, NAME, ...)" % m)
    s = (s.replace("__##NAME##__ DIMENSIONS;", "__##NAME##__ __VA_ARGS__;")
          .replace("(&NAME()) DIMENSIONS {", "(&NAME()) __VA_ARGS__ {")
          .replace("shared_var_reference<TYPE DIMENSIONS> NAME;", "shared_var_reference<TYPE __VA_ARGS__> NAME;")
          .replace("TYPE NAME DIMENSIONS;", "TYPE NAME __VA_ARGS__;")
          .replace("(TYPE) (&NAME) DIMENSIONS;", "TYPE (&NAME) __VA_ARGS__;")
          .replace("(TYPE) const (&NAME) DIMENSIONS;", "TYPE const (&NAME) __VA_ARGS__;"))
    body = re.search(r"struct IControlPoint : public IQueryable\s*\{.*?\};\s*", s, flags=re.S).group(0)
    s = s.replace(body, "").replace("// Brick - Sink", body + "\n// Brick - Sink", 1)
    return s.replace('#include "demux.h"', "", 1) + '\n#include "demux.h"\n'
edit("brick.h", brick)
# ---- pinqueue.h: members of a dependent base named at class scope
edit("pinqueue.h", lambda s: s.replace("[nstream][qsize]", "[NSTREAM][lcm<N,M>::value]")
     .replace("[NSTREAM][lcm<N,M>::value], unsigned int cnt)", "[NSTREAM][N], unsigned int cnt)"))

# ---- the 11a receive graph: the hop to the decoder thread becomes a same-thread pass-through (TNoInline swallows the
#      sink's "stop" like the thread boundary does).  This is the deterministic limit of the two-thread harness -- an
#      infinitely fast ViterbiThread -- which is also what oracle/so_rx11a.c and the GPU path implement.
if not KEEP_SEPARATOR_11A:
    edit("fb11ademod_config.hpp", lambda s: s.replace("TThreadSeparator<>::Filter", "TNoInline").replace("srcViterbi = vit0;", "srcViterbi = NULL;"))
if HIPBRICKS:
    # The drop-in, as a maintainer would make it: copies of the reference's own CreateDemodGraph11a_40M in which only the brick NAMES on
    # three kinds of CREATE_BRICK_FILTER lines change.  Generated here from the reference's text -- nothing of it is kept in this repository.
    cfg = open(OUT + "/fb11ademod_config.hpp", encoding="latin-1").read()
    m = re.search(r"static inline\s*\nvoid CreateDemodGraph11a_40M \(.*?\n\}\n", cfg, flags=re.S)
    if not m:
        raise SystemExit("ref_flatten.py: CreateDemodGraph11a_40M not found (reference layout changed?)")
    body = m.group(0)
    def variant(name, subs):
        t = body.replace("CreateDemodGraph11a_40M", name)
        for a, b in subs:
            if a not in t:
                raise SystemExit("ref_flatten.py: '%s' not in CreateDemodGraph11a_40M (reference layout changed?)" % a)
            t = t.replace(a, b)
        return t
    fft = [("TFFT64, BB11aDemodCtx", "THipFFT64, BB11aDemodCtx")]
    dmd = [("T11aDemapBPSK::Filter", "THip11aDemap<1>::Filter"), ("T11aDemapQPSK::Filter", "THip11aDemap<2>::Filter"), ("T11aDemapQAM16::Filter", "THip11aDemap<4>::Filter"),
           ("T11aDemapQAM64::Filter", "THip11aDemap<6>::Filter"),
           ("T11aDeinterleaveBPSK, BB11aDemodCtx", "THip11aDeinterleave<1>::Filter, BB11aDemodCtx"), ("T11aDeinterleaveQPSK, BB11aDemodCtx", "THip11aDeinterleave<2>::Filter, BB11aDemodCtx"),
           ("T11aDeinterleaveQAM16, BB11aDemodCtx", "THip11aDeinterleave<4>::Filter, BB11aDemodCtx"), ("T11aDeinterleaveQAM64, BB11aDemodCtx", "THip11aDeinterleave<6>::Filter, BB11aDemodCtx")]
    # round 6: the five bricks that work on the context facades -- the decoder (viterbi), the tracker (pilot) and the three one-multiply bricks (pcomp, chequ, fcomp)
    ctxb = [("typedef T11aViterbi <5000*8, 48, 256> T11aViterbiComm;", "typedef THip11aViterbi <5000*8, 48, 256> T11aViterbiComm;"), ("TPilotTrack, BB11aDemodCtx", "THip11aPilotTrack, BB11aDemodCtx"),
            ("TPhaseCompensate, BB11aDemodCtx", "THipPhaseCompensate, BB11aDemodCtx"), ("TChannelEqualization, BB11aDemodCtx", "THipChannelEqualization, BB11aDemodCtx"),
            ("TFreqCompensation, BB11aDemodCtx", "THipFreqCompensation, BB11aDemodCtx")]
    write("fb11ademod_config_hip.hpp", "#pragma once\n" + variant("CreateDemodGraph11a_40M_HipFFT", fft) + "\n" + variant("CreateDemodGraph11a_40M_HipFFTDemapDeint", fft + dmd)
          + "\n" + variant("CreateDemodGraph11a_40M_HipAll", fft + dmd + ctxb))
    diff = ["--- kernel/bb/demod11/fb11ademod_config.hpp (CreateDemodGraph11a_40M)", "+++ the same with HIP bricks"]
    for a, b in fft + dmd + ctxb:
        for line in body.split("\n"):
            if a in line:
                diff += ["-" + line.strip(), "+" + line.strip().replace(a, b)]
    write("fb11ademod_config_hip.diff", "\n".join(diff) + "\n")

# ---- mapper11a.hpp: an array bound that this clang's declaration/expression disambiguation trips over (same value)
edit("mapper11a.hpp", lambda s: s.replace("(&lut)[intpow<2, LUT_BITS>::value][LUT_BITS/M/2]", "(&lut)[(1 << LUT_BITS)][LUT_BITS/M/2]"))

# ---- 11b graphs: bbb.h drags in the live RX-stream helpers; Windows file systems ignore the case of include names
write("rxstream.h", "#pragma once\n")
for lower, real in (("phy_11b.hpp", "PHY_11b.hpp"), ("phy_11a.hpp", "PHY_11a.hpp"), ("phy_11n.hpp", "PHY_11n.hpp")):
    if os.path.exists(OUT + "/" + real) and not os.path.exists(OUT + "/" + lower):
        write(lower, '#pragma once\n#include "%s"\n' % real)

# ---- 11n graphs: casts of an rvalue to a reference (an MSVC extension) go through a named temporary
_RV = re.compile(r"c = \((\w+)&\)(_mm_shuffle_p[sd]\(.*\)); return c;")
for h in ("sora_matrix.h", "vector128.h"):
    edit(h, lambda s: _RV.sub(r"{ auto t_ = \2; c = (\1&)t_; } return c;", s))
# ---- the 11n receive graph: same thread-hop substitution as the 11a graph
edit("fb11ndemod_config.hpp", lambda s: s.replace("TThreadSeparator<>::Filter", "TNoInline").replace("srcViterbi = vit0;", "srcViterbi = NULL;"))
# ---- stdbrick.hpp: a temporary bound to a non-const reference parameter (an MSVC extension)
edit("stdbrick.hpp", lambda s: s.replace("Next0()->Process(ipin.clone());", "{ auto c_ = ipin.clone(); Next0()->Process(c_); }"))
# ---- brick.h: TraverseGraph finds bricks by MSVC's spelling of typeid names ("class Name<...>")
edit("brick.h", lambda s: s.replace("        const char *self = typeid(*this).name();",
     '        char self[2048]; { int st_; char* d_ = abi::__cxa_demangle(typeid(*this).name(), 0, 0, &st_);'
     ' snprintf(self, sizeof(self), "class %s", d_ ? d_ : ""); free(d_); }'))

if LEGACY:
    write("timing.h", "#pragma once\n")                                   # bba.h: "../timing.h" (the TIMINGINFO stop-watch: oracle/ref_legacy_shim.cpp)
    for d in ("bb/mod", "inc/bb/mod"):
        # a path that climbs out of the tree it was written in
        edit(d + "/afreq.h", lambda s: s.replace('"../../../../../brick/inc/bb_debug.h"', '"bb_debug.h"')
             # a temporary bound to a non-const reference (an MSVC extension)
             .replace("vcs m2 = (vcs)mul_high((vs&)flip(a), sin_nsin);", "vcs fl_ = flip(a); vcs m2 = (vcs)mul_high((vs&)fl_, sin_nsin);"))
        # SoraRadioReadRxStream (core/inc/rxstream.h:8-37) belongs to the radio manager; the offline harness's version of it is in the shim
        edit(d + "/fetchdt.h", lambda s: s.replace('#include "44MTo40M.h"', '#include "44MTo40M.h"\n#include "ref_legacy_rxstream.h"'))
