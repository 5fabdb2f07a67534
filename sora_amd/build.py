"""Build libsora_hip.so (hand-written HIP for gfx950) in-tree with hipcc.  `python -m sora_amd.build`.

Every source is compiled to its own object (in parallel, only when it or a header changed) and the objects are linked
into sora_amd/lib/libsora_hip.so.  hipcc cross-compiles for gfx950 without a GPU."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "libsora_hip.so")
SOURCES = ["k_scan.hip", "k_rx.hip", "k_vit16.hip", "k_vitwin.hip", "k_stage.hip", "k_tx.hip", "k_rx11b.hip", "k_11n.hip", "k_rx11n.hip", "k_ht40.hip", "k_deliver.hip",
    "sora_hip.cpp", "sora_shard.cpp"]
# k_decode.hip (the data field in one kernel, sora_rx_set_fused) left the default library in round 4: build_variant("fused", ["SORA_WITH_K_DECODE"])
# compiles it in (the entry point answers SORA_E_NOT_SUPPORTED otherwise).
VARIANT_SOURCES = {"SORA_WITH_K_DECODE": ["k_decode.hip"]}
HEADERS = ["dev_arith.h", "dev_viterbi.h", "dev_vit16.h", "dev_winplan.h", "dev_vitwin.h", "dev_11n.h", "rx_types.h", "kernels.h", os.path.join("..", "..", "include", "sora_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-x", "hip"]   # hidden: the library exports what include/sora_hip.h declares, nothing else


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libsora_hip.so cannot be built (there is no CPU fallback)")


def _newest_header():
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)


def _obj(src):
    return os.path.join(OBJ, os.path.splitext(src)[0] + ".o")


def _stale(src, hdr_t):
    o = _obj(src)
    return not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(os.path.join(CSRC, src)), hdr_t)


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


INFO = os.path.join(HERE, "lib", "libsora_hip.buildinfo.json")


def sources_sha256():
    """One hash over every source and header the library is built from (what a build stamp is compared with)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES) + sorted(HEADERS):
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode()); h.update(fh.read())
    return h.hexdigest()


def build_info():
    """Where the library that is loaded came from (the .so is a git-ignored artefact that travels with the snapshot):
    the stamp build() wrote beside it, whether its source hash is the tree's, and whether it was built on this host."""
    import json, socket, hashlib
    out = {"library": os.path.relpath(LIB, os.path.dirname(HERE)), "exists": os.path.exists(LIB)}
    if out["exists"]:
        with open(LIB, "rb") as fh:
            out["library_sha256"] = hashlib.sha256(fh.read()).hexdigest()
    try:
        with open(INFO) as fh:
            st = json.load(fh)
    except Exception:
        st = None
    now = sources_sha256()
    out["stamp"] = st
    out["sources_sha256_now"] = now
    out["built_from_this_tree"] = bool(st) and st.get("sources_sha256") == now and st.get("library_sha256") == out.get("library_sha256")
    out["built_on_this_host"] = bool(st) and st.get("host") == socket.gethostname()
    return out


def _write_info(ncompiled):
    import json, socket, hashlib, time
    try:
        ver = subprocess.run([hipcc(), "--version"], capture_output=True, text=True).stdout.splitlines()[0]
    except Exception:
        ver = None
    with open(LIB, "rb") as fh:
        lib_sha = hashlib.sha256(fh.read()).hexdigest()
    with open(INFO, "w") as fh:
        json.dump({"sources_sha256": sources_sha256(), "library_sha256": lib_sha, "host": socket.gethostname(), "built_at_unix": int(time.time()), "hipcc": ver,
                   "flags": FLAGS, "objects_compiled_in_this_build": ncompiled}, fh, indent=1)


def build_variant(name, defines):
    """An experimental build of the whole library with extra -D flags -> sora_amd/lib/variants/<name>.so (A/B measurements:
    run any entry point with SORA_HIP_LIB=<that file>)."""
    out_dir = os.path.join(HERE, "lib", "variants"); os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, name + ".so")
    if any(d.startswith(("SORA_EXP_", "SORA_DBG_", "SORA_SCAN_PROBE")) for d in defines) and "SORA_TOOLS" not in defines:
        defines = list(defines) + ["SORA_TOOLS"]                                  # every experiment / probe switch lives in the tools variant only (kernels.h refuses otherwise)
    extra = [f for d in defines for f in VARIANT_SOURCES.get(d.split("=")[0], [])]
    cmd = [hipcc()] + FLAGS + ["-shared", "-w"] + ["-D" + d for d in defines] + [os.path.join(CSRC, f) for f in SOURCES + extra] + ["-ldl", "-o", out]
    subprocess.check_call(cmd)
    return out


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    cc = hipcc()
    hdr_t = _newest_header()
    todo = [s for s in SOURCES if force or _stale(s, hdr_t)]

    def compile_one(src):
        cmd = [cc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", _obj(src)]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr[-4000:]))

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(todo)))) as ex:
        list(ex.map(compile_one, todo))
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in SOURCES] + ["-ldl", "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    _write_info(len(todo))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
