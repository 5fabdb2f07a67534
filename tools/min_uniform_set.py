import re, subprocess, sys, os
src = open('/root/repo/sora_amd/csrc/k_rx11b.hip').read()
a = src.index("#define U(v)"); b = src.index("#undef U")
block = src[a:b]
vars_ = re.findall(r"U\((\w+)\);", block)
def count(keep):
    body = "#define U(v) v = (decltype(v))uni((int)v)\n        " + " ".join("U(%s);" % v for v in keep) + "\n"
    s = src[:a] + body + src[b:]
    open('/tmp/k11b_try.hip', 'w').write(s)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-S", "-emit-llvm", "--cuda-device-only",
                    "-I/root/repo/sora_amd/csrc", "/tmp/k11b_try.hip", "-o", "/tmp/k11b_try.ll"], stderr=subprocess.DEVNULL, check=True)
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/opt", "-passes=print<uniformity>", "-disable-output", "/tmp/k11b_try.ll"], capture_output=True, text=True)
    return (out.stdout + out.stderr).count("DIVERGENT")
base = count(vars_); print("all", len(vars_), base, flush=True)
keep = list(vars_)
for v in vars_:
    trial = [k for k in keep if k != v]
    c = count(trial)
    if c <= base + 5:
        keep = trial
    print(v, c, "dropped" if v not in keep else "KEPT", flush=True)
print("minimal:", keep, count(keep))
