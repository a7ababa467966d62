"""Build recipe for the native libraries (all in-tree; the .so files travel with the snapshot).

  libokrt.so       hipcc --offload-arch=gfx950   csrc/*.hip          the HIP product
  libokrt_host.so  g++                           host/*.cpp          C++ host layer
  oracle/liboracle.so (+ oracle/_ref when /root/reference exists)    test infrastructure
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(PKG)

HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-value"]
HOST_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-pthread", "-Wall", "-Wno-class-memaccess"]
DEVICE_LINK = ["-ldl"]  # okrt_mgpu.hip binds RCCL with dlopen


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd, cwd=None):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, cwd=cwd, check=True)


def build_device(force=False):
    """One object per source (cached under _build/, compiled in parallel), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    src = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")) + glob.glob(os.path.join(PKG, "csrc", "*.cpp")))
    headers = glob.glob(os.path.join(PKG, "csrc", "*.h")) + [os.path.join(REPO, "include", "okrt.h")]
    out = os.path.join(PKG, "libokrt.so")
    objdir = os.path.join(PKG, "_build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [f for f in HIP_FLAGS if f != "-shared"]
    jobs = []
    for f in src:
        obj = os.path.join(objdir, os.path.basename(f) + ".o")
        if force or _newer(obj, [f] + headers):
            jobs.append([hipcc] + flags + ["-c", f, "-o", obj])
    with ThreadPoolExecutor(max_workers=4) as pool:
        list(pool.map(_run, jobs))
    objs = [os.path.join(objdir, os.path.basename(f) + ".o") for f in src]
    if force or jobs or _newer(out, objs):
        _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + DEVICE_LINK)
    return out


def build_host(force=False):
    src = sorted(glob.glob(os.path.join(PKG, "host", "*.cpp")))
    deps = src + glob.glob(os.path.join(PKG, "host", "*.h")) + [os.path.join(REPO, "include", "okrt.h")]
    out = os.path.join(PKG, "libokrt_host.so")
    if force or _newer(out, deps):
        _run([os.environ.get("CXX", "g++")] + HOST_FLAGS + ["-o", out] + src)
    return out


def build_oracle():
    _run(["make", "-C", os.path.join(REPO, "oracle"), "all"])
    if os.path.isdir("/root/reference/source"):
        _run(["make", "-C", os.path.join(REPO, "oracle"), "ref"])


def build_examples(force=False):
    """examples/: C++ hosts linked against the host layer and the device layer (no Python in between)."""
    outs = []
    for name, extra in (("headless_host", []), ("headless_host_mgpu", ["-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__"])):
        src = os.path.join(REPO, "examples", name + ".cpp")
        out = os.path.join(REPO, "examples", name)
        deps = [src, os.path.join(PKG, "libokrt.so"), os.path.join(PKG, "libokrt_host.so")] + glob.glob(os.path.join(PKG, "host", "*.h"))
        if force or _newer(out, deps):
            _run([os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-ffp-contract=off", src] + extra + ["-I" + os.path.join(REPO, "include"),
                  "-I" + os.path.join(PKG, "host"), "-L" + PKG, "-lokrt_host", "-lokrt", "-L/opt/rocm/lib", "-lamdhip64",
                  "-Wl,-rpath,$ORIGIN/../gpu-raytracer_amd", "-Wl,-rpath,/opt/rocm/lib", "-o", out])
        outs.append(out)
    return outs


def build_all(force=False):
    build_device(force)
    build_host(force)
    build_examples(force)
    build_oracle()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
