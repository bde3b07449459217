"""Build the in-tree native library cafe_amd/lib/libcafehip.so for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container and on the GPU box.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcafehip.so")

# translation units of libcafehip.so: compiled in parallel into cafe_amd/lib/obj/, relinked when any object changes
SOURCES = ["cafehip.hip", "cafehip_comm.hip", "k1_matrices.hip", "k2_walk16.hip", "k2_walk16o.hip", "k2_walk4.hip", "k2_walk4o.hip", "k2_walk4s.hip", "k2c_tables.hip", "k2c_gemm.hip", "k_misc.hip",
           os.path.join("host", "cafe_host.cpp")]
# per-unit flags: k2c_gemm keeps its accumulators in VGPRs (with 256 registers per lane the compiler otherwise shuttles them
# between AGPRs inside the chunk loop and VGPRs across its back edge: 128 moves per chunk)
UNIT_FLAGS = {"k2c_gemm.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
HEADERS = ["k2c_gemm.hpp", "context.hpp", "matrix_store.hpp", "compression_plan.hpp", "k2_launch.hpp", "k3_device.hpp", "exp_like_host.hpp", "exp_like_host_table.inc", "device_types.hpp", "kernels.hpp", "comm.hpp", "k2_mfma.hpp", "host_math.hpp", "schedule.hpp",
           os.path.join("host", "tree_table.hpp"), os.path.join("host", "nelder_mead.hpp"), os.path.join("host", "glibc_rand.hpp"),
           os.path.join("host", "poisson_prior.hpp"), os.path.join("host", "host_util.hpp"),
           os.path.join("..", "..", "include", "cafehip.h"), os.path.join("..", "..", "include", "cafehost.h")]
OBJDIR = os.path.join(LIBDIR, "obj")
PROBE_LIB = os.path.join(LIBDIR, "libcafeprobe.so")   # measured HBM / MFMA ceilings for bench.py (csrc/probe.hip)
BINDIR = os.path.join(HERE, "bin")
CLI = os.path.join(BINDIR, "cafehip")
DEPS = SOURCES + HEADERS + [os.path.join("host", "main.cpp")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result"]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cafe_amd needs the ROCm toolchain to build its HIP library")


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(CLI) or not os.path.exists(PROBE_LIB):
        return True
    t = os.path.getmtime(LIB)
    if os.path.getmtime(os.path.join(CSRC, "probe.hip")) > os.path.getmtime(PROBE_LIB):
        return True
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    # debugging / A-B runs: CAFEHIP_LIB=<path> loads a variant library built by tools/build_variant.py
    override = os.environ.get("CAFEHIP_LIB")
    if override:
        if not os.path.exists(override):
            raise RuntimeError("CAFEHIP_LIB=%s does not exist" % override)
        return override
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    newest_header = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    jobs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, os.path.basename(src).rsplit(".", 1)[0] + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(path), newest_header):
            jobs.append([hipcc()] + CFLAGS + UNIT_FLAGS.get(src, []) + os.environ.get("CAFEHIP_EXTRA_CFLAGS", "").split() + ["-c", "-o", obj, path])
    if verbose:
        for j in jobs:
            print(" ".join(j))
    procs = [subprocess.Popen(j) for j in jobs]   # the units are independent: compile side by side
    failed = [j for j, pr in zip(jobs, procs) if pr.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed: " + " ".join(failed[0]))
    objs = [os.path.join(OBJDIR, os.path.basename(src).rsplit(".", 1)[0] + ".o") for src in SOURCES]
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    # command-line front end (host C++ only), linked against the in-tree library
    os.makedirs(BINDIR, exist_ok=True)
    cli = [hipcc(), "-O2", "-std=c++17", "-o", CLI, os.path.join(CSRC, "host", "main.cpp"), "-L" + LIBDIR, "-lcafehip",
           "-Wl,-rpath,$ORIGIN/../lib"]
    if verbose:
        print(" ".join(cli))
    subprocess.check_call(cli)
    probe_src = os.path.join(CSRC, "probe.hip")
    if force or not os.path.exists(PROBE_LIB) or os.path.getmtime(PROBE_LIB) < os.path.getmtime(probe_src):
        probe = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", "-o", PROBE_LIB,
                 probe_src]
        if verbose:
            print(" ".join(probe))
        subprocess.check_call(probe)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
