"""Build recipe for the native parts (hipcc cross-compiles gfx950 without a GPU).

    python -m vcf2prot_amd.build          # build everything that is out of date
    python -m vcf2prot_amd.build --force

Outputs (git-ignored, shipped to the GPU box by gpurun):
    vcf2prot_amd/lib/libvcf2prot_hip.so   HIP kernels + C ABI (include/vcf2prot_hip.h)
    vcf2prot_amd/lib/libv2p_cohort.so     synthetic cohort generator (include/v2p_cohort.h), plain C++
"""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
ARCH = "gfx950"

HIP_LIB = os.path.join(LIBDIR, "libvcf2prot_hip.so")
COHORT_LIB = os.path.join(LIBDIR, "libv2p_cohort.so")

HIP_SOURCES = ["stitch_kernels.hip", "stitch_wave.hip", "build_kernels.hip", "build_rows.hip", "dense_pieces.hip", "v2p_api.hip", "decode_kernels.hip", "v2p_decode_api.hip"]
HIP_DEPS = HIP_SOURCES + ["rows_image.hpp", "build_rows.h", "patch_image.h", "dense_pieces.h", "patch_format.hpp", "stitch_kernels.h", "stitch_device.hpp", "build_kernels.h", "decode_kernels.h", "v2p_ctx_internal.h", "sir_pack.hpp",
                          os.path.join(ROOT, "include", "vcf2prot_hip.h"), os.path.join(ROOT, "include", "v2p_frontend.h")]
COHORT_SOURCES = ["cohort_gen.cpp", os.path.join("host", "transcript_tasks.cpp"), os.path.join("host", "vcf_index.cpp"),
                  os.path.join("host", "group_muts.cpp"), os.path.join("host", "instructions.cpp")]
COHORT_DEPS = COHORT_SOURCES + ["sir_pack.hpp", "rows_image.hpp", "patch_image_host.hpp", "patch_format.hpp", os.path.join(ROOT, "include", "v2p_cohort.h"),
                                os.path.join(ROOT, "include", "v2p_step4b.h"), os.path.join(ROOT, "include", "v2p_frontend.h"),
                                os.path.join(ROOT, "include", "v2p_step4a.h"), os.path.join("host", "frontend_common.hpp")]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    for d in deps:
        p = d if os.path.isabs(d) else os.path.join(CSRC, d)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def _hipcc() -> str:
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.isabs(c) and os.path.exists(c):
            return c
    return "hipcc"


def _compile_objects(sources, deps, objdir, extra, force, verbose):
    """One object per source, compiled in parallel (hipcc takes most of a minute for the larger files); an object is rebuilt when its
    source or any header / dependency is newer."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(objdir, exist_ok=True)
    headers = [d for d in deps if d not in sources]
    jobs, objs = [], []
    for src in sources:
        obj = os.path.join(objdir, os.path.basename(src).rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([_hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-result", *extra,
                         "-c", os.path.join(CSRC, src), "-o", obj])
    if jobs:
        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(min(len(jobs), max(1, (os.cpu_count() or 2) - 1))) as pool:
            list(pool.map(run, jobs))
    return objs, bool(jobs)


def build_hip(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    if force or _stale(HIP_LIB, HIP_DEPS):
        objs, _ = _compile_objects(HIP_SOURCES, HIP_DEPS, os.path.join(LIBDIR, "obj"), [], force, verbose)
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-fPIC", "-shared", *objs, "-o", HIP_LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HIP_LIB


def build_cohort(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    if not all(os.path.exists(os.path.join(CSRC, s)) for s in COHORT_SOURCES):
        return ""
    if force or _stale(COHORT_LIB, COHORT_DEPS):
        cmd = ["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall",
               *[os.path.join(CSRC, s) for s in COHORT_SOURCES], "-o", COHORT_LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return COHORT_LIB


BENCH_LIB = os.path.join(LIBDIR, "libv2p_bench.so")
# (the development library also carries what no routing rule of the product picks: PATCH images, the grid builders of rounds 2-3, the
# A/B switches of the builders and launchers -- compiled in by V2P_BENCH_VARIANTS)
BENCH_SOURCES = HIP_SOURCES + ["patch_image.hip", os.path.join("bench", "bench_kernels.hip"), os.path.join("bench", "wave_copy_bench.hip")]
BENCH_DEPS = HIP_DEPS + [os.path.join("bench", "bench_kernels.hip"), os.path.join("bench", "wave_copy_bench.hip"), os.path.join("bench", "v2p_bench.h")]


def build_bench(force: bool = False, verbose: bool = False) -> str:
    """Development tools only (tools/*.py): the micro-benchmarks of csrc/bench/ and a V2P_BENCH_VARIANTS build of the engine that
    accepts the timing-only kernel ablations.  Nothing in the package or the tests loads it."""
    os.makedirs(LIBDIR, exist_ok=True)
    if force or _stale(BENCH_LIB, BENCH_DEPS):
        objs, _ = _compile_objects(BENCH_SOURCES, BENCH_DEPS, os.path.join(LIBDIR, "obj_bench"), ["-DV2P_BENCH_VARIANTS"], force, verbose)
        cmd = [_hipcc(), f"--offload-arch={ARCH}", "-fPIC", "-shared", *objs, "-o", BENCH_LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return BENCH_LIB


HARNESS_BIN = os.path.join(LIBDIR, "v2p_harness")
HARNESS_DEPS = [os.path.join("host", "v2p_harness.cpp"), os.path.join("host", "ppgg_gpu.hpp"),
                os.path.join(ROOT, "include", "vcf2prot_hip.h"), os.path.join(ROOT, "include", "v2p_cohort.h"),
                os.path.join(ROOT, "include", "v2p_frontend.h"), os.path.join(ROOT, "include", "v2p_step4a.h"),
                os.path.join(ROOT, "include", "v2p_step4b.h")]


def build_harness(force: bool = False, verbose: bool = False) -> str:
    """C++ host harness (the role of the reference's exec::execute) linked against both libraries."""
    if force or _stale(HARNESS_BIN, HARNESS_DEPS) or _stale(HARNESS_BIN, [HIP_LIB, COHORT_LIB]):
        cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-Wall", os.path.join(CSRC, "host", "v2p_harness.cpp"),
               "-L" + LIBDIR, "-lvcf2prot_hip", "-lv2p_cohort", "-lz", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + "/opt/rocm/lib",
               "-o", HARNESS_BIN]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return HARNESS_BIN


def build_all(force: bool = False, verbose: bool = False):
    """The product: engine library, host library, C++ harness.  (libv2p_bench.so -- development tools, kernel variants -- is built by
    build_bench(), lazily by _native.bench_lib(), and by __graft_entry__.build() so that it travels to the GPU box.)"""
    libs = build_hip(force, verbose), build_cohort(force, verbose)
    build_harness(force, verbose)
    return libs


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
    print(build_bench(force="--force" in sys.argv, verbose=True))
