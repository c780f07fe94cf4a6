#!/usr/bin/env python3
"""Why is the FIRST execute of a freshly built batch slower than the steady state (C3 whole: 9.3 ms against 7.4)?

Every scenario builds the cohort's image on the device (v2p_batch_build_on_device) and then times `--execs` executes one by one
(HIP events on the launch stream), varying what the GPU did just before:

    cold            the host sleeps 1 s before the build (what bench.py's one_shot sees: the stream's H2D ran, the shader engines idled)
    busy_before     a compute-bound torch loop keeps the GPU busy for ~150 ms right before the build
    busy_between    build, then ~60 ms of the busy loop, then the executes
    touch_arena     build, then the whole arena is written once by a fill kernel (hipMemset), then the executes
    idle_between    build, the host sleeps 0.3 s, then the executes
    re_idle         (same batch as the last scenario) after the executes the host sleeps 0.5 s and executes again: a clock ramp comes back,
                    a first-touch cost does not
A probe (a fixed compute-bound kernel, ~0.1 ms, timed the same way) is launched before and after every step: its duration follows the
shader clock.  The amdgpu sysfs clock level is sampled too when the box lets an ordinary user read it.

    python tools/first_execute.py [--workload C3 --samples 10000] > gpurun_out/first_execute.json
"""
import argparse
import glob
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sclk():
    """the amdgpu driver's current clock levels (shader, memory, fabric, SoC), where an ordinary user may read them"""
    out = {}
    for name in ("sclk", "mclk", "fclk", "socclk"):
        for p in glob.glob(f"/sys/class/drm/card*/device/pp_dpm_{name}")[:1]:
            try:
                cur = [ln.strip() for ln in open(p).read().splitlines() if "*" in ln]
                out[name] = cur[0] if cur else None
            except Exception as e:          # noqa: BLE001
                out[name] = f"unreadable: {e!r}"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C3")
    ap.add_argument("--samples", type=int, default=10000)
    ap.add_argument("--execs", type=int, default=8)
    a = ap.parse_args()
    import torch
    from vcf2prot_amd import build
    build.build_hip(); build.build_cohort()
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.engine import Context
    from vcf2prot_amd.txstream import build_on_device_auto
    nt = min(64, os.cpu_count() or 1)
    cohort = Cohort.preset(a.workload, n_samples=a.samples)
    n = cohort.n_haplotypes
    stream = cohort.txstream(0, n, n_threads=nt)
    rb = int(cohort.result_sizes(0, n, n_threads=nt).sum())
    ts = torch.cuda.Stream()
    x = torch.randn(2048, 2048, device="cuda", dtype=torch.float32)
    px = torch.randn(1024, 1024, device="cuda", dtype=torch.float32)

    def ev():
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def probe():
        """a fixed compute-bound piece of work on the launch stream: ms follows the shader clock"""
        e0, e1 = ev()
        with torch.cuda.stream(ts):
            e0.record(ts)
            y = px
            for _ in range(4):
                y = torch.sin(y) * 1.0001
            e1.record(ts)
        e1.synchronize()
        return e0.elapsed_time(e1)

    def busy(ms):
        t0 = time.perf_counter()
        with torch.cuda.stream(ts):
            while (time.perf_counter() - t0) * 1e3 < ms:
                y = x
                for _ in range(8):
                    y = torch.sin(y) * 1.0001
                ts.synchronize()

    out = {"workload": a.workload, "samples": a.samples, "haplotypes": n, "result_bytes": rb, "scenarios": {}}
    with Context(0) as ctx:
        ctx.upload_proteome(cohort.proteome())
        ctx.set_stream(ts.cuda_stream)
        for _ in range(3):
            probe()
        # hipMalloc / hipFree of an arena-sized buffer: what a one-shot call pays when it does not recycle its arena
        t0 = time.perf_counter(); big = torch.empty(rb, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); t_alloc = time.perf_counter() - t0
        del big
        t0 = time.perf_counter(); torch.cuda.empty_cache(); torch.cuda.synchronize(); t_free = time.perf_counter() - t0
        out["arena_alloc_ms"], out["arena_free_ms"] = t_alloc * 1e3, t_free * 1e3

        def executes(b, k):
            res = []
            for _ in range(k):
                p0 = probe()
                e0, e1 = ev()
                e0.record(ts); b.execute(); e1.record(ts); b.sync()
                res.append({"probe_before_ms": p0, "execute_ms": e0.elapsed_time(e1)})
            res.append({"probe_after_ms": probe()})
            return res

        def scenario(name, before=None, between=None):
            rec = {"sclk_start": sclk()}
            if before:
                before()
            rec["probe_before_build_ms"] = probe() if name != "cold_noprobe" else None
            b = ctx.batch()
            t0 = time.perf_counter()
            info = build_on_device_auto(b, stream, rb)
            rec["build_call_s"] = time.perf_counter() - t0
            rec["build_kernels_ms"] = info["build_ms"]
            rec["sclk_after_build"] = sclk()
            if between:
                between(b)
            rec["executes"] = executes(b, a.execs)
            rec["sclk_end"] = sclk()
            out["scenarios"][name] = rec
            return b

        for rep in range(2):
            scenario(f"cold_{rep}", before=lambda: time.sleep(1.0)).close()
            torch.cuda.empty_cache()
            scenario(f"busy_before_{rep}", before=lambda: busy(150)).close()
            torch.cuda.empty_cache()
        scenario("busy_between", between=lambda b: busy(60)).close()
        torch.cuda.empty_cache()

        def touch(b):
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
            hip.hipMemsetAsync(ctypes.c_void_p(b.device_out()), 0x2E, rb, ctypes.c_void_p(ts.cuda_stream))
            ts.synchronize()
        scenario("touch_arena", before=lambda: time.sleep(1.0), between=touch).close()
        torch.cuda.empty_cache()
        b = scenario("idle_between", before=lambda: time.sleep(1.0), between=lambda b: time.sleep(0.3))
        time.sleep(0.5)
        out["scenarios"]["re_idle_same_batch_after_0.5s"] = {"executes": executes(b, a.execs)}
        busy(100)
        out["scenarios"]["same_batch_after_busy"] = {"executes": executes(b, a.execs)}
        b.digests()
        t0 = time.perf_counter(); b.digests(); out["digests_call_ms"] = (time.perf_counter() - t0) * 1e3
        b.close()
        ctx.set_stream(0)
    stream.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
