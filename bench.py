#!/usr/bin/env python3
"""Benchmark of the MI355X SIR executor (vcf2prot step 6) -- driver contract in the task brief.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C2|C3|C4|C5] [--samples S]
                    [--scaling weak|strong]

One "step" = one pass of the hot path (stitch kernel: K0 fill + K1 in-chunk scan + K2
gather/scatter) over this rank's whole shard, inputs already resident in HBM.  At N=1 the
workload is BASELINE.json configs[1] ("C2": 1 000 samples x 20 k transcripts x ~400 aa, one
missense per transcript => 2 000 haplotypes, A = 1.6e10 residues, N = 1.2e8 tasks); the same
line carries `north_star_cohort`: configs[2] ("C3", the north star's 10 000-sample cohort,
20 000 haplotypes, 36 GB of result) whole, in one launch, every haplotype verified.

--gpus N > 1 launches itself: the parent spawns `python -m torch.distributed.run` with N ranks
(one per GPU, RCCL) before it touches any GPU, and exits with the children's code; under an
external launcher (RANK/WORLD_SIZE set) it just runs as a rank.  Its default is the north
star's run: --workload C3 --scaling strong.
  --scaling weak   every rank executes its own `--samples`-sample shard of a samples*N cohort (default at N = 1)
  --scaling strong one cohort of `--samples` samples (C3: the 10 000-sample cohort of the north star),
                   cut into contiguous haplotype ranges of equal result bytes (shard.shard_by_bytes);
                   rank 0 first times the whole cohort alone for `speedup_vs_1` (default at N > 1)
Haplotypes are independent: no data-path collective; the step's only exchange is the all-gather of
{haplotypes, result bytes} per rank.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (about 6.3 TB/s is achievable by a copy)

# samples per GPU (weak) / per cohort (strong)
DEFAULT_SAMPLES = {"weak": {"C2": 1000, "C3": 2000, "C4": 313, "C5": 10000},
                   "strong": {"C2": 1000, "C3": 10000, "C4": 2504, "C5": 50000}}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None, choices=["C2", "C3", "C4", "C5"], help="default: C2 at one GPU, C3 (the north star's cohort) at several")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"], help="default: weak at one GPU, strong at several")
    ap.add_argument("--no-north-star", action="store_true", help="N = 1: skip the north_star_cohort leg (C3 whole, 36 GB, one launch)")
    ap.add_argument("--no-speedup-ref", action="store_true", help="N > 1, strong scaling: do not time the whole cohort on rank 0 alone first")
    ap.add_argument("--samples", type=int, default=0, help="weak: samples per GPU; strong: samples of the whole cohort (0 = the config's own size)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the transfers-inclusive leg (v2p_pipeline_*)")
    ap.add_argument("--no-device-build", action="store_true", help="skip the device-side image build leg (v2p_batch_build_on_device)")
    ap.add_argument("--verify", default="all", choices=["all", "sample", "none"],
                    help="haplotypes whose digest is compared with the oracle before timing")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--temporal", action="store_true", help="plain result stores instead of non-temporal")
    ap.add_argument("--max-blocks", type=int, default=0)
    ap.add_argument("--fasta", action="store_true", help="FASTA-emitting image: headers and line feeds fused into the scatter (SURVEY 8f rank 1)")
    ap.add_argument("--var", type=int, default=0, help="K2 variant (0 = default, 1 = legacy byte-granular gathers)")
    ap.add_argument("--kernel", type=int, default=0, help="image routing: 0 = the packer's choice, 1 = long-run (stitch4_kernel), 2 = per block, 3 = dense, 4 = wave (stitchw_kernel)")
    ap.add_argument("--wpg", type=int, default=0, help="stitchw_kernel: waves per workgroup selector (0 = 1, 1 = 2, 2 = 4)")
    ap.add_argument("--no-fuse", action="store_true", help="one descriptor per task (no fused substitutions)")
    ap.add_argument("--cut-align", type=int, default=0)
    ap.add_argument("--chunk-tasks", type=int, default=0)
    ap.add_argument("--chunk-bytes", type=int, default=0)
    ap.add_argument("--dbg", type=int, default=0, help="timing-only kernel ablation (results are wrong; implies --verify none)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: exercise launch, sharding and the size all-gather over gloo (CPU test of the N-rank path)")
    ap.add_argument("--xcd-order", type=int, default=-1, help="0: launch chunks in result order instead of dealing them to XCDs by proteome slice")
    a = ap.parse_args()
    if a.no_verify or a.dbg or a.dry_run:
        a.verify = "none"
    return a


def self_launch(n: int) -> int:
    """Parent of an N-rank run: nothing here touches torch.cuda or HIP."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def oracle_digests(workload, n_samples, haps, workers):
    """Digest of the oracle's result tape for every haplotype index in `haps` (thread pool; the C oracle
    and the generator release the GIL)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from sir_oracle import COracle
    from vcf2prot_amd.cohort import Cohort
    orc = COracle()
    workers = max(1, min(workers, len(haps)))

    def work(w):
        cc = Cohort.preset(workload, n_samples=n_samples)       # own generator state per thread
        out = {}
        for h in haps[w::workers]:
            hap = cc.haplotype(h)
            t = orc.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
            want = orc.gir_execute_u8(t, cc.ref_tape_u32(h).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8))
            out[h] = orc.digest_u8(want)
        return out
    res = {}
    with ThreadPoolExecutor(workers) as pool:
        for part in pool.map(work, range(workers)):
            res.update(part)
    return res


def cpu_baseline(cohort, h0, n_haps, n_threads, budget_s=12.0):
    """Oracle (C restatement of task.rs:38-50 / gir.rs:230-234 / exec.rs:34-40), reference-faithful
    flavour: u32 chars, 32-byte AoS tasks, '.' fill, haplotypes over a persistent thread pool."""
    import numpy as np
    import psutil
    from sir_oracle import COracle
    orc = COracle()
    first = cohort.haplotype(h0)
    per_job = 8 * max(first.n_res, 1) + 32 * max(first.n_tasks, 1)          # u32 ref + u32 result + AoS tasks
    fit = int(0.25 * psutil.virtual_memory().available // per_job)
    n_h = max(2, min(n_haps, 2 * n_threads, fit))
    jobs, jobs8, aa, host_bytes = [], [], 0, 0
    for h in range(h0, h0 + n_h):
        hap = cohort.haplotype(h)
        t = orc.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
        ref = cohort.ref_tape_u32(h)
        jobs.append((t, ref, hap.alt.astype(np.uint32), np.empty(hap.n_res, dtype=np.uint32)))
        aa += int(hap.length.sum())
        host_bytes += 8 * int(hap.length.sum()) + 4 * hap.n_res + 32 * hap.n_tasks     # read + write 4 B per residue, '.' fill, tasks
    t1 = orc.mt_execute(jobs, n_threads, wide=True, reps=1)
    reps = max(1, min(200, int(budget_s / max(t1, 1e-3))))
    secs = orc.mt_execute(jobs, n_threads, wide=True, reps=reps)
    for (t, ref, alt, res) in jobs[:max(2, n_h // 4)]:                                   # the 1-byte flavour on a quarter of the jobs
        jobs8.append((t, ref.astype(np.uint8), alt.astype(np.uint8), np.empty(res.size, dtype=np.uint8)))
    aa8 = sum(int(j[3].size) for j in jobs8)
    t8 = orc.mt_execute(jobs8, n_threads, wide=False, reps=1)
    reps8 = max(1, min(200, int(0.3 * budget_s / max(t8, 1e-3))))
    secs8 = orc.mt_execute(jobs8, n_threads, wide=False, reps=reps8)
    return {"value": aa * reps / secs, "unit": "aa/s", "cores": n_threads, "kind": "port",
            "sample": f"first {n_h} haplotypes of the workload ({aa:.3e} aa) x {reps} passes, u32 chars + 32-B tasks + '.' fill, "
                      f"persistent thread pool over haplotypes (Rayon-MT equivalent), {n_h / n_threads:.1f} jobs per thread",
            "host_GBps": host_bytes * reps / secs / 1e9,
            "cpu_best_u8_memcpy": aa8 * reps8 / secs8}


def pcie_inclusive(cohort, h0, h1, n_threads, slots=3, target_image_bytes=2 << 30):
    """Transfers-inclusive rate through the C ABI's streamed pipeline (v2p_pipeline_*): H2D of descriptors + alt bytes,
    stitch kernel, D2H of the result into pinned host memory, `slots` images in flight.  Not `value`."""
    from vcf2prot_amd.engine import Context, Pipeline
    sizes = cohort.result_sizes(h0, h1)
    total = int(sizes.sum())
    n_img = max(2, min(16, (total + target_image_bytes - 1) // target_image_bytes))
    per = (h1 - h0 + n_img - 1) // n_img
    imgs = [cohort.pack(h, min(h1, h + per), n_threads=n_threads) for h in range(h0, h1, per)]
    aa = sum(i.n_copy_bytes for i in imgs)
    out_total = sum(i.out_bytes for i in imgs)
    h2d = sum(i.desc.nbytes + i.chunks.nbytes + i.payload.nbytes for i in imgs)
    best = None
    with Context(0) as ctx:
        ctx.upload_proteome(cohort.proteome())
        pipe = Pipeline(ctx, slots)
        for rep in range(3):                           # the first pass allocates and pins
            t0 = time.perf_counter()
            inflight = []
            for img in imgs:
                if len(inflight) == slots:
                    t = inflight.pop(0)
                    pipe.wait(t)
                    pipe.release(t)
                inflight.append(pipe.submit(img.desc, img.chunks, img.payload, img.out_bytes))
            for t in inflight:
                pipe.wait(t)
                pipe.release(t)
            secs = time.perf_counter() - t0
            if rep and (best is None or secs < best):
                best = secs
        pipe.close()
    return {"aa_per_s": aa / best, "seconds": best, "images": len(imgs), "slots": slots, "h2d_bytes": h2d, "d2h_bytes": out_total,
            "d2h_GBps": out_total / best / 1e9, "what": "packed images -> pinned H2D -> stitch kernel -> D2H into pinned host memory"}


def device_image_build(cohort, h0, h1, n_threads, long_run, want_digests, dense=False, wave=False, result_bytes=0):
    """the same shard's image built ON the device from the per-transcript GIRs of step 4b (v2p_batch_build_on_device:
    step 5's running sums as prefix scans, descriptors, chunk table, XCD order).  Returns the build kernels' time and whether the
    image executes to the same per-haplotype digests."""
    import numpy as np
    from vcf2prot_amd.engine import Context
    t0 = time.perf_counter()
    stream = cohort.txstream(h0, h1, n_threads=n_threads)
    t_stream = time.perf_counter() - t0
    from vcf2prot_amd._native import V2PError
    from vcf2prot_amd.txstream import build_plan
    # the routing of vcf2prot_amd/txstream.py::build_plan (what pipeline.vcf_to_fasta and v2p_harness vcf use): wave windows that may
    # split once, sized by the image's result bytes per task; dense: one window = the kernel's 12 KiB LDS image when its descriptors fit
    if dense:
        plan = [(3, 12288), (3, 8192), (3, 4096)]
    elif wave:
        plan = build_plan(result_bytes / max(stream.n_tasks, 1) if result_bytes else 134.0)
    else:
        plan = [(1, 28672)] if long_run else [(2, 32768)]
    with Context(0) as ctx:
        ctx.upload_proteome(cohort.proteome())
        for kernel, window in plan:
            b = ctx.batch()
            try:
                t0 = time.perf_counter()
                ms = b.build_on_device(stream, window, kernel)
                t_call = time.perf_counter() - t0
                break
            except V2PError:
                b.close()
                if (kernel, window) == plan[-1]:
                    raise
        cn = b.counts()
        same = None
        exec_ms = None
        if want_digests is not None:
            import torch
            b.execute()
            b.sync()
            same = bool(np.array_equal(b.digests(), want_digests))
            ts = torch.cuda.Stream()                             # (a stream of its own: the null stream would mean "the ctx's own" to the ABI)
            ctx.set_stream(ts.cuda_stream)                       # HIP events on the stream the kernel is launched on
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            for e0, e1 in evs:
                e0.record(ts)
                b.execute()
                e1.record(ts)
            b.sync()
            exec_ms = sum(e0.elapsed_time(e1) for e0, e1 in evs) / len(evs)
            ctx.set_stream(0)
        b.close()
    res = {"build_kernels_ms": ms, "call_s_incl_h2d_of_the_stream": t_call, "stream_bytes": stream.nbytes, "stream_generation_s": t_stream,
           "window_bytes": window, "kernel_choice": kernel, "descriptors": cn["n_desc"], "chunks": cn["n_chunks"], "digests_equal_host_built_image": same,
           "execute_ms_device_built_image": exec_ms,
           "what": "per-transcript GIRs (un-rebased Task SoA, transcript offsets, alt bytes) -> descriptors + chunk table + hap_out_begin in HBM"}
    stream.close()
    return res


def whole_cohort_leg(workload, samples, steps, n_threads, verify_every=True, temporal=False, device_build=False):
    """One GPU, one launch per step over a WHOLE cohort (no sharding, no collective): pack, upload, verify every haplotype's digest
    against the oracle, then time `steps` launches with HIP events on the launch stream.  Used for the north star's 10 000-sample
    cohort next to the N = 1 line and for `speedup_vs_1` of a strong-scaling run."""
    import numpy as np
    import torch
    from vcf2prot_amd import _native as N
    from vcf2prot_amd.cohort import Cohort
    lib = N.hip_lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    cohort = Cohort.preset(workload, n_samples=samples)
    n_haps = cohort.n_haplotypes
    t0 = time.perf_counter()
    img = cohort.pack(0, n_haps, n_threads=min(n_threads, 64))
    t_pack = time.perf_counter() - t0
    proteome = cohort.proteome()
    PAD = 64

    def padded(arr):
        t = torch.zeros(arr.size + 2 * PAD, dtype=torch.uint8, device=dev)
        if arr.size:
            t[PAD:PAD + arr.size] = torch.from_numpy(arr).to(dev)
        return t
    d_prot, d_payload = padded(proteome), padded(img.payload)
    chunks = np.ascontiguousarray(img.chunks)
    assert lib.v2p_order_chunks_for_xcds(chunks.ctypes.data, chunks.shape[0], img.desc.ctypes.data, img.desc.size, proteome.size) == 0
    d_desc = padded(img.desc.view(np.uint8))                   # (64 readable bytes either side: the ABI asks for 16 before and 32 behind)
    d_chunks = torch.from_numpy(chunks.view(np.int64)).to(dev)
    d_hap = torch.from_numpy(img.hap_out_begin.view(np.int64)).to(dev)
    out_bytes = img.out_bytes
    d_out = torch.empty(out_bytes + 32, dtype=torch.uint8, device=dev)
    d_status = torch.full((1,), -1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()
    flags = (0 if temporal else 1) | img.launch_bits

    def launch():
        rc = lib.v2p_stitch_launch(ctypes.c_void_p(stream.cuda_stream), d_desc.data_ptr() + PAD, img.desc.size, d_chunks.data_ptr(), chunks.shape[0],
                                   d_prot.data_ptr() + PAD, proteome.size, d_payload.data_ptr() + PAD, img.payload.size,
                                   d_out.data_ptr(), out_bytes, d_status.data_ptr(), flags, 0)
        if rc != 0:
            raise RuntimeError(f"v2p_stitch_launch failed: {rc}")
    launch()
    launch()
    torch.cuda.synchronize()
    if int(d_status.item()) != -1:
        raise RuntimeError(f"device reported a task error: status={int(d_status.item()):#x}")
    d_dig = torch.zeros(n_haps, dtype=torch.int64, device=dev)
    lib.v2p_digest_launch(ctypes.c_void_p(stream.cuda_stream), d_out.data_ptr(), d_hap.data_ptr(), n_haps, out_bytes, d_dig.data_ptr())
    torch.cuda.synchronize()
    dig = d_dig.cpu().numpy().view(np.uint64)
    check = list(range(n_haps)) if verify_every else sorted(set(np.linspace(0, n_haps - 1, min(n_haps, 512)).astype(int).tolist()))
    t_v = time.perf_counter()
    want = oracle_digests(workload, samples, check, min(n_threads, 64))
    bad = [h for h in check if int(dig[h]) != want[h]]
    if bad:
        raise RuntimeError(f"PARITY FAILURE: haplotypes {bad[:8]} of {workload} differ from the oracle ({len(bad)} of {len(check)})")
    t_v = time.perf_counter() - t_v
    for _ in range(3):                                        # (the GPU sat idle while the host checked the digests: untimed warm-up, as the main leg has)
        launch()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for e0, e1 in ev:
        e0.record(stream)
        launch()
        e1.record(stream)
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    spaces = (img.desc >> np.uint64(62)).astype(np.uint8)
    lens = ((img.desc >> np.uint64(40)) & np.uint64((1 << 22) - 1)).astype(np.int64)
    n_fused = int(((img.desc >> np.uint64(61)) == 7).sum())
    hbm_min = out_bytes + 8 * int(img.desc.size) + 16 * int(chunks.shape[0]) + int(lens[spaces == 1].sum()) + int(proteome.size)
    avg = sum(ms) / len(ms)
    bits = img.launch_bits
    res = {"workload": f"{workload}: {samples} samples = {n_haps} haplotypes x {cohort.n_transcripts} transcripts, whole cohort in ONE launch on one GPU",
           "aa": int(img.n_copy_bytes), "tasks": int(img.n_tasks), "result_bytes": int(out_bytes), "descriptors": int(img.desc.size),
           "fused_substitution_descriptors": n_fused, "chunks": int(chunks.shape[0]),
           "kernel": "stitchw_kernel" if bits & 4 else ("stitch4_kernel" if not (bits & 16) else ("stitch_dense_kernel" if bits & 2 else "stitch_kernel (per block)")),
           "steps": steps, "warmup": 3, "ms": avg, "ms_min": min(ms), "aa_per_s": img.n_copy_bytes / (avg * 1e-3),
           "hbm_bytes_min_per_launch": hbm_min, "achieved_GBps": hbm_min / (avg * 1e-3) / 1e9, "frac": hbm_min / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "every_haplotype": bool(len(check) == n_haps), "haplotypes_checked": len(check), "oracle_seconds": t_v, "image_build_s": t_pack}
    del d_out, d_desc, d_chunks, d_payload, d_prot
    torch.cuda.empty_cache()
    if device_build:                                          # the same cohort's image built on the device from the transcript stream
        try:
            res["device_image_build"] = device_image_build(cohort, 0, n_haps, min(n_threads, 64), not (bits & 16), dig, dense=bool(bits & 2), wave=bool(bits & 4),
                                                           result_bytes=int(out_bytes))
        except Exception as e:                                # noqa: BLE001  (the leg's own numbers stand)
            res["device_image_build"] = {"error": repr(e)}
    return res


def main():
    args = parse_args()
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    if args.workload is None:
        args.workload = "C2" if world == 1 else "C3"
    if args.scaling is None:
        args.scaling = "weak" if world == 1 else "strong"

    import numpy as np
    import torch
    import torch.distributed as dist

    if args.dry_run:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            sys.exit("bench.py needs an MI355X: the gpu engine has no CPU fallback")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    dist_on = "RANK" in os.environ and "MASTER_PORT" in os.environ      # launched by torch.distributed.run
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_run:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    # native pieces: built once per node (local rank 0), everybody else waits
    from vcf2prot_amd import build
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    if local_rank == 0:
        if args.dry_run:
            build.build_cohort()                               # host logic only: no hipcc, no HIP runtime
        else:
            build.build_all()
        if args.verify != "none" or not args.no_cpu_baseline:
            import sir_oracle
            sir_oracle.build_c_oracle()
    if dist_on:
        dist.barrier()
    from vcf2prot_amd import _native as N
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.shard import shard_by_bytes
    # (--dbg: timing-only ablations live in libv2p_bench.so, the V2P_BENCH_VARIANTS build of the engine)
    lib = None if args.dry_run else (N.bench_lib() if args.dbg else N.hip_lib())

    # ---- synthetic cohort at the Task boundary; this rank's shard --------------------
    samples = args.samples or DEFAULT_SAMPLES[args.scaling][args.workload]
    n_threads = max(1, (os.cpu_count() or 1) // world)
    if args.scaling == "weak":
        cohort_samples = samples * world
        cohort = Cohort.preset(args.workload, n_samples=cohort_samples)
        h0, h1 = 2 * samples * rank, 2 * samples * (rank + 1)
    else:
        cohort_samples = samples
        cohort = Cohort.preset(args.workload, n_samples=cohort_samples)
        sizes = cohort.result_sizes(0, cohort.n_haplotypes, n_threads=min(n_threads, 64))
        h0, h1 = shard_by_bytes(sizes.tolist(), world)[rank]              # equal result bytes per rank
    speedup_ref = None
    if world > 1 and args.scaling == "strong" and rank == 0 and not args.no_speedup_ref and not args.dry_run:
        try:                                                   # the same cohort alone on this GPU, one launch (the 1-GPU point of the curve)
            speedup_ref = whole_cohort_leg(args.workload, cohort_samples, max(3, min(args.steps, 10)), n_threads, verify_every=args.verify == "all")
        except Exception as e:
            speedup_ref = {"error": repr(e)}
    if dist_on:
        dist.barrier()
    t_gen = time.perf_counter()
    img = cohort.pack(h0, h1, n_threads=min(n_threads, 64), chunk_tasks=args.chunk_tasks, chunk_bytes=args.chunk_bytes, fasta=args.fasta, cut_align=args.cut_align,
                      fuse=not args.no_fuse and args.var not in (1, 2, 3) and not args.max_blocks, kernel=2 if (args.var in (1, 2, 3) or args.max_blocks) else args.kernel)
    t_gen = time.perf_counter() - t_gen
    A, NT = img.n_copy_bytes, img.n_tasks
    b_alg = 2 * A + 16 * NT
    proteome = cohort.proteome()
    n_proteome = proteome.size
    if args.fasta:                                             # resident reference = proteome + record headers
        proteome = np.concatenate([proteome, cohort.fasta_headers()])
        args.verify = "none"                                   # digests are defined on the plain result tape

    PAD = 64                                                   # the ABI asks for 32 readable bytes either side

    def padded(arr):
        t = torch.zeros(arr.size + 2 * PAD, dtype=torch.uint8, device=dev)
        if arr.size:
            t[PAD:PAD + arr.size] = torch.from_numpy(arr).to(dev)
        return t

    d_prot, d_payload = padded(proteome), padded(img.payload)
    d_desc = padded(img.desc.view(np.uint8))                   # (readable slack either side, like the source tapes: include/vcf2prot_hip.h)
    n_desc = int(img.desc.size)
    img.chunks = np.ascontiguousarray(img.chunks)
    if args.xcd_order != 0 and not args.dry_run:           # XCD-aware launch order (speed only; chunks are independent)
        rc = lib.v2p_order_chunks_for_xcds(img.chunks.ctypes.data, img.chunks.shape[0], img.desc.ctypes.data, img.desc.size, n_proteome)
        assert rc == 0
    d_chunks = torch.from_numpy(np.ascontiguousarray(img.chunks).view(np.int64)).to(dev)
    d_hap = torch.from_numpy(img.hap_out_begin.view(np.int64)).to(dev)
    out_bytes = img.out_bytes
    d_out = torch.empty(out_bytes + 32, dtype=torch.uint8, device=dev)
    d_status = torch.full((1,), -1, dtype=torch.int64, device=dev)
    n_chunks = int(img.chunks.shape[0])
    n_haps = int(img.hap_out_begin.size - 1)
    # immediate descriptors carry their bytes; what the kernel can touch of the payload arena is the rest
    spaces = (img.desc >> np.uint64(62)).astype(np.uint8)
    lens = ((img.desc >> np.uint64(40)) & np.uint64((1 << 22) - 1)).astype(np.int64)
    payload_touched = int(lens[spaces == 1].sum())
    n_fused = int(((img.desc >> np.uint64(61)) == 7).sum())
    n_imm = int((spaces == 3).sum()) - n_fused
    # bytes that must cross the HBM interface per launch: every result byte written once, every descriptor and chunk
    # header read once, the alt bytes that are not inside a descriptor, the proteome once
    hbm_min = out_bytes + 8 * n_desc + 16 * n_chunks + payload_touched + int(proteome.size)
    del spaces, lens
    assert d_out.data_ptr() % 16 == 0
    stream = None if args.dry_run else torch.cuda.current_stream()
    sizes_t = torch.tensor([n_haps, out_bytes], dtype=torch.int64, device=dev)
    all_sizes = torch.zeros(2 * world, dtype=torch.int64, device=dev)
    flags = (0 if args.temporal else 1) | img.launch_bits | (args.var << 12) | (args.dbg << 16) | (args.wpg << 28)

    def launch():
        if args.dry_run:
            if dist_on:
                dist.all_gather_into_tensor(all_sizes, sizes_t)
            return
        rc = lib.v2p_stitch_launch(ctypes.c_void_p(stream.cuda_stream), d_desc.data_ptr() + PAD, n_desc, d_chunks.data_ptr(), n_chunks,
                                   d_prot.data_ptr() + PAD, proteome.size, d_payload.data_ptr() + PAD, img.payload.size,
                                   d_out.data_ptr(), out_bytes, d_status.data_ptr(), flags, args.max_blocks)
        if rc != 0:
            raise RuntimeError(f"v2p_stitch_launch failed: {rc}")
        if dist_on:                                            # the path's only exchange: result sizes for the global offsets
            dist.all_gather_into_tensor(all_sizes, sizes_t)

    sync = (lambda: None) if args.dry_run else torch.cuda.synchronize
    for _ in range(max(args.warmup, 1) if args.verify != "none" else args.warmup):
        launch()
    sync()
    if int(d_status.item()) != -1 and not args.dbg:
        sys.exit(f"device reported a task error: status={int(d_status.item()):#x}")

    # ---- parity before timing: per-haplotype digests vs the oracle --------------------
    verified = None
    dig_all = None
    if args.verify != "none":
        d_dig = torch.zeros(n_haps, dtype=torch.int64, device=dev)
        lib.v2p_digest_launch(ctypes.c_void_p(stream.cuda_stream), d_out.data_ptr(), d_hap.data_ptr(), n_haps, out_bytes, d_dig.data_ptr())
        torch.cuda.synchronize()
        dig = d_dig.cpu().numpy().view(np.uint64)
        dig_all = dig.copy()
        every = args.verify == "all" and out_bytes <= 24 * 10 ** 9
        check = list(range(n_haps)) if every else sorted(set(np.linspace(0, n_haps - 1, min(n_haps, 512)).astype(int).tolist()))
        t_v = time.perf_counter()
        want = oracle_digests(args.workload, cohort_samples, [h0 + i for i in check], min(n_threads, 64))
        bad = [h0 + i for i in check if int(dig[i]) != want[h0 + i]]
        if bad:
            sys.exit(f"PARITY FAILURE: haplotypes {bad[:8]} differ from the oracle ({len(bad)} of {len(check)})")
        verified = {"haplotypes_checked": len(check), "of": n_haps, "every_haplotype": bool(len(check) == n_haps),
                    "digest_of_digests": f"{int(np.bitwise_xor.reduce(dig)):016x}", "oracle_seconds": time.perf_counter() - t_v}

    # ---- timed region -------------------------------------------------------------
    ev = [] if args.dry_run else [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if dist_on:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        if ev:
            ev[k][0].record(stream)
        launch()
        if ev:
            ev[k][1].record(stream)
    sync()
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = [a.elapsed_time(b) for a, b in ev] or [1e3 * elapsed / max(args.steps, 1)]
    if int(d_status.item()) != -1 and not args.dbg:
        sys.exit(f"device reported a task error: status={int(d_status.item()):#x}")

    tot = torch.tensor([elapsed, float(A), float(NT), float(n_haps), float(out_bytes)], dtype=torch.float64, device=dev)
    per_rank = None
    if dist_on:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        gathered = torch.zeros(5 * world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(gathered, tot)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0].item())
        A_all, NT_all = float(tot[1].item()), float(tot[2].item())
        g = gathered.cpu().view(world, 5).tolist()
        per_rank = [{"rank": r, "seconds": g[r][0], "aa": int(g[r][1]), "haplotypes": int(g[r][3]), "result_bytes": int(g[r][4])} for r in range(world)]
        seen = all_sizes.cpu().view(world, 2).tolist()          # what the in-step all-gather delivered: every rank's sizes
        assert [int(x[0]) for x in seen] == [p["haplotypes"] for p in per_rank], "all-gather of result sizes disagrees"
    else:
        A_all, NT_all = float(A), float(NT)

    if rank == 0:
        avg_ms = sum(kern_ms) / len(kern_ms)
        achieved = hbm_min / (avg_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                tj = tj.get(args.workload, tj)
                if tj.get("workload") == args.workload and tj.get("haplotypes") == n_haps:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = ("replayed, not measured in this run: " + str(tj.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command")) +
                                      " (profiles/traffic_latest.json)")
            except Exception:
                traffic = None
        line = {
            "metric": "amino-acids written/sec", "value": A_all * args.steps / elapsed, "unit": "aa/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.workload}: " + (f"{samples} samples/GPU x {world} GPU(s)" if args.scaling == "weak" else f"one {samples}-sample cohort over {world} GPU(s), equal result bytes per rank")
                                   + f" ({int(A_all):.3e} aa) x {cohort.n_transcripts} transcripts, SIR Task vectors at the step-6 boundary",
                       "haplotypes_rank0": n_haps, "tasks_rank0": NT, "aa_rank0": A, "chunks_rank0": n_chunks,
                       "descriptor_bytes": 8, "long_run_chunks_rank0": int((img.chunks[:, 1] >> np.uint64(63)).sum()), "immediate_descriptors_rank0": n_imm, "fused_substitution_descriptors_rank0": n_fused, "descriptors_rank0": n_desc, 
                       "parallelism": f"haplotype-sharded x{world}, no data-path collective; per step one all-gather of 16 B per rank (RCCL)" if world > 1 else "1 GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "bytes": "achieved = hbm_bytes_min / kernel time: result bytes written once + 8 B per descriptor + 16 B per chunk + alt bytes "
                                  "outside descriptors + the proteome once (reference reads are served by L2 and not counted)",
                         "hbm_bytes_min_per_launch": hbm_min,
                         "algorithmic_bytes_per_launch": b_alg, "algorithmic_GBps": b_alg / (avg_ms * 1e-3) / 1e9,
                         "kernel": "stitchw_kernel (wave image: one wave per chunk; a step = its phases of 64 MB of image, each read ahead into the memory-side cache -- kernel_ms is the whole step)" if (img.launch_bits & 4) else "stitch4_kernel (long-run image)" if not (img.launch_bits & 16) else ("stitch_dense_kernel (short tasks)" if ((img.launch_bits & 2) or ((img.launch_bits >> 8) & 15) > 2) and args.var in (0, 8, 9) else "stitch_kernel (per-block)"), "kernel_ms_avg": avg_ms, "kernel_ms_min": min(kern_ms)},
            "kernel_only_aa_per_s_rank0": A / (avg_ms * 1e-3),
            "verified": verified, "image_build_s": t_gen,
        }
        if per_rank:
            line["per_rank"] = per_rank
            line["world_size_seen_by_rccl"] = int(dist.get_world_size()) if dist_on else 1
        if speedup_ref is not None:
            line["one_gpu_reference"] = speedup_ref
            if "ms" in speedup_ref:                              # same cohort, same step: time on one GPU / time on `world` GPUs
                line["speedup_vs_1"] = speedup_ref["ms"] / (1e3 * elapsed / args.steps)
        if args.dry_run:
            line["data"] = "synthetic (dry run: no kernel was launched, value is meaningless)"
        # free the big device buffers before the host-side legs
        del d_out
        if not args.dry_run:
            torch.cuda.empty_cache()
        if world == 1 and not args.no_pcie and not args.fasta and not args.dbg and not args.dry_run:
            try:
                pc = pcie_inclusive(cohort, h0, h1, min(n_threads, 64))
                line["incl_transfers_aa_per_s"] = pc.pop("aa_per_s")
                line["incl_transfers"] = pc
            except Exception as e:           # never lose the bench line to the secondary leg
                line["incl_transfers"] = {"error": repr(e)}
        if world == 1 and not args.no_device_build and not args.fasta and not args.dbg and not args.dry_run:
            try:
                line["device_image_build"] = device_image_build(cohort, h0, h1, min(n_threads, 64), not (img.launch_bits & 16), dig_all, dense=bool(img.launch_bits & 2), wave=bool(img.launch_bits & 4), result_bytes=int(out_bytes))
            except Exception as e:
                line["device_image_build"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline and not args.dry_run:
            line["cpu_baseline"] = cpu_baseline(cohort, h0, n_haps, os.cpu_count() or 1)
        if world == 1 and not args.no_north_star and not args.dry_run and not args.dbg and not args.fasta and args.workload == "C2" and args.kernel == 0 and not args.var:
            # the north star's own cohort (BASELINE.json configs[2]): 10 000 samples, whole, one launch, every haplotype verified
            try:
                del d_desc, d_chunks, d_payload
                torch.cuda.empty_cache()
                line["north_star_cohort"] = whole_cohort_leg("C3", DEFAULT_SAMPLES["strong"]["C3"], 10, n_threads, verify_every=True, device_build=not args.no_device_build)
            except Exception as e:
                line["north_star_cohort"] = {"error": repr(e)}
        print(json.dumps(line))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
