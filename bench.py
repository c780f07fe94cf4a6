#!/usr/bin/env python3
"""Benchmark of the MI355X SIR executor (vcf2prot step 6) -- driver contract in the task brief.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C2|C3|C4|C5] [--samples S] [--scaling weak|strong]

One "step" = one pass of the hot path -- v2p_batch_execute through the C ABI: the stitch kernel's phases (K0 fill + K1 in-chunk
scan + K2 gather / scatter) over this rank's whole shard, inputs resident in HBM.  The workload at every N is the north star's
cohort, BASELINE.json configs[2] ("C3": 10 000 samples = 20 000 haplotypes, full alteration mix, 3.6e10 residues), strong scaling:
at N = 1 the whole cohort is ONE image and one execute per step; at N > 1 it is cut into contiguous haplotype ranges of equal
result bytes (shard.shard_by_bytes), one rank per GPU, no data-path collective.  The image every step executes is the one the
product builds: ON the device, from the per-transcript Task vectors of step 4b, by the one call that also executes it the first time
(v2p_stream_upload + v2p_batch_build_and_execute, rows images).

The N = 1 line also carries
  one_shot        Task vectors -> result bytes ONCE, as the reference does it: ONE C-ABI call (v2p_batch_build_and_execute) on the resident stream
  host_packed     the same cohort's host-packed image executed alternately with the device-built one
  c2_cohort       BASELINE.json configs[1] ("C2": 1 000 samples x 20 k transcripts, one missense each) the same way
  cpu_baseline    the oracle's reference-faithful flavour on this box's host cores (bounded sample)
  incl_transfers  PCIe-inclusive rate through v2p_pipeline_* (not `value`)
and roofline.traffic measured in the run itself: last of all, two child runs of this file under `rocprofv3 --pmc FETCH_SIZE` / `--pmc
WRITE_SIZE` (separate passes, --kernel-trace only beside them) execute the same image between two marker launches; any failure there
leaves the figure replayed from profiles/traffic_latest.json, labelled so (--no-live-traffic: skip).

--gpus N > 1 launches itself: the parent spawns `python -m torch.distributed.run` with N ranks (one per GPU, RCCL) before it touches
any GPU, and exits with the children's code; under an external launcher (RANK/WORLD_SIZE set) it just runs as a rank.
  --scaling strong one cohort of `--samples` samples cut by result bytes (default)
  --scaling weak   every rank executes its own `--samples`-sample shard of a samples*N cohort
Haplotypes are independent: the path's only exchange is ONE all-gather of {haplotypes, result bytes} per image (the sizes do not
change between steps); it is timed by itself (`allgather_us`).  Every rank verifies its own shard against the oracle; rank 0 prints
ONE JSON line.
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
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C3", choices=["C2", "C3", "C4", "C5"], help="default: C3, the north star's cohort")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"])
    ap.add_argument("--samples", type=int, default=0, help="weak: samples per GPU; strong: samples of the whole cohort (0 = the config's own size)")
    ap.add_argument("--no-c2", action="store_true", help="N = 1: skip the c2_cohort leg (BASELINE configs[1])")
    ap.add_argument("--no-host-packed", action="store_true", help="skip the host-packed image of the same cohort (A/B of the two builders)")
    ap.add_argument("--speedup-ref", action="store_true", help="N > 1, strong scaling: rank 0 first times the whole cohort alone (speedup_vs_1 on one clock); off by default -- "
                                                                "the other ranks would wait in a barrier for it, and the driver computes the scaling curve from its own N = 1 run")
    ap.add_argument("--clock-settle-ms", type=float, default=50.0, help="GPU kept busy with the checker's digest kernel for this long right before the W warm-up steps (0: off): the "
                    "clocks are down after the host's oracle checks and W short steps of a small shard do not bring them back")
    ap.add_argument("--no-live-traffic", action="store_true", help="N = 1: do not measure roofline.traffic in this run (two child runs of this file under rocprofv3 --pmc, "
                    "about a minute); the figure is then replayed from profiles/traffic_latest.json and labelled so")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)      # (the child of live_traffic(): the image, a few executes between two digest launches, nothing else)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the transfers-inclusive leg (v2p_pipeline_*)")
    ap.add_argument("--verify", default="all", choices=["all", "sample", "none"], help="haplotypes whose digest is compared with the oracle before timing")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--temporal", action="store_true", help="plain result stores instead of non-temporal")
    ap.add_argument("--host-image", action="store_true", help="time the host-packed image instead of the device-built one")
    ap.add_argument("--collectives", default="nccl", choices=["nccl", "gloo"], help="nccl (= RCCL; the product) or gloo: a TEST mode in which ranks may share a GPU "
                    "(device = local rank mod devices present) and the size exchange runs on CPU tensors")
    ap.add_argument("--dry-run", action="store_true", help="no GPU work: exercise launch, sharding and the size all-gather over gloo (CPU test of the N-rank path)")
    a = ap.parse_args()
    if a.no_verify or a.dry_run:
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
    """Digest of the oracle's result tape for every haplotype index in `haps` (thread pool; the C oracle and the generator release the GIL)."""
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
    """Oracle (C restatement of task.rs:38-50 / gir.rs:230-234 / exec.rs:34-40), reference-faithful flavour: u32 chars, 32-byte AoS
    tasks, '.' fill, haplotypes over a persistent thread pool."""
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
    """Transfers-inclusive rate through the C ABI's streamed pipeline (v2p_pipeline_*): H2D of descriptors + alt bytes, stitch
    kernel, D2H of the result into pinned host memory, `slots` images in flight.  Not `value`."""
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


def stream_pipeline_leg(workload, cohort_samples, n_threads, ref_digests, target_slice_bytes=1152 << 20, slots=4, reps=2, device=None):
    """Transfers-inclusive rate FROM THE TASK STREAM (round 6): the whole cohort, slice by slice, through v2p_pipeline_submit_stream -- the
    per-transcript Task vectors of step 4b sit in host memory (pageable, as a Rust host's Vecs would), each slice is checked and copied into
    pinned staging by the submitting thread's copy team, uploaded, built + executed by the one call, and its arena comes back into pinned
    host memory; `slots` slices in flight.  Nothing is packed on the host; the upload is inside the timed region.  Every haplotype's device
    digest is compared with `ref_digests` (each of which was compared with the oracle's) in EVERY pass; the first, untimed pass (it pins the
    buffers) also digests a sample of the HOST bytes with the oracle's digest function.  Not `value`."""
    import numpy as np
    from sir_oracle import COracle
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.engine import Context, Pipeline
    from vcf2prot_amd.shard import shard_by_bytes
    if device is None:
        import torch
        device = torch.cuda.current_device()
    cohort = Cohort.preset(workload, n_samples=cohort_samples)
    n = cohort.n_haplotypes
    sizes = cohort.result_sizes(0, n, n_threads=n_threads)
    total = int(sizes.sum())
    n_sl = int(max(slots, min(48, (total + target_slice_bytes - 1) // target_slice_bytes)))
    ranges = [(a, b) for a, b in shard_by_bytes(sizes.tolist(), n_sl) if b > a]
    t0 = time.perf_counter()
    streams = [cohort.txstream(a, b, n_threads=n_threads) for a, b in ranges]
    t_gen = time.perf_counter() - t0
    A = 0
    for st in streams:
        if st.n_tasks:
            A += int(np.ctypeslib.as_array(st.struct.length, shape=(st.n_tasks,)).sum(dtype=np.int64))
    h2d = sum(st.nbytes for st in streams)
    out_max = max(int(sizes[a:b].sum()) for a, b in ranges)
    orc = COracle()
    copy_threads = max(1, min(16, n_threads))
    res = {"input": "task stream", "host_packing_s": 0.0, "workload": f"{workload}: the whole {cohort_samples}-sample cohort ({n} haplotypes)", "slices": len(ranges), "slots": slots,
           "copy_threads": copy_threads, "h2d_bytes": h2d, "d2h_bytes": total, "stream_generation_s_outside": t_gen}
    with Context(device) as ctx:
        ctx.upload_proteome(cohort.proteome())
        pipe = Pipeline(ctx, slots)
        t0 = time.perf_counter()
        pipe.reserve(int(1.05 * max(st.nbytes for st in streams)) + (1 << 20), out_max + 8 * n + (1 << 20), copy_threads)
        res["pinning_s_outside"] = time.perf_counter() - t0
        passes, stage_ms, runner_ms, host_checked = [], [], [], 0

        def consume(job, first):
            nonlocal host_checked
            t, i = job
            a, b = ranges[i]
            out = pipe.wait(t)
            info = pipe.result_info(t)
            if not np.array_equal(info["digests"], ref_digests[a:b]):
                bad = int((info["digests"] != ref_digests[a:b]).sum())
                raise RuntimeError(f"PARITY FAILURE: {bad} haplotypes of slice {i} of {workload} came back from the stream pipeline with another digest")
            if out.size != int(sizes[a:b].sum()):
                raise RuntimeError(f"PARITY FAILURE: slice {i} returned {out.size} bytes")
            if first:                                       # what crossed the link: host bytes digested by the oracle's function
                hob = info["hap_out_begin"]
                for j in sorted(set(np.linspace(0, b - a - 1, 3).astype(int).tolist())):
                    if orc.digest_u8(np.ascontiguousarray(out[int(hob[j]):int(hob[j + 1])])) != int(ref_digests[a + j]):
                        raise RuntimeError(f"PARITY FAILURE: the host bytes of haplotype {a + j} differ from the oracle")
                    host_checked += 1
            stage_ms.append(info["stage_ms"]); runner_ms.append(info["runner_ms"])
            pipe.release(t)

        for rep in range(1 + reps):
            stage_ms.clear(); runner_ms.clear()
            t0 = time.perf_counter()
            inflight = []
            for i, st in enumerate(streams):
                if len(inflight) == slots:
                    consume(inflight.pop(0), rep == 0)
                inflight.append((pipe.submit_stream(st, 0, True), i))
            while inflight:
                consume(inflight.pop(0), rep == 0)
            secs = time.perf_counter() - t0
            if rep:
                passes.append(secs)
        pipe.close()
    for st in streams:
        st.close()
    best = min(passes)
    res.update({"seconds": best, "seconds_all_passes": passes, "aa_per_s": A / best, "d2h_GBps": total / best / 1e9, "h2d_GBps": h2d / best / 1e9,
                "stage_ms_per_slice": sum(stage_ms) / len(stage_ms), "runner_ms_per_slice": sum(runner_ms) / len(runner_ms),
                "verified": {"every_haplotype_digest_every_pass": True, "host_bytes_digested": host_checked},
                "what": "Task vectors (pageable host memory) -> table checks + copy into pinned staging on the submitting thread's copy team -> H2D -> "
                        "v2p_batch_build_and_execute's one call (runner thread) -> D2H of the arena into pinned host memory; upload inside the timed region, nothing packed on the host"})
    return res


def image_stats(desc, chunks, proteome_bytes, out_bytes):
    """Bytes that must cross the HBM interface per execute: every result byte written once, every descriptor and chunk record read
    once, the alt bytes that are not inside a descriptor, the proteome once."""
    import numpy as np
    spaces = (desc >> np.uint64(62)).astype(np.uint8)
    fused = ((desc >> np.uint64(61)) == 7) | ((desc >> np.uint64(60)) == 0xD)
    lens = ((desc >> np.uint64(40)) & np.uint64((1 << 22) - 1)).astype(np.int64)
    payload_touched = int(lens[(spaces == 1)].sum())
    n_fused = int(fused.sum())
    n_imm = int(((spaces == 3) & ~fused).sum())
    hbm_min = int(out_bytes) + 8 * int(desc.size) + 16 * int(chunks.shape[0]) + payload_touched + int(proteome_bytes)
    return hbm_min, n_fused, n_imm


class Timed:
    """A finalized batch on a context whose launches go to a torch stream: execute() timed with HIP events on that stream."""

    def __init__(self, ctx, batch):
        import torch
        self.torch, self.ctx, self.b = torch, ctx, batch
        self.ts = torch.cuda.Stream()

    def settle_clocks(self, ms):
        """Keeps the GPU busy for `ms` with the checker's own kernel -- v2p_batch_digests over the batch's arena, not a step of the path -- right
        before the warm-up steps.  After the host's seconds of oracle checking the GPU's clocks are down and come back over ~40 ms of work
        of this kind (profiles/r05_first_execute.txt; a bare fill or copy brings them back only part of the way: profiles/
        r05_clock_settle.txt).  A whole-cohort step's W = 5 warm-up steps are 36 ms and cover most of that; an eighth of the cohort's are
        4.6 ms and do not, so without this the N-rank points of the scaling curve would be timed on a lower clock than the 1-rank point."""
        if ms <= 0:
            return 0.0
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < ms:
            self.b.digests()
        return (time.perf_counter() - t0) * 1e3

    def run(self, steps, warmup, barrier=lambda: None, settle_ms=0.0):
        """[settle_ms of the digest kernel: clocks up] W untimed steps, then EXACTLY K steps bracketed by a barrier + torch.cuda.synchronize() on both
        sides (the driver's contract)."""
        torch, ts = self.torch, self.ts
        self.ctx.set_stream(ts.cuda_stream)
        barrier()                                               # (ranks meet BEFORE warming up: a rank that finished its checks early must not idle -- clocks down again -- between its warm-up and the timed region)
        self.settled_ms = self.settle_clocks(settle_ms)
        self.b.scribble()                                       # (the arena verified before is overwritten: what is digested after the timed steps is what THEY wrote, in the image's re-execution form)
        for _ in range(warmup):
            self.b.execute()
        self.b.sync()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        import gc
        gc_was = gc.isenabled()
        gc.disable()                                            # (a collection inside a 20 ms timed region of 1 ms steps is 5 % of it)
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for e0, e1 in ev:
            e0.record(ts)
            self.b.execute()
            e1.record(ts)
        self.b.sync()                                           # waits for the stream, collects the device status word
        torch.cuda.synchronize()
        barrier()
        elapsed = time.perf_counter() - t0
        if gc_was:
            gc.enable()
        self.ctx.set_stream(0)
        return elapsed, [a.elapsed_time(b) for a, b in ev]

    def once(self):
        torch, ts = self.torch, self.ts
        self.ctx.set_stream(ts.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ts)
        self.b.execute()
        e1.record(ts)
        self.b.sync()
        self.ctx.set_stream(0)
        return e0.elapsed_time(e1)


def box_fill_GBps(n_bytes=8 << 30, reps=4):
    """What this box's memory system takes from a bare fill kernel (torch's fill_ of an 8 GiB buffer): boxes of the pool differ by +-8 %
    on this path, so the step's rate is reported next to the box's own ceiling."""
    import torch
    x = torch.empty(n_bytes, dtype=torch.uint8, device="cuda")
    x.fill_(46)
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); x.fill_(46); e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None or ms < best else best
    del x
    torch.cuda.empty_cache()
    return n_bytes / (best * 1e-3) / 1e9


def cohort_leg(workload, cohort_samples, h0, h1, steps, warmup, n_threads, verify, temporal=False, host_packed=True, time_host_image=False, label="", barrier=lambda: None,
               settle_ms=0.0):
    """One GPU (this rank's), haplotypes [h0, h1) of the cohort.  The transcript stream is made RESIDENT (v2p_stream_upload), then the
    product's one call -- v2p_batch_build_and_execute: image built on the device, executed -- gives the one-shot numbers and the batch
    every step re-executes; every haplotype's digest is checked against the oracle before anything is timed; optionally the
    host-packed image of the same haplotypes is executed alternately.  Returns the numbers as a dict."""
    import numpy as np
    import torch
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.engine import Context
    cohort = Cohort.preset(workload, n_samples=cohort_samples)
    proteome = cohort.proteome()
    n_haps = h1 - h0
    t0 = time.perf_counter()
    stream = cohort.txstream(h0, h1, n_threads=n_threads)
    t_stream = time.perf_counter() - t0
    A = int(np.ctypeslib.as_array(stream.struct.length, shape=(max(stream.n_tasks, 1),))[:stream.n_tasks].sum(dtype=np.int64)) if stream.n_tasks else 0
    NT = stream.n_tasks
    stream_bytes = stream.nbytes
    sizes = cohort.result_sizes(h0, h1, n_threads=n_threads)
    out_bytes = int(sizes.sum())
    res = {"workload": f"{workload}: {cohort_samples} samples, haplotypes [{h0}, {h1}) x {cohort.n_transcripts} transcripts" + (f" ({label})" if label else ""),
           "haplotypes": n_haps, "aa": A, "tasks": NT, "result_bytes": out_bytes, "stream_bytes": stream_bytes, "stream_generation_s": t_stream}
    ctx = Context(torch.cuda.current_device(), temporal_stores=temporal)
    rs = None
    try:
        ctx.upload_proteome(proteome)
        t0 = time.perf_counter()
        rs = ctx.upload_stream(stream)                          # tables checked on the host, arrays to HBM once: the timed regions below start with it resident
        t_upload = time.perf_counter() - t0
        stream.close()
        assert rs.counts()["out_bytes"] == out_bytes
        # ---- one shot: Task vectors -> result bytes ONCE, in one C-ABI call (v2p_batch_build_and_execute).  Three times: on a batch that
        # owns nothing yet (its arena, descriptor array and scratch are allocated inside the call), on the same batch reset (everything
        # recycled) behind the upload's idle time -- what a host that has just uploaded a cohort sees --, and reset again right behind four
        # executes (the GPU's clocks up: cohorts back to back).  profiles/r05_first_execute.txt: the difference is the shader clock's ramp.
        ts = torch.cuda.Stream()
        ctx.set_stream(ts.cuda_stream)
        b = ctx.batch()
        one = {}
        b.build_and_execute(rs, 0, 0); b.sync()
        i0 = b.oneshot_info()
        one["total_ms_first_call_allocating"] = i0["total_ms"]; one["call_wall_ms_first_call_allocating"] = i0["call_wall_ms"]
        b.reset()
        time.sleep(0.5)
        b.build_and_execute(rs, 0, 0); b.sync()
        i1 = b.oneshot_info()
        # (clocks up: at least four executes and at least ~40 ms of them -- four executes of an eighth of the cohort are 4 ms, which leaves the
        # shader clock where the idle host left it: profiles/r05_first_execute.txt)
        n_busy = max(4, min(400, int(40.0 / max(i1["total_ms"] - i1["build_ms"] - i1["tables_ms"], 0.05))))
        for _ in range(n_busy):
            b.execute()
        b.sync()
        b.reset()
        b.build_and_execute(rs, 0, 0); b.sync()
        i2 = b.oneshot_info()
        ctx.set_stream(0)
        one.update({"total_ms": i1["total_ms"], "tables_ms": i1["tables_ms"], "build_kernels_ms": i1["build_ms"], "first_execute_ms": i1["total_ms"] - i1["build_ms"] - i1["tables_ms"],
                    "call_wall_ms": i1["call_wall_ms"], "aa_per_s": A / (i1["total_ms"] * 1e-3) if A else 0.0,
                    "total_ms_gpu_busy_before": i2["total_ms"], "build_kernels_ms_gpu_busy_before": i2["build_ms"],
                    "aa_per_s_gpu_busy_before": A / (i2["total_ms"] * 1e-3) if A else 0.0,
                    "kernel_choice": i1["kernel"], "n_slices": i1["n_slices"], "stream_upload_s_incl_host_checks": t_upload,
                    "what": "ONE call, v2p_batch_build_and_execute, on the resident per-transcript Task vectors (un-rebased SoA, transcript offsets, alt bytes): "
                            "res_counter per tile / haplotype are tables of the resident stream (made at its upload); one-pass parse; a rich wave image (C3, C4) stays padded -- no compaction pass -- and "
                            "the read-ahead of every phase stages its descriptors in launch order for the stitch kernel (a thin or dense image: compaction beside the row cutter); row cutter, "
                            "XCD order, then every phase of the stitch kernel; the timed steps behind it re-execute the same batch (made dense at its first re-execute); total_ms = HIP events "
                            "from before the first build kernel to behind the last stitch kernel, buffers recycled, 0.5 s of host sleep in front (clocks down); "
                            "*_gpu_busy_before: the same call right behind >= 4 executes and >= 40 ms of them (clocks up)"})
        tiles = b.image_form().get("tiles", False)
        if tiles:
            # a TILE image (deep Task vectors, round 6): 8-byte pieces in the tiles' slots; a tile's record is its count (4 B) and its res_counter (8 B)
            cn = b.counts()
            n_desc_img, n_chunks_img, image_bytes = cn["n_desc"], cn["n_chunks"], 8 * cn["n_desc"] + 12 * cn["n_chunks"]
            hbm_min, n_fused, n_imm = out_bytes + image_bytes + int(proteome.size), None, None
        else:
            desc, chunks, hb = b.download_image()
            hbm_min, n_fused, n_imm = image_stats(desc, chunks, proteome.size, out_bytes)
            n_desc_img, n_chunks_img, image_bytes = int(desc.size), int(chunks.shape[0]), 8 * int(desc.size) + 16 * int(chunks.shape[0])
            del desc, chunks
        # bytes one shot must move: the stream read once, the image written and read once, the result written once
        one_bytes = stream_bytes + 2 * image_bytes + out_bytes + int(proteome.size)
        one["hbm_bytes_min"] = one_bytes
        one["frac_physical"] = one_bytes / (one["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        one["frac_physical_gpu_busy_before"] = one_bytes / (one["total_ms_gpu_busy_before"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        res.update({"descriptors": n_desc_img, "chunks": n_chunks_img, "fused_substitution_descriptors": n_fused, "immediate_descriptors": n_imm,
                    "kernel": "stitchw_kernel (rows image)" if i1["kernel"] == 6 else ("stitch_tiles_kernel (a tile image: <= 16-byte pieces straight from the parse, one workgroup per tile of transcripts; first execute and every later one)" if tiles else
                                                                                         "stitch_pieces_kernel (a dense rows image, re-written as <= 16-byte pieces at its first re-execute; its first execute: stitch_dense_kernel)"),
                    "hbm_bytes_min_per_launch": hbm_min, "algorithmic_bytes_per_launch": 2 * A + 16 * NT, "one_shot": one})
        t = Timed(ctx, b)
        # ---- parity before timing: per-haplotype digests vs the oracle ----
        dig = b.digests()
        verified = None
        if verify != "none":
            every = verify == "all"
            check = list(range(n_haps)) if every else sorted(set(np.linspace(0, n_haps - 1, min(n_haps, 512)).astype(int).tolist()))
            t_v = time.perf_counter()
            want = oracle_digests(workload, cohort_samples, [h0 + i for i in check], n_threads)
            bad = [h0 + i for i in check if int(dig[i]) != want[h0 + i]]
            if bad:
                raise RuntimeError(f"PARITY FAILURE: haplotypes {bad[:8]} of {workload} differ from the oracle ({len(bad)} of {len(check)})")
            verified = {"haplotypes_checked": len(check), "of": n_haps, "every_haplotype": bool(len(check) == n_haps),
                        "digest_of_digests": f"{int(np.bitwise_xor.reduce(dig)) if dig.size else 0:016x}", "oracle_seconds": time.perf_counter() - t_v}
        res["verified"] = verified
        if verified and verified["every_haplotype"]:
            res["_oracle_checked_digests"] = np.array(dig, dtype=np.uint64)     # (every one of them compared with the oracle's above; main() pops it)
        # ---- the host-packed image of the same haplotypes (the packer of rounds 1-3): A/B in one process ----
        hbatch = None
        if host_packed or time_host_image:
            t0 = time.perf_counter()
            img = cohort.pack(h0, h1, n_threads=n_threads)
            t_pack = time.perf_counter() - t0
            hbatch = ctx.batch()
            hbatch.set_packed(img.desc, img.chunks, img.payload, img.hap_out_begin)
            hbatch.finalize()
            hbatch.execute()
            hbatch.sync()
            same = bool(np.array_equal(hbatch.digests(), dig))
            th = Timed(ctx, hbatch)
            ab_h, ab_d = [], []
            for _ in range(3):
                th.once(); t.once()
            for _ in range(9):
                ab_h.append(th.once()); ab_d.append(t.once())
            res["host_packed"] = {"image_build_s": t_pack, "descriptors": int(img.desc.size), "chunks": int(img.chunks.shape[0]), "digests_equal_device_built_image": same,
                                  "ab_execute_ms_host_packed": sorted(ab_h)[4], "ab_execute_ms_device_built": sorted(ab_d)[4],
                                  "what": "the same haplotypes packed on the host (sir_pack.hpp, greedy chunks), executed alternately with the device-built image in this process"}
            del img
        # ---- timed region ----
        timed = Timed(ctx, hbatch) if time_host_image else t
        elapsed, kern_ms = timed.run(steps, warmup, barrier, settle_ms)
        res.update({"elapsed_s": elapsed, "kernel_ms": kern_ms, "image_timed": "host-packed" if time_host_image else "device-built", "clock_settle_ms": timed.settled_ms})
        # ---- parity AFTER timing: the timed steps re-execute the image in its re-execution form (a padded image made dense, its descriptors
        # staged; a dense image from its pieces) -- the arena they leave must be the one that was verified above (Timed.run scribbled it in between)
        if not time_host_image:
            after = b.digests()
            if not np.array_equal(after, dig):
                raise RuntimeError(f"PARITY FAILURE: the arena after the timed steps differs from the verified one ({int((after != dig).sum())} haplotypes of {workload})")
            res["digests_equal_after_timed_steps"] = True
            res["image_form_timed"] = b.image_form()
        if hbatch is not None:
            hbatch.close()
        b.close()
    finally:
        if rs is not None:
            rs.close()
        stream.close()
        ctx.close()
        torch.cuda.empty_cache()
    return res


def traffic_child(args):
    """What live_traffic() runs under rocprofv3: this workload's image through the one call, warm-up executes, then K executes BETWEEN TWO
    digest launches (the markers the parent finds the K steps by).  No oracle, no second leg; prints one JSON line."""
    from vcf2prot_amd import build
    build.build_hip(); build.build_cohort()
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.engine import Context
    samples = args.samples or DEFAULT_SAMPLES[args.scaling][args.workload]
    cohort = Cohort.preset(args.workload, n_samples=samples)
    nt = max(1, min(64, os.cpu_count() or 1))
    stream = cohort.txstream(0, cohort.n_haplotypes, n_threads=nt)
    with Context(0, temporal_stores=args.temporal) as ctx:
        ctx.upload_proteome(cohort.proteome())
        rs = ctx.upload_stream(stream)
        stream.close()
        b = ctx.batch()
        b.build_and_execute(rs, 0, 0); b.sync()
        for _ in range(3):
            b.execute()
        b.sync()
        b.digests()
        for _ in range(args.steps):
            b.execute()
        b.sync()
        b.digests()
        print(json.dumps({"traffic_child": True, "steps": args.steps, "haplotypes": cohort.n_haplotypes}))
        b.close(); rs.close()


def live_traffic(args, steps=3, timeout_s=240):
    """roofline.traffic measured in THIS run: two child runs of this file under `rocprofv3 --pmc` -- FETCH_SIZE, then WRITE_SIZE: separate
    passes, as MI355X_MICROARCH.md's HBM section prescribes; --kernel-trace only beside them -- each building the same image and executing it
    `steps` times between two digest launches.  Bytes per step = (2 x FETCH_SIZE + WRITE_SIZE) x 1 024 summed over the dispatches between the
    markers / steps (FETCH_SIZE doubled: the guide's gfx950 correction for wide coalesced reads).  Returns (bytes per step or None, how)."""
    import csv, glob, shutil, signal, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself being profiled: no profiler inside a profiler"
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="v2p_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "t", "--", sys.executable, os.path.abspath(__file__), "--traffic-child",
               "--workload", args.workload, "--scaling", args.scaling, "--samples", str(args.samples), "--steps", str(steps)] + (["--temporal"] if args.temporal else [])
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
        env["TMPDIR"] = "/tmp"
        try:
            p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd="/tmp", start_new_session=True)
            try:
                out, err = p.communicate(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)            # (the exact process group this call started)
                except OSError:
                    pass
                p.communicate()
                return None, f"the {counter} pass did not finish in {timeout_s} s"
            if p.returncode != 0:
                return None, f"the {counter} pass failed: " + (err or b"").decode(errors="replace")[-300:]
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, f"the {counter} pass wrote no counter file"
            per = {}                                            # dispatch -> (kernel, value): a dispatch's counter comes in one row per XCD / dimension
            for r in csv.DictReader(open(files[0])):
                if r.get("Counter_Name") != counter:
                    continue
                k = int(r["Dispatch_Id"])
                name, v = per.get(k, (r["Kernel_Name"], 0.0))
                per[k] = (name, v + float(r["Counter_Value"]))
            order = sorted(per)
            marks = [k for k in order if "digest" in per[k][0]]
            if len(marks) < 2:
                return None, f"the {counter} pass: the marker launches are not in the trace"
            # (between the markers: the K executes' stitch launches and read-ahead kernels -- and the first marker's own copy-back, which is not a step's)
            between = [per[k] for k in order if marks[-2] < k < marks[-1] and (("stitch" in per[k][0] and "_kernel" in per[k][0]) or "touch_image" in per[k][0])]
            if not between:
                return None, f"the {counter} pass: no stitch launch between the markers"
            got[counter] = sum(v for _, v in between) * 1024.0 / steps
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return 2.0 * got["FETCH_SIZE"] + got["WRITE_SIZE"], (
        f"measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over {steps} executes of the same image in child runs of this "
        f"file; (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B per execute -- FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction "
        f"(fetch {2.0 * got['FETCH_SIZE'] / 1e9:.2f} GB, write {got['WRITE_SIZE'] / 1e9:.2f} GB)")


def main():
    args = parse_args()
    if args.traffic_child:
        return traffic_child(args)
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    if args.dry_run:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            sys.exit("bench.py needs an MI355X: the gpu engine has no CPU fallback")
        if args.collectives == "gloo":
            # (test mode, tests/test_gpu_bench_ranks.py: the whole N-rank path -- launch, per-rank GPU legs, barriers, the size exchange, the line --
            # on a box with fewer GPUs than ranks; ranks share devices and the 16-byte exchange goes over gloo on CPU tensors)
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dev = torch.device("cpu")
        else:
            torch.cuda.set_device(local_rank)
            dev = torch.device("cuda", local_rank)
    dist_on = "RANK" in os.environ and "MASTER_PORT" in os.environ      # launched by torch.distributed.run
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dry_run or args.collectives == "gloo":
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
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.shard import layout_from_sizes, shard_by_bytes

    # ---- synthetic cohort at the Task boundary; this rank's shard --------------------
    samples = args.samples or DEFAULT_SAMPLES[args.scaling][args.workload]
    n_threads = max(1, min(64, (os.cpu_count() or 1) // world))
    if args.scaling == "weak":
        cohort_samples = samples * world
        cohort = Cohort.preset(args.workload, n_samples=cohort_samples)
        h0, h1 = 2 * samples * rank, 2 * samples * (rank + 1)
    else:
        cohort_samples = samples
        cohort = Cohort.preset(args.workload, n_samples=cohort_samples)
        sizes = cohort.result_sizes(0, cohort.n_haplotypes, n_threads=n_threads)
        h0, h1 = shard_by_bytes(sizes.tolist(), world)[rank]              # equal result bytes per rank
    speedup_ref = None
    if world > 1 and args.scaling == "strong" and rank == 0 and args.speedup_ref and not args.dry_run:
        try:                                                   # the same cohort alone on this GPU, one image (the 1-GPU point of the curve)
            r1 = cohort_leg(args.workload, cohort_samples, 0, cohort.n_haplotypes, max(3, min(args.steps, 10)), 3, n_threads, "sample", temporal=args.temporal, host_packed=False)
            speedup_ref = {"ms_per_step_wall": 1e3 * r1["elapsed_s"] / len(r1["kernel_ms"]), "kernel_ms_avg": sum(r1["kernel_ms"]) / len(r1["kernel_ms"]),
                           "haplotypes": r1["haplotypes"], "result_bytes": r1["result_bytes"], "verified": r1["verified"],
                           "method": "the whole cohort as one device-built image on rank 0 alone: host wall-clock around the same execute loop the N-rank run times"}
        except Exception as e:
            speedup_ref = {"error": repr(e)}
    if dist_on:
        dist.barrier()

    if args.dry_run:
        # no kernel: the shard's sizes from the generator, the exchange over gloo
        st = cohort.txstream(h0, h1, n_threads=n_threads)
        A = int(np.ctypeslib.as_array(st.struct.length, shape=(max(st.n_tasks, 1),))[:st.n_tasks].sum(dtype=np.int64)) if st.n_tasks else 0
        leg = {"haplotypes": h1 - h0, "aa": A, "tasks": st.n_tasks, "result_bytes": int(cohort.result_sizes(h0, h1, n_threads=n_threads).sum()),
               "elapsed_s": 1e-3, "kernel_ms": [1.0] * max(args.steps, 1), "verified": None, "descriptors": 0, "chunks": 0, "hbm_bytes_min_per_launch": 0,
               "algorithmic_bytes_per_launch": 2 * A + 16 * st.n_tasks, "workload": f"{args.workload} (dry run)", "kernel": "none (dry run)"}
        st.close()
    else:
        leg = cohort_leg(args.workload, cohort_samples, h0, h1, args.steps, args.warmup, n_threads, args.verify, temporal=args.temporal,
                         host_packed=not args.no_host_packed and world == 1, time_host_image=args.host_image,
                         barrier=(dist.barrier if dist_on else (lambda: None)), settle_ms=args.clock_settle_ms)
    n_haps, out_bytes, A, NT = leg["haplotypes"], leg["result_bytes"], leg["aa"], leg["tasks"]

    # ---- the path's only exchange: {haplotypes, result bytes} of every rank, ONCE per image (sizes do not change between steps) ----
    layout, allgather_us = None, None
    sizes_t = torch.tensor([n_haps, out_bytes], dtype=torch.int64, device=dev)
    all_sizes = torch.zeros(2 * world, dtype=torch.int64, device=dev)
    if dist_on:
        dist.all_gather_into_tensor(all_sizes, sizes_t)        # (the first one sets the communicator up)
        if not args.dry_run:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            dist.all_gather_into_tensor(all_sizes, sizes_t)
        if not args.dry_run:
            torch.cuda.synchronize()
        allgather_us = (time.perf_counter() - t0) / 10 * 1e6
        flat = all_sizes.cpu().tolist()
        layout = layout_from_sizes(rank, flat[0::2], flat[1::2])

    # ---- the timed region of the contract: barrier + synchronize on both sides of K steps, MAX over ranks ----
    elapsed = leg["elapsed_s"]
    kern_ms = leg["kernel_ms"]
    ok_flag = 1.0 if (args.verify == "none" or leg["verified"] is not None) else 0.0
    tot = torch.tensor([elapsed, float(A), float(NT), float(n_haps), float(out_bytes), ok_flag, sum(kern_ms) / len(kern_ms)], dtype=torch.float64, device=dev)
    per_rank = None
    if dist_on:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        gathered = torch.zeros(7 * world, dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(gathered, tot)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0].item())
        A_all, NT_all = float(tot[1].item()), float(tot[2].item())
        g = gathered.cpu().view(world, 7).tolist()
        per_rank = [{"rank": r, "seconds": g[r][0], "aa": int(g[r][1]), "haplotypes": int(g[r][3]), "result_bytes": int(g[r][4]), "verified": bool(g[r][5]),
                     "kernel_ms_avg": g[r][6], "first_haplotype": sum(int(g[q][3]) for q in range(r)), "byte_offset": sum(int(g[q][4]) for q in range(r))} for r in range(world)]
        assert [int(x) for x in layout.n_haps] == [p["haplotypes"] for p in per_rank], "all-gather of result sizes disagrees"
    else:
        A_all, NT_all = float(A), float(NT)

    if rank == 0:
        steps = len(kern_ms)
        avg_ms = sum(kern_ms) / len(kern_ms)
        hbm_min = leg["hbm_bytes_min_per_launch"]
        b_alg = leg["algorithmic_bytes_per_launch"]
        achieved = hbm_min / (avg_ms * 1e-3) / 1e9 if not args.dry_run else 0.0
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath) and not args.dry_run:
            try:
                tj = json.load(open(tpath))
                tj = tj.get(args.workload, tj)
                if tj.get("workload") == args.workload and tj.get("haplotypes") == n_haps:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = ("replayed, not measured in this run: " + str(tj.get("source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command")) +
                                      " (profiles/traffic_latest.json)")
            except Exception:
                traffic = None
        whole = world == 1 and args.scaling == "strong"
        fill_gbps = None
        if not args.dry_run:
            try:
                fill_gbps = box_fill_GBps()
            except Exception:
                fill_gbps = None
        line = {
            "metric": "amino-acids written/sec", "value": A_all * steps / elapsed, "unit": "aa/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / steps,
            "clock_settle": {"ms_rank0": leg.get("clock_settle_ms"), "what": "the checker's digest kernel (v2p_batch_digests over the arena, not a step of the path) keeps the GPU busy this long right before the W warm-up steps: "
                             "the clocks are down after the host's oracle checks and need ~40 ms of such work to come back (profiles/r05_first_execute.txt, r05_clock_settle.txt); the W warm-up "
                             "steps of a whole cohort cover most of that, those of an eighth of it do not -- every point of the scaling curve is timed on the same clock (--clock-settle-ms 0: off)"},
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.workload}: " + (f"{samples} samples/GPU x {world} GPU(s)" if args.scaling == "weak" else
                                                          (f"the whole {samples}-sample cohort as ONE image, one v2p_batch_execute per step" if whole else
                                                           f"one {samples}-sample cohort over {world} GPU(s), equal result bytes per rank"))
                                   + f" ({int(A_all):.3e} aa) x {cohort.n_transcripts} transcripts, SIR Task vectors at the step-6 boundary",
                       "haplotypes_rank0": n_haps, "tasks_rank0": NT, "aa_rank0": A, "chunks_rank0": leg["chunks"], "descriptors_rank0": leg["descriptors"],
                       "descriptor_bytes": 8, "image": leg.get("image_timed", "none") + " (v2p_batch_build_and_execute on the resident transcript stream)" if not args.host_image else "host-packed",
                       "fused_substitution_descriptors_rank0": leg.get("fused_substitution_descriptors"), "immediate_descriptors_rank0": leg.get("immediate_descriptors"),
                       "step": "v2p_batch_execute through the C ABI (ctypes), HIP events on the launch stream",
                       "parallelism": f"haplotype-sharded x{world}, no data-path collective; one all-gather of 16 B per rank per image (RCCL), outside the step loop" if world > 1 else "1 GPU"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_definition": "physical",
                         "traffic": traffic, "traffic_source": traffic_source,
                         "bytes": "achieved = hbm_bytes_min / kernel time: result bytes written once + 8 B per descriptor + 16 B per chunk + alt bytes "
                                  "outside descriptors + the proteome once (reference reads are served by L2 and not counted)",
                         "hbm_bytes_min_per_launch": hbm_min,
                         "algorithmic_bytes_per_launch": b_alg, "algorithmic_GBps": b_alg / (avg_ms * 1e-3) / 1e9,
                         "frac_b_alg": b_alg / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "frac_b_alg_note": "SURVEY 8d's B_alg = 2A + 16N prices one HBM read per residue and 16 B per Task; the design serves reference reads from L2 and fuses "
                                            "Task triples into 8-byte descriptors, so this ratio exceeds 1 without skipping work -- `frac` (physical bytes) is the roofline figure",
                         "kernel": leg["kernel"] + ": a step = the image's phases, each read ahead into the memory-side cache (touch_image_kernel + stitch launches) -- kernel_ms is the whole step",
                         "kernel_ms_avg": avg_ms, "kernel_ms_min": min(kern_ms),
                         "box_fill_GBps": fill_gbps, "frac_of_box_fill": (achieved / fill_gbps) if fill_gbps else None,
                         "box_fill_note": "this box's rate for a bare fill of 8 GiB (torch fill_), measured in this process: boxes of the pool differ by several per cent"},
            "comparable_to_previous_rounds": "since round 4 the N = 1 line is C3 whole (the north star's cohort), strong scaling, the size all-gather outside the step loop; "
                                             "rounds 1-3 reported C2 / weak.  Round 5: same metric and config as round 4; one_shot is now ONE C-ABI call on a resident stream",
            "kernel_only_aa_per_s_rank0": A / (avg_ms * 1e-3),
            "verified": leg["verified"], "stream_generation_s": leg.get("stream_generation_s"),
        }
        for k in ("one_shot", "host_packed", "digests_equal_after_timed_steps", "image_form_timed"):
            if k in leg:
                line[k] = leg[k]
        if per_rank:
            line["per_rank"] = per_rank
            line["world_size_seen_by_rccl"] = int(dist.get_world_size()) if dist_on else 1
            line["collectives"] = "gloo (test mode: ranks may share a GPU)" if args.collectives == "gloo" or args.dry_run else "nccl (RCCL)"
            line["verified_ranks"] = sum(1 for p in per_rank if p["verified"]) if args.verify != "none" else 0
            line["allgather_us"] = allgather_us
        if speedup_ref is not None:
            line["one_gpu_reference"] = speedup_ref
            if "ms_per_step_wall" in speedup_ref:              # same cohort, same loop, same clock: host wall-clock per step on one GPU / on `world` GPUs
                line["speedup_vs_1"] = speedup_ref["ms_per_step_wall"] / (1e3 * elapsed / steps)
                line["speedup_method"] = "host wall-clock per step of the execute loop (barrier + synchronize on both sides), 1 GPU whole cohort / max over ranks"
        if args.dry_run:
            line["data"] = "synthetic (dry run: no kernel was launched, value is meaningless)"
        if world == 1 and not args.dry_run:
            if not args.no_pcie:
                # (right behind the headline leg, before the other cohorts come and go: 0.71 s here, 0.74-0.78 behind them)
                ref_dig = leg.get("_oracle_checked_digests")
                if ref_dig is not None and whole:
                    try:                                       # Task vectors in, host bytes out: the whole cohort of the headline through the stream-fed pipeline
                        torch.cuda.empty_cache()
                        sp = stream_pipeline_leg(args.workload, cohort_samples, n_threads, ref_dig)
                        line["incl_transfers_aa_per_s"] = sp.pop("aa_per_s")
                        line["incl_transfers"] = sp
                    except Exception as e:
                        line["incl_transfers"] = {"error": repr(e)}
            if not args.no_c2 and args.workload == "C3" and not args.samples:
                # BASELINE.json's other single-GPU configurations, whole, the same way (one call on the resident stream, EVERY haplotype's digest
                # against the oracle, then steady-state executes): configs[1] C2 -- the SNV-only cohort of 1 000 samples --, configs[3] C4 -- 2 504
                # samples x the whole proteome, as ONE image on this GPU --, configs[4] C5 -- 100 000 haplotypes of deep Task vectors
                for wl, key in (("C2", "c2_cohort"), ("C4", "c4_cohort"), ("C5", "c5_cohort")):
                    try:
                        torch.cuda.empty_cache()
                        t_leg = time.perf_counter()
                        cl = cohort_leg(wl, DEFAULT_SAMPLES["strong"][wl], 0, 2 * DEFAULT_SAMPLES["strong"][wl], min(args.steps, 50 if wl == "C2" else 20), 5, n_threads, args.verify,
                                        temporal=args.temporal, host_packed=(not args.no_host_packed) and wl == "C2", settle_ms=args.clock_settle_ms)
                        msl = sum(cl["kernel_ms"]) / len(cl["kernel_ms"])
                        cl.update({"ms": msl, "ms_min": min(cl["kernel_ms"]), "aa_per_s": cl["aa"] / (msl * 1e-3), "achieved_GBps": cl["hbm_bytes_min_per_launch"] / (msl * 1e-3) / 1e9,
                                   "frac": cl["hbm_bytes_min_per_launch"] / (msl * 1e-3) / 1e9 / HBM_PEAK_GBS, "steps": len(cl["kernel_ms"]), "leg_seconds": time.perf_counter() - t_leg})
                        del cl["kernel_ms"]
                        cl.pop("_oracle_checked_digests", None)
                        line[key] = cl
                    except Exception as e:                     # never lose the bench line to a secondary leg
                        line[key] = {"error": repr(e)}
            if not args.no_pcie:
                try:
                    c2c = Cohort.preset("C2", n_samples=1000)
                    pc = pcie_inclusive(c2c, 0, c2c.n_haplotypes, n_threads)
                    pc["workload"] = "C2, 1 000 samples"
                    pc["input"] = "host-packed images (packing outside the timed region)"
                    if "incl_transfers" not in line or "error" in line["incl_transfers"]:
                        line["incl_transfers_aa_per_s"] = pc["aa_per_s"]
                    line["incl_transfers_packed_images"] = pc
                except Exception as e:
                    line["incl_transfers_packed_images"] = {"error": repr(e)}
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(cohort, h0, n_haps, os.cpu_count() or 1)
            if not args.no_live_traffic and not args.host_image:
                # LAST (every number above stands whatever happens here): the PMC passes in child processes; on any failure the replayed figure stays
                try:
                    torch.cuda.empty_cache()
                    t_live = time.perf_counter()
                    tb, how = live_traffic(args)
                    if tb is not None:
                        line["roofline"]["traffic_replayed"] = {"bytes": line["roofline"]["traffic"], "source": line["roofline"]["traffic_source"]}
                        line["roofline"]["traffic"] = tb
                        line["roofline"]["traffic_source"] = how
                        line["roofline"]["traffic_over_hbm_bytes_min"] = tb / leg["hbm_bytes_min_per_launch"]
                    else:
                        line["roofline"]["traffic_live_failed"] = how
                    line["roofline"]["traffic_live_seconds"] = time.perf_counter() - t_live
                except Exception as e:                         # noqa: BLE001
                    line["roofline"]["traffic_live_failed"] = repr(e)[:300]
        leg.pop("_oracle_checked_digests", None)
        print(json.dumps(line))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
