#!/usr/bin/env python3
"""Benchmark of the MI355X SIR executor (vcf2prot step 6) -- driver contract in the task brief.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C2|C3|C4|C5] [--samples S]

One "step" = one pass of the hot path (stitch kernel: K0 fill + K1 in-chunk scan + K2
gather/scatter) over the whole synthetic batch, inputs already resident in HBM.  At N=1
the workload is BASELINE.json configs[1] ("C2": 1 000 samples x 20 k transcripts x ~400 aa,
one missense per transcript => 2 000 haplotypes, A = 1.6e10 residues, N = 1.2e8 tasks).
For N>1 every rank executes its own 1 000-sample shard of a 1 000*N-sample cohort
(weak scaling; haplotypes are independent, the only collective is the all-gather of
per-rank {haplotypes, result bytes} over RCCL).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="C2", choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--samples", type=int, default=0, help="samples per GPU (default: the config's own size, capped to fit HBM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--temporal", action="store_true", help="plain result stores instead of non-temporal")
    ap.add_argument("--max-blocks", type=int, default=0)
    ap.add_argument("--fasta", action="store_true", help="FASTA-emitting image: headers and line feeds fused into the scatter (SURVEY 8f rank 1)")
    ap.add_argument("--tpt", type=int, default=0, help="descriptors per lane (chunks of up to 256*tpt tasks; 0 = library default)")
    ap.add_argument("--cut-align", type=int, default=0)
    ap.add_argument("--chunk-tasks", type=int, default=0)
    ap.add_argument("--chunk-bytes", type=int, default=0)
    ap.add_argument("--dbg", type=int, default=0, help="timing-only kernel ablation (results are wrong; implies --no-verify)")
    ap.add_argument("--xcd-order", type=int, default=-1, help="0: launch chunks in result order instead of dealing them to XCDs by proteome slice")
    return ap.parse_args()


DEFAULT_SAMPLES = {"C2": 1000, "C3": 2000, "C4": 313, "C5": 10000}   # per GPU; C3/C5 full cohorts are processed in HBM-sized batches


def cpu_baseline(cohort, n_threads, budget_s=12.0):
    """Oracle (C restatement of task.rs:38-50 / gir.rs:230-234 / exec.rs:34-40), reference-faithful
    flavour: u32 chars, 32-byte AoS tasks, '.' fill, haplotypes over a thread pool."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from sir_oracle import COracle
    orc = COracle()
    n_h = max(2, min(cohort.n_haplotypes, 2 * n_threads, 48))
    jobs, jobs8, aa = [], [], 0
    for h in range(n_h):
        hap = cohort.haplotype(h)
        t = orc.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
        ref = cohort.ref_tape_u32(h)
        jobs.append((t, ref, hap.alt.astype(np.uint32), np.empty(hap.n_res, dtype=np.uint32)))
        jobs8.append((t, ref.astype(np.uint8), hap.alt, np.empty(hap.n_res, dtype=np.uint8)))
        aa += int(hap.length.sum())
    t1 = orc.mt_execute(jobs, n_threads, wide=True, reps=1)
    reps = max(1, min(200, int(budget_s / max(t1, 1e-3))))
    secs = orc.mt_execute(jobs, n_threads, wide=True, reps=reps)
    t8 = orc.mt_execute(jobs8, n_threads, wide=False, reps=1)
    reps8 = max(1, min(200, int(0.4 * budget_s / max(t8, 1e-3))))
    secs8 = orc.mt_execute(jobs8, n_threads, wide=False, reps=reps8)
    return {"value": aa * reps / secs, "unit": "aa/s", "cores": n_threads, "kind": "port",
            "sample": f"first {n_h} haplotypes of the workload ({aa:.3e} aa) x {reps} passes, u32 chars + 32-B tasks + '.' fill, "
                      f"thread pool over haplotypes (Rayon-MT equivalent)",
            "cpu_best_u8_memcpy": aa * reps8 / secs8}


def main():
    args = parse_args()
    if args.dbg:
        args.no_verify = True
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N")
        args.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the gpu engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_on = "RANK" in os.environ and "MASTER_PORT" in os.environ      # launched by torch.distributed.run
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    # native pieces: built once per node (local rank 0), everybody else waits
    from vcf2prot_amd import build
    if local_rank == 0:
        build.build_all()
        if not args.no_verify or not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import sir_oracle
            sir_oracle.build_c_oracle()
    if dist_on:
        dist.barrier()
    from vcf2prot_amd import _native as N
    from vcf2prot_amd.cohort import Cohort
    lib = N.hip_lib()

    # ---- synthetic cohort at the Task boundary; this rank's shard --------------------
    samples = args.samples or DEFAULT_SAMPLES[args.workload]
    cohort = Cohort.preset(args.workload, n_samples=samples * world)
    h0, h1 = 2 * samples * rank, 2 * samples * (rank + 1)
    n_threads = max(1, (os.cpu_count() or 1) // world)
    t_gen = time.perf_counter()
    img = cohort.pack(h0, h1, n_threads=min(n_threads, 64), chunk_tasks=args.chunk_tasks, chunk_bytes=args.chunk_bytes, fasta=args.fasta, cut_align=args.cut_align)
    t_gen = time.perf_counter() - t_gen
    A, NT = img.n_copy_bytes, img.n_tasks
    b_alg = 2 * A + 16 * NT
    proteome = cohort.proteome()
    n_proteome = proteome.size
    if args.fasta:                                             # resident reference = proteome + record headers
        proteome = np.concatenate([proteome, cohort.fasta_headers()])
        args.no_verify = True                                  # digests are defined on the plain result tape

    def padded(arr):                                           # 16 readable bytes either side (16-byte gathers)
        t = torch.zeros(arr.size + 48, dtype=torch.uint8, device=dev)
        if arr.size:
            t[16:16 + arr.size] = torch.from_numpy(arr).to(dev)
        return t

    d_prot, d_payload = padded(proteome), padded(img.payload)
    d_desc = torch.from_numpy(img.desc.view(np.int64)).to(dev)
    img.chunks = np.ascontiguousarray(img.chunks)
    if args.xcd_order != 0:           # XCD-aware launch order (speed only; chunks are independent)
        rc = lib.v2p_order_chunks_for_xcds(img.chunks.ctypes.data, img.chunks.shape[0], img.desc.ctypes.data, img.desc.size, n_proteome)
        assert rc == 0
    d_chunks = torch.from_numpy(np.ascontiguousarray(img.chunks).view(np.int64)).to(dev)
    d_hap = torch.from_numpy(img.hap_out_begin.view(np.int64)).to(dev)
    out_bytes = img.out_bytes
    d_out = torch.empty(out_bytes + 32, dtype=torch.uint8, device=dev)
    d_status = torch.full((1,), -1, dtype=torch.int64, device=dev)
    n_chunks = int(img.chunks.shape[0])
    n_haps = int(img.hap_out_begin.size - 1)
    assert d_out.data_ptr() % 16 == 0
    stream = torch.cuda.current_stream()
    sizes = torch.tensor([n_haps, out_bytes], dtype=torch.int64, device=dev)
    all_sizes = torch.zeros(2 * world, dtype=torch.int64, device=dev)

    def launch():
        rc = lib.v2p_stitch_launch(ctypes.c_void_p(stream.cuda_stream), d_desc.data_ptr(), d_chunks.data_ptr(), n_chunks,
                                   d_prot.data_ptr() + 16, proteome.size, d_payload.data_ptr() + 16, img.payload.size,
                                   d_out.data_ptr(), out_bytes, d_status.data_ptr(),
                                   (0 if args.temporal else 1) | ((args.tpt or img.tasks_per_lane) << 8) | (args.dbg << 16), args.max_blocks)
        if rc != 0:
            raise RuntimeError(f"v2p_stitch_launch failed: {rc}")
        if dist_on:                                            # the path's only exchange: result sizes for the global offsets
            dist.all_gather_into_tensor(all_sizes, sizes)

    for _ in range(max(args.warmup, 1) if not args.no_verify else args.warmup):
        launch()
    torch.cuda.synchronize()
    if int(d_status.item()) != -1 and not args.dbg:
        sys.exit(f"device reported a task error: status={int(d_status.item()):#x}")

    # ---- parity before timing: per-haplotype digests vs the oracle on a sample --------
    verified = None
    if not args.no_verify:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        from sir_oracle import COracle
        orc = COracle()
        d_dig = torch.zeros(n_haps, dtype=torch.int64, device=dev)
        lib.v2p_digest_launch(ctypes.c_void_p(stream.cuda_stream), d_out.data_ptr(), d_hap.data_ptr(), n_haps, out_bytes, d_dig.data_ptr())
        torch.cuda.synchronize()
        dig = d_dig.cpu().numpy().view(np.uint64)
        check = sorted({0, n_haps // 2, n_haps - 1})
        for i in check:
            hap = cohort.haplotype(h0 + i)
            t = orc.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
            want = orc.gir_execute_u8(t, cohort.ref_tape_u32(h0 + i).astype(np.uint8), hap.alt,
                                      np.full(hap.n_res, ord("."), dtype=np.uint8))
            if int(dig[i]) != orc.digest_u8(want):
                sys.exit(f"PARITY FAILURE: haplotype {h0 + i} differs from the oracle")
        verified = {"haplotypes_checked": [int(h0 + i) for i in check],
                    "digest_of_digests": f"{int(np.bitwise_xor.reduce(dig)):016x}"}

    # ---- timed region -------------------------------------------------------------
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record(stream)
        launch()
        ev[k][1].record(stream)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = [a.elapsed_time(b) for a, b in ev]
    if int(d_status.item()) != -1 and not args.dbg:
        sys.exit(f"device reported a task error: status={int(d_status.item()):#x}")

    tot = torch.tensor([elapsed, float(A), float(NT)], dtype=torch.float64, device=dev)
    if dist_on:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0].item())
        A_all, NT_all = float(tot[1].item()), float(tot[2].item())
    else:
        A_all, NT_all = float(A), float(NT)

    if rank == 0:
        avg_ms = sum(kern_ms) / len(kern_ms)
        achieved = b_alg / (avg_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == args.workload and tj.get("samples_per_gpu") == samples:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "amino-acids written/sec", "value": A_all * args.steps / elapsed, "unit": "aa/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {samples} samples/GPU ({2 * samples} haplotypes) x "
                                   f"{cohort.n_transcripts} transcripts, SIR Task vectors at the step-6 boundary",
                       "haplotypes_per_gpu": n_haps, "tasks_per_gpu": NT, "aa_per_gpu": A, "chunks_per_gpu": n_chunks,
                       "descriptor_bytes": 8, "descriptors_per_lane": args.tpt or img.tasks_per_lane, "parallelism": f"haplotype-sharded x{world}, no data-path collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": b_alg, "kernel": "stitch_kernel", "kernel_ms_avg": avg_ms,
                         "kernel_ms_min": min(kern_ms)},
            "kernel_only_aa_per_s_per_gpu": A / (avg_ms * 1e-3),
            "verified": verified, "image_build_s": t_gen,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cohort, os.cpu_count() or 1)
        print(json.dumps(line))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
