#!/usr/bin/env python3
"""Interleaved A/B timing of stitch-kernel variants in ONE process (perf deltas between separate
runs/boxes are within the noise; cdna guide rule 24).

    python tools/ab.py --rounds 15 "cut_align=16" "cut_align=4096" "cut_align=4096,chunk_tasks=192"

Each variant is a comma-separated list of key=value: pack options (chunk_tasks, chunk_bytes,
cut_align, fasta, xcd=0/1) and launch options (nt=0/1, dbg=N).  Prints median/min ms per variant.
"""
import argparse
import ctypes
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vcf2prot_amd import _native as N  # noqa: E402
from vcf2prot_amd.cohort import Cohort  # noqa: E402


def l1_order(chunks, desc, n_prot, R, n_xcd=8):
    """Experiment: XCD slices as usual, but every (haplotype, slice) run is padded with empty chunks to a multiple
    of R entries, so that workgroups R*8 apart in launch order work on the same proteome region of consecutive
    haplotypes (candidates for sharing a CU and its L1)."""
    tb = chunks[:, 0].astype(np.int64)
    key = np.zeros(tb.size, dtype=np.int64)
    found = np.zeros(tb.size, dtype=bool)
    for k in range(4):
        d = desc[np.minimum(tb + k, desc.size - 1)]
        is_ref = ((d >> np.uint64(62)) == 0) & ((d & np.uint64((1 << 40) - 1)) < np.uint64(n_prot))
        take = is_ref & ~found
        key[take] = (d[take] & np.uint64((1 << 40) - 1)).astype(np.int64)
        found |= is_ref
    per = (n_prot + n_xcd - 1) // n_xcd
    bucket = np.minimum(key // per, n_xcd - 1)
    seqs = []
    for x in range(n_xcd):
        idx = np.nonzero(bucket == x)[0]
        k = key[idx]
        starts = np.concatenate([[0], np.nonzero(np.diff(k) < 0)[0] + 1, [idx.size]])
        out = []
        for a, b in zip(starts[:-1], starts[1:]):
            run = idx[a:b]
            pad = (-run.size) % R
            out.append(run)
            if pad:
                out.append(np.full(pad, -1, dtype=np.int64))
        seqs.append(np.concatenate(out) if out else np.zeros(0, np.int64))
    L = max(s.size for s in seqs)
    grid = np.full((L, n_xcd), -1, dtype=np.int64)
    for x, sq in enumerate(seqs):
        grid[:sq.size, x] = sq
    order = grid.reshape(-1)
    res = np.zeros((order.size, 2), dtype=np.uint64)
    ok = order >= 0
    res[ok] = chunks[order[ok]]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--mean-len", type=float, default=0.0, help="override the preset's mean transcript length")
    ap.add_argument("--transcripts", type=int, default=0, help="override the preset's transcript count")
    ap.add_argument("--lib", default="", help="load this build of libvcf2prot_hip.so (tools/build_variant.sh) instead of the in-tree one")
    a = ap.parse_args()
    if a.lib:
        N.HIP_LIB_PATH = os.path.abspath(a.lib)
    # timing-only ablations (dbg=N: results are wrong) exist only in libv2p_bench.so, the V2P_BENCH_VARIANTS build of the engine
    lib = N.bench_lib()        # (the launcher with the packed flag word -- variants, ablations, env switches -- lives in libv2p_bench.so)
    dev = torch.device("cuda", 0)
    over = {}
    if a.mean_len:
        over["mean_len"] = a.mean_len
    if a.transcripts:
        over["n_transcripts"] = a.transcripts
    cohort = Cohort.preset(a.workload, n_samples=a.samples, **over)
    prot = cohort.proteome()
    n_prot = prot.size
    resident = np.concatenate([prot, cohort.fasta_headers()])
    d_prot = torch.zeros(resident.size + 128, dtype=torch.uint8, device=dev)
    d_prot[64:64 + resident.size] = torch.from_numpy(resident).to(dev)
    stream = torch.cuda.current_stream()
    d_status = torch.full((1,), -1, dtype=torch.int64, device=dev)
    vs, max_out = [], 0
    for spec in a.variants:
        kv = dict(x.split("=") for x in spec.split(",") if x)
        pack = {k: int(kv[k]) for k in ("chunk_tasks", "chunk_bytes", "cut_align", "soft_window") if k in kv}
        pack["line_cut"] = bool(int(kv.get("linecut", 1)))
        if int(kv.get("grid", 0)):            # chunks cut on a fixed result grid (the device builder's rule)
            img = cohort.pack_grid(0, cohort.n_haplotypes, int(kv["grid"]), 2 if int(kv.get("var", 0)) in (1, 2) else int(kv.get("kernel", 1)))
        else:
          img = cohort.pack(0, cohort.n_haplotypes, n_threads=min(64, os.cpu_count() or 1), fasta=bool(int(kv.get("fasta", 0))), inline_payload=bool(int(kv.get("imm", 1))), fuse=bool(int(kv.get("fuse", 1))), double=bool(int(kv.get("double", 1))), kernel=(2 if int(kv.get("var", 0)) in (1, 2) else int(kv.get("kernel", 0))), **pack)
        chunks = np.ascontiguousarray(img.chunks)
        rep = int(kv.get("rep", 0))
        if rep:
            # experiment: the first `rep` haplotypes' image replicated over the whole arena -- every copy reads the SAME descriptors
            # (cache-resident), so the HBM descriptor stream disappears while everything else stays
            hb = img.hap_out_begin
            nbytes = int(hb[rep])
            sel = (chunks[:, 1] & np.uint64((1 << 48) - 1)) < np.uint64(nbytes)
            base = chunks[sel]
            n_desc_rep = int(base[:, 0].max()) + 256
            copies = int(img.out_bytes // ((nbytes + 15) // 16 * 16))
            stride = (nbytes + 15) // 16 * 16
            allc = np.concatenate([np.stack([base[:, 0], base[:, 1] + np.uint64(k * stride)], axis=1) for k in range(copies)])
            chunks = np.ascontiguousarray(allc)
            img.desc = img.desc[:n_desc_rep].copy()
            print(f"rep={rep}: {copies} copies of {base.shape[0]} chunks, {n_desc_rep * 8 / 1e6:.2f} MB of descriptors")
        if int(kv.get("gwin", 0)):               # experiment: window-major over the WHOLE proteome (no per-XCD slices): every XCD sweeps the same windows
            tb = chunks[:, 0].astype(np.int64)
            key = np.zeros(tb.size, dtype=np.int64)
            found = np.zeros(tb.size, dtype=bool)
            for k in range(6):
                d = img.desc[np.minimum(tb + k, img.desc.size - 1)]
                snv = (d >> np.uint64(61)) == np.uint64(7)
                src = np.where(snv, d & np.uint64((1 << 29) - 1), d & np.uint64((1 << 40) - 1)).astype(np.int64)
                is_ref = (snv | ((d >> np.uint64(62)) == 0)) & (src < n_prot)
                take = is_ref & ~found
                key[take] = src[take]
                found |= is_ref
            w = max(1, n_prot // int(kv["gwin"]))
            chunks = np.ascontiguousarray(chunks[np.argsort(key // w, kind="stable")])
        elif int(kv.get("l1", 0)):
            chunks = l1_order(chunks, img.desc, n_prot, int(kv["l1"]))
        elif int(kv.get("blocks", 0)) > 1:        # experiment: the XCD / window order inside blocks of haplotypes (equal result bytes), block after block
            nb = int(kv["blocks"])
            dstv = (chunks[:, 1] & np.uint64((1 << 48) - 1)).astype(np.int64)
            blk = np.minimum(dstv * nb // max(int(img.out_bytes), 1), nb - 1)
            parts = []
            for k in range(nb):
                sub_c = np.ascontiguousarray(chunks[blk == k])
                if sub_c.shape[0]:
                    lib.v2p_order_chunks_for_xcds(sub_c.ctypes.data, sub_c.shape[0], img.desc.ctypes.data, img.desc.size, n_prot)
                    pad = (-sub_c.shape[0]) % 8              # (keep every block's first entry on XCD 0: empty chunks)
                    if pad:
                        sub_c = np.concatenate([sub_c, np.zeros((pad, 2), dtype=np.uint64) + np.array([0, 1 << 60], dtype=np.uint64)])
                    parts.append(sub_c)
            chunks = np.ascontiguousarray(np.concatenate(parts))
        elif int(kv.get("xcd", 1)):
            os.environ["V2P_ORDER_WINDOWS"] = str(int(kv.get("sub", 1)))
            if "maxblocks" in kv:
                os.environ["V2P_ORDER_MAX_BLOCKS"] = str(int(kv["maxblocks"]))
            else:
                os.environ.pop("V2P_ORDER_MAX_BLOCKS", None)
            lib.v2p_order_chunks_for_xcds(chunks.ctypes.data, chunks.shape[0], img.desc.ctypes.data, img.desc.size, n_prot)
        desc_arr = img.desc
        if int(kv.get("relayout", 0)):            # experiment: the descriptors physically in launch order (chunk after chunk)
            nd = ((chunks[:, 1] >> np.uint64(48)) & np.uint64(0x7FF)).astype(np.int64)
            tb = chunks[:, 0].astype(np.int64)
            new_tb = np.concatenate([[0], np.cumsum(nd)[:-1]])
            src_idx = np.repeat(tb - new_tb, nd) + np.arange(int(nd.sum()), dtype=np.int64)
            desc_arr = np.ascontiguousarray(img.desc[src_idx])
            chunks = np.ascontiguousarray(np.stack([new_tb.astype(np.uint64), chunks[:, 1]], axis=1))
        d_pay = torch.zeros(img.payload.size + 128, dtype=torch.uint8, device=dev)
        d_pay[64:64 + img.payload.size] = torch.from_numpy(img.payload).to(dev)
        v_bits = int(lib.v2p_stitch_launch_bits(chunks.ctypes.data, chunks.shape[0]))
        d_desc = torch.zeros(desc_arr.size + 16, dtype=torch.int64, device=dev)       # (64 readable bytes either side of the descriptors)
        d_desc[8:8 + desc_arr.size] = torch.from_numpy(desc_arr.view(np.int64)).to(dev)
        v = dict(spec=spec, d_desc=d_desc, n_desc=int(desc_arr.size), d_chunks=torch.from_numpy(chunks.view(np.int64)).to(dev),
                 d_pay=d_pay, n_pay=img.payload.size, phase=kv.get("phase"), gap=int(kv.get("gap", 0)), sync=int(kv.get("sync", 0)), onelaunch=int(kv.get("onelaunch", 0)), sc1=int(kv.get("sc1", 0)), n_chunks=chunks.shape[0], out=img.out_bytes,
                 flags=int(kv.get("nt", 1)) | v_bits | (int(kv.get("var", 0)) << 12) | (int(kv.get("dbg", 0)) << 16) | (int(kv.get("wgs", 0)) << 24) | (int(kv.get("wpg", 0)) << 28), ms=[])
        vs.append(v)
        max_out = max(max_out, img.out_bytes)
    d_out = torch.empty(max_out + 32, dtype=torch.uint8, device=dev)

    def launch(v):
        if v["phase"] is None:
            os.environ.pop("V2P_PHASE_BYTES", None)
        else:
            os.environ["V2P_PHASE_BYTES"] = str(int(float(v["phase"]) * (1 << 20)))      # phase=<MB of image per phase>, 0 = one launch, no touch
        for key, env in (("gap", "V2P_PHASE_GAP_US"), ("sync", "V2P_PHASE_SYNC"), ("onelaunch", "V2P_PHASE_ONE_LAUNCH"), ("sc1", "V2P_WAVE_SC1")):
            if v[key]:
                os.environ[env] = str(v[key])
            else:
                os.environ.pop(env, None)
        rc = lib.v2p_stitch_launch(ctypes.c_void_p(stream.cuda_stream), v["d_desc"].data_ptr() + 64, v["n_desc"], v["d_chunks"].data_ptr(), v["n_chunks"],
                                   d_prot.data_ptr() + 64, resident.size, v["d_pay"].data_ptr() + 64, v["n_pay"],
                                   d_out.data_ptr(), v["out"], d_status.data_ptr(), v["flags"], 0)
        assert rc == 0
    for v in vs:
        d_out.zero_()
        launch(v)
        torch.cuda.synchronize()
        n8 = (v["out"] // 8) * 8
        w = d_out[:n8].view(torch.int64)
        v["chk"] = (int(w.sum().item()) ^ int(w[::7].sum().item() << 1)) & 0xFFFFFFFFFFFF     # all variants of one workload must agree
        launch(v)
    torch.cuda.synchronize()
    # identical variants differ by up to ~6 % with the placement of their buffers in HBM, so every round
    # re-places all input buffers (fresh allocations behind a random-sized spacer) before timing
    rng = np.random.default_rng(1)
    spacers = []
    for rnd in range(a.rounds):
        spacers.append(torch.empty(int(rng.integers(1, 64)) * (1 << 20) + int(rng.integers(0, 4096)) * 256, dtype=torch.uint8, device=dev))
        for v in vs:
            for k in ("d_desc", "d_chunks", "d_pay"):
                v[k] = v[k].clone()
        if len(spacers) > 3:
            spacers.pop(0)
        for v in vs:
            launch(v)
        torch.cuda.synchronize()
        for v in vs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            launch(v)
            e1.record(stream)
            torch.cuda.synchronize()
            v["ms"].append(e0.elapsed_time(e1))
    for v in vs:
        ms = v["ms"]
        print(f"{v['spec']:48s} mean {statistics.mean(ms):.3f} +- {statistics.pstdev(ms):.3f}  median {statistics.median(ms):.3f}  min {min(ms):.3f}  chunks {v['n_chunks']}  chk {v['chk']:012x}")


if __name__ == "__main__":
    main()
