#!/usr/bin/env python3
"""Bench of the BCSQ bitmask decode on one MI355X: the four kernels of
decode_kernels.hip on a synthetic VCF resident in HBM, next to the C restatement of the reference's decode on the
host cores.  Prints one JSON line.

    python tools/decode_bench.py [--records 100000] [--samples 2504] [--format min|rich] [--density 0.05] [--steps 10]

Algorithmic bytes of one pass = bytes of the sample columns (each read once) + 4 B per emitted consequence id.
"""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from vcf2prot_amd import _native as N  # noqa: E402
from vcf2prot_amd.frontend import VcfIndex, _hip  # noqa: E402

CELLS = {"min": b"0|1:0\t", "rich": b"0|1:17,22:39:99:381,0,512:0\t"}


def make_vcf(R, S, fmt, density, seed):
    rng = np.random.default_rng(seed)
    m = rng.integers(1, 4, size=(R, S), dtype=np.uint8)
    m[rng.random((R, S)) >= density] = 0
    cell = np.frombuffer(CELLS[fmt], dtype=np.uint8)
    body = np.tile(cell, (R, S, 1))
    body[:, :, cell.size - 2] = m + ord("0")
    body[:, :, 0] = (m & 1) + ord("0")
    body[:, :, 2] = (m >> 1) + ord("0")
    body[:, -1, cell.size - 1] = ord("\n")
    fmt_col = "GT:BCSQ" if fmt == "min" else "GT:AD:DP:GQ:PL:BCSQ"
    head = ("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"HG{i:05d}" for i in range(S)) + "\n").encode()
    pre = [f"1\t{1000 + r}\t.\tA\tC\t.\tPASS\tBCSQ=missense|GENE{r % 20011}|ENST{r % 20011:011d}|protein_coding|+|{1 + r // 20011}A>{1 + r // 20011}C|{r}A>C\t{fmt_col}\t".encode()
           for r in range(R)]
    parts = [head]
    for r in range(R):
        parts.append(pre[r])
        parts.append(body[r].tobytes())
    return b"".join(parts), m


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=100000)
    ap.add_argument("--samples", type=int, default=2504)
    ap.add_argument("--format", default="min", choices=list(CELLS))
    ap.add_argument("--density", type=float, default=0.05)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lib", default="", help="A/B: load this build of libvcf2prot_hip.so instead of the in-tree one")
    ap.add_argument("--cpu-records", type=int, default=0, help="records of the CPU baseline sample (0 = sized for ~10 s)")
    a = ap.parse_args()
    if a.lib:
        N.HIP_LIB_PATH = os.path.abspath(a.lib)
    R, S = a.records, a.samples
    t0 = time.time()
    text, m = make_vcf(R, S, a.format, a.density, 3)
    t_gen = time.time() - t0
    t0 = time.time()
    idx = VcfIndex(text)
    t_index = time.time() - t0
    assert (idx.n_records, idx.n_samples) == (R, S)
    col_bytes = int((idx.row_end - idx.row_begin).sum())

    lib = _hip()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    d_text = torch.zeros(len(text) + 512, dtype=torch.uint8, device=dev)
    d_text[256:256 + len(text)] = torch.frombuffer(bytearray(text), dtype=torch.uint8).to(dev)
    d_rb = torch.from_numpy(idx.row_begin.astype(np.int64)).to(dev)
    d_re = torch.from_numpy(idx.row_end.astype(np.int64)).to(dev)
    d_cb = torch.from_numpy(idx.csq_begin.astype(np.int32)).to(dev)
    sup_pairs = np.zeros(R, dtype=np.uint32)
    nb = np.diff(idx.csq_begin.astype(np.int64))
    for j in range(16):
        sel = nb > j
        ok = np.zeros(R, dtype=bool)
        ok[sel] = idx.csq_supported[idx.csq_begin[:-1][sel].astype(np.int64) + j] != 0
        sup_pairs |= np.where(ok, np.uint32(3 << (2 * j)), np.uint32(0)).astype(np.uint32)
    sup_bits = np.packbits(idx.csq_supported.astype(bool), bitorder="little")
    sup_bits = np.concatenate([sup_bits, np.zeros(8, dtype=np.uint8)])[: (sup_bits.size + 7) // 4 * 4].view(np.uint32)
    d_sp = torch.from_numpy(sup_pairs.view(np.int32)).to(dev)
    d_sb = torch.from_numpy(sup_bits.view(np.int32).copy()).to(dev)
    ovf_words = 1 << 20
    ws = int(lib.v2p_decode_workspace_bytes(R, S, ovf_words))
    d_ws = torch.empty(ws + 256, dtype=torch.uint8, device=dev)
    ws_ptr = (d_ws.data_ptr() + 255) & ~255
    n_ids_expected = int((m & 1).astype(bool).sum() + (m >> 1).astype(bool).sum())
    d_hb = torch.zeros(2 * S + 1, dtype=torch.int64, device=dev)
    d_ids = torch.empty(n_ids_expected + 64, dtype=torch.int32, device=dev)
    d_status = torch.zeros(2, dtype=torch.int64, device=dev)

    def launch(phases):
        rc = lib.v2p_decode_launch(ctypes.c_void_p(stream.cuda_stream), d_text.data_ptr() + 256, len(text), d_rb.data_ptr(), d_re.data_ptr(), R, S,
                                   d_cb.data_ptr(), d_sp.data_ptr(), d_sb.data_ptr(), ws_ptr, ovf_words, d_hb.data_ptr(), d_ids.data_ptr(),
                                   n_ids_expected, d_status.data_ptr(), phases)
        assert rc == 0, rc

    names = ("parse", "count", "scan", "emit")
    ms = {k: [] for k in names}
    total = []
    for step in range(a.warmup + a.steps):
        d_status.fill_(-1)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record(stream)
        for k in range(4):
            launch(1 << k)
            ev[k + 1].record(stream)
        torch.cuda.synchronize()
        if step >= a.warmup:
            for k, nm in enumerate(names):
                ms[nm].append(ev[k].elapsed_time(ev[k + 1]))
            total.append(ev[0].elapsed_time(ev[4]))
    st = d_status.cpu().numpy().view(np.uint64)
    assert st[0] == np.uint64(0xFFFFFFFFFFFFFFFF), f"device status {st[0]:x}"
    hb = d_hb.cpu().numpy().astype(np.int64)
    ids = d_ids.cpu().numpy().view(np.uint32)
    if not os.environ.get("V2P_DECODE_DBG"):
        assert int(hb[-1]) == n_ids_expected
        for h in (0, 1):
            assert (np.diff(hb)[h::2] == ((m >> h) & 1).sum(axis=0)).all()
    for s in (() if os.environ.get("V2P_DECODE_DBG") else (0, S // 2, S - 1)):
        for h in (0, 1):
            assert (ids[hb[2 * s + h]:hb[2 * s + h + 1]] == np.nonzero((m[:, s] >> h) & 1)[0]).all()

    t_total = float(np.mean(total)) * 1e-3
    alg_bytes = col_bytes + 4 * n_ids_expected
    out = {"metric": "sample columns decoded/sec", "value": R * S / t_total, "unit": "columns/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": t_total * 1e3, "higher_is_better": True, "dtype": "u8", "data": "synthetic",
           "config": {"workload": f"BCSQ bitmask decode: {R} records x {S} samples, FORMAT {a.format} ({len(CELLS[a.format])} B per column), "
                                  f"{a.density:.0%} non-zero masks", "sample_column_bytes": col_bytes, "ids_emitted": n_ids_expected,
                      "vcf_bytes": len(text)},
           "kernels_ms": {k: float(np.mean(v)) for k, v in ms.items()},
           "roofline": {"bound": "hbm", "achieved": alg_bytes / t_total / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg_bytes / t_total / 8e12,
                        "algorithmic_bytes_per_pass": alg_bytes, "traffic": None,
                        "parse_only_GBs": col_bytes / (float(np.mean(ms["parse"])) * 1e-3) / 1e9},
           "verified": "list lengths of all haplotypes and the ids of 6 haplotypes against numpy", "host_index_s": t_index, "vcf_generate_s": t_gen}
    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r02_decode_summary.json")
    if os.path.exists(prof) and (R, S, a.format, a.density) == (100000, 2504, "min", 0.05):
        # HBM bytes of the kernels from the PMC passes of tools/profile_decode.sh (same workload)
        out["roofline"]["traffic"] = json.load(open(prof)).get("hbm_bytes_per_pass")
    if not a.no_cpu_baseline:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
        from frontend_oracle import CFrontend
        C = CFrontend()
        cores = os.cpu_count() or 1
        rs = a.cpu_records or max(256, min(R, int(2.0e8 * cores / max(S, 1) / 10)))
        t0 = time.time()
        rc, chb, cids, _ = C.decode(idx.text, idx.row_begin[:rs].copy(), idx.row_end[:rs].copy(), S, idx.csq_begin[:rs + 1].copy(), idx.csq_supported, cores)
        dt = time.time() - t0
        assert rc == 0
        for s in (0, S - 1):
            assert (cids[int(chb[2 * s]):int(chb[2 * s + 1])] == np.nonzero(m[:rs, s] & 1)[0]).all()
        out["cpu_baseline"] = {"value": rs * S / dt, "unit": "columns/s", "cores": cores, "kind": "port",
                               "sample": f"first {rs} records x {S} samples, oracle/frontend_oracle.c (get_patient_fields + per-proband decode_back, "
                                         f"threads over record chunks then probands like the Engine::MT arm), {dt:.2f} s"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
