"""Random transcript streams (v2p_txstream) with their expected results in plain numpy -- shared by the device-build fuzz tests and
the host tests of ROWS images.  The preset cohorts have regular Task shapes; these streams do not: empty haplotypes, transcripts
without tasks, zero-length tasks, cells no task covers, alt payloads from 1 byte to longer than a chunk, reference runs from 0 to
several windows, substitution triples at every distance from a window boundary, tasks ending exactly on one."""
import ctypes

import numpy as np


class Stream:
    """A v2p_txstream over numpy arrays (kept alive here)."""

    def __init__(self, hap_tx_begin, tx_off, tx_ref_len, tx_res_len, tx_task_begin, tx_alt_begin, code, sp, ln, sr, alt, header_off=None, header_len=None):
        from vcf2prot_amd._cohort_api import TxStreamBuf
        pad = 64                                     # the builder reads a few entries past the last task / alt byte of a transcript
        self.keep = [np.ascontiguousarray(hap_tx_begin, dtype=np.uint64), np.ascontiguousarray(tx_off, dtype=np.uint64),
                     np.ascontiguousarray(tx_ref_len, dtype=np.uint32), np.ascontiguousarray(tx_res_len, dtype=np.uint32),
                     np.ascontiguousarray(tx_task_begin, dtype=np.uint64), np.ascontiguousarray(tx_alt_begin, dtype=np.uint64),
                     np.concatenate([np.asarray(code, dtype=np.uint8), np.zeros(pad, np.uint8)]),
                     np.concatenate([np.asarray(sp, dtype=np.uint32), np.zeros(pad, np.uint32)]),
                     np.concatenate([np.asarray(ln, dtype=np.uint32), np.zeros(pad, np.uint32)]),
                     np.concatenate([np.asarray(sr, dtype=np.uint32), np.zeros(pad, np.uint32)]),
                     np.concatenate([np.asarray(alt, dtype=np.uint8), np.zeros(pad, np.uint8)])]
        k = self.keep
        s = TxStreamBuf()
        s.n_haps, s.n_tx, s.n_tasks, s.n_alt = len(hap_tx_begin) - 1, len(tx_off), len(code), len(alt)
        P64, P32, P8 = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint8)
        s.hap_tx_begin, s.tx_proteome_off = k[0].ctypes.data_as(P64), k[1].ctypes.data_as(P64)
        s.tx_ref_len, s.tx_res_len = k[2].ctypes.data_as(P32), k[3].ctypes.data_as(P32)
        s.tx_task_begin, s.tx_alt_begin = k[4].ctypes.data_as(P64), k[5].ctypes.data_as(P64)
        s.code, s.start_pos, s.length, s.start_pos_res = k[6].ctypes.data_as(P8), k[7].ctypes.data_as(P32), k[8].ctypes.data_as(P32), k[9].ctypes.data_as(P32)
        s.alt = k[10].ctypes.data_as(P8)
        if header_off is not None:                   # FASTA emit: every transcript's record header in the resident header table (length 0: none)
            k += [np.ascontiguousarray(header_off, dtype=np.uint64), np.ascontiguousarray(header_len, dtype=np.uint32)]
            s.tx_header_off, s.tx_header_len = k[11].ctypes.data_as(P64), k[12].ctypes.data_as(P32)
        self.struct = s


def random_stream(rng, n_haps, n_ref_tx, shape, window, fasta=False):
    """Returns (proteome, Stream, [expected result of every haplotype]); fasta: (proteome, header table, Stream, [expected FASTA text]) --
    every transcript a record `>name_1\n` + residues + `\n` (personalized_genome.rs:90-113), a few without a header (bare residues)."""
    AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    ref_len = rng.integers(1, 3 * window if shape == "long" else 900, size=n_ref_tx)
    ref_off = np.concatenate([[0], np.cumsum(ref_len)])
    proteome = AA[rng.integers(0, AA.size, size=int(ref_off[-1]))]
    hap_tx_begin, tx_off, tx_ref_len, tx_res_len, tx_task_begin, tx_alt_begin = [0], [], [], [], [0], [0]
    code, sp, ln, sr, alt, want = [], [], [], [], [], []
    headers, hoff, hlen = bytearray(b"\n"), [], []
    for h in range(n_haps):
        n_tx = 0 if rng.random() < 0.1 else int(rng.integers(1, 12))
        res_h = []
        for _ in range(n_tx):
            t = int(rng.integers(0, n_ref_tx))
            L = int(ref_len[t])
            ref = proteome[ref_off[t]:ref_off[t] + L]
            tasks, talt, cur_ref, cur_res = [], [], 0, 0
            if rng.random() < 0.05:
                pass                                                      # a transcript without tasks
            else:
                while cur_ref < L:
                    r = rng.random()
                    if shape == "snv":
                        run = int(rng.integers(0, 14))
                    elif shape == "long":
                        run = int(rng.integers(0, 2 * window))
                    else:
                        run = int(rng.integers(0, 200))
                    if r < 0.08:
                        run = max(0, (window - cur_res % window) - int(rng.integers(0, 3)))      # end on / next to a window boundary
                    run = min(run, L - cur_ref)
                    if rng.random() < 0.05:
                        cur_res += int(rng.integers(1, 40))              # cells nothing writes
                    if run or rng.random() < 0.1:
                        tasks.append((0, cur_ref, run, cur_res))          # (a zero-length task now and then)
                    cur_ref += run
                    cur_res += run
                    if cur_ref >= L:
                        break
                    r = rng.random()
                    if r < (0.85 if shape == "snv" else 0.5):             # substitution: one alt byte, the reference goes on one residue later
                        n_alt_b, skip = 1, 1
                    elif r < 0.8:                                         # insertion / delins
                        n_alt_b, skip = int(rng.integers(1, 9 if shape != "long" else window + 100)), int(rng.integers(0, 4))
                    elif r < 0.9:                                         # deletion
                        n_alt_b, skip = 0, int(rng.integers(1, 30))
                    else:                                                 # frameshift-like: a long alt tail, the rest of the reference dropped
                        n_alt_b, skip = int(rng.integers(6, 70)), L
                    if n_alt_b:
                        tasks.append((1, len(talt), n_alt_b, cur_res))
                        talt.extend(AA[rng.integers(0, AA.size, size=n_alt_b)].tolist())
                        cur_res += n_alt_b
                    cur_ref += skip
            res_len = cur_res + (int(rng.integers(1, 20)) if rng.random() < 0.1 else 0)          # trailing uncovered cells
            out = np.full(res_len, ord("."), dtype=np.uint8)
            ta = np.asarray(talt, dtype=np.uint8)
            for c, s_, l_, r_ in tasks:
                out[r_:r_ + l_] = (ref if c == 0 else ta)[s_:s_ + l_]
            if fasta:
                if rng.random() < 0.05:
                    hoff.append(0); hlen.append(0)
                else:
                    name = (">T%d_%s_1\n" % (t, "x" * int(rng.integers(0, 40)))).encode()
                    hoff.append(len(headers)); hlen.append(len(name)); headers += name
                    out = np.concatenate([np.frombuffer(name, dtype=np.uint8), out, np.frombuffer(b"\n", dtype=np.uint8)])
            res_h.append(out)
            tx_off.append(int(ref_off[t])); tx_ref_len.append(L); tx_res_len.append(res_len)
            for c, s_, l_, r_ in tasks:
                code.append(c); sp.append(s_); ln.append(l_); sr.append(r_)
            alt.extend(talt)
            tx_task_begin.append(len(code)); tx_alt_begin.append(len(alt))
        hap_tx_begin.append(len(tx_off))
        want.append(np.concatenate(res_h) if res_h else np.zeros(0, np.uint8))
    if fasta:
        return (proteome, np.frombuffer(bytes(headers), dtype=np.uint8),
                Stream(hap_tx_begin, tx_off, tx_ref_len, tx_res_len, tx_task_begin, tx_alt_begin, code, sp, ln, sr, alt, hoff, hlen), want)
    return proteome, Stream(hap_tx_begin, tx_off, tx_ref_len, tx_res_len, tx_task_begin, tx_alt_begin, code, sp, ln, sr, alt), want


def regular_stream(n_haps, tx_per_hap, dense_every=1, seed=0, ref_len=1000, run=40, sub=8):
    """A large regular stream built with numpy alone (millions of Tasks in a second): every haplotype carries `tx_per_hap` transcripts of
    `ref_len` residues.  A DENSE transcript alternates a reference run of `run` residues with `sub` substituted residues from its alt tape
    (a delins per run + sub residues: about 1024 / (run + sub) * 2 descriptors per KiB of result, none of them fusable or immediate for
    sub > 5); the others are ONE reference copy.  Haplotype h is dense when h % dense_every == 0.  Returns (proteome, Stream, want(h) ->
    expected bytes of haplotype h)."""
    rng = np.random.default_rng(seed)
    AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    n_ref_tx = 64
    proteome2d = AA[rng.integers(0, AA.size, size=(n_ref_tx, ref_len))]
    pairs = ref_len // (run + sub) - 1                       # (run, sub) pairs, then the rest of the reference
    tail0 = pairs * (run + sub)
    # one dense transcript's Tasks, relative to the transcript
    k = np.arange(pairs)
    d_code = np.concatenate([np.stack([np.zeros(pairs, np.uint8), np.ones(pairs, np.uint8)], 1).ravel(), [0]]).astype(np.uint8)
    d_sp = np.concatenate([np.stack([k * (run + sub), k * sub], 1).ravel(), [tail0]]).astype(np.uint32)
    d_ln = np.concatenate([np.stack([np.full(pairs, run), np.full(pairs, sub)], 1).ravel(), [ref_len - tail0]]).astype(np.uint32)
    d_sr = np.concatenate([np.stack([k * (run + sub), k * (run + sub) + run], 1).ravel(), [tail0]]).astype(np.uint32)
    nd = d_code.size
    dense_h = (np.arange(n_haps) % dense_every) == 0
    tx_dense = np.repeat(dense_h, tx_per_hap)
    n_tx = n_haps * tx_per_hap
    tx_id = rng.integers(0, n_ref_tx, size=n_tx)
    n_tasks_tx = np.where(tx_dense, nd, 1)
    tx_task_begin = np.concatenate([[0], np.cumsum(n_tasks_tx)]).astype(np.uint64)
    tx_alt_begin = np.concatenate([[0], np.cumsum(np.where(tx_dense, pairs * sub, 0))]).astype(np.uint64)
    n_tasks = int(tx_task_begin[-1])
    code = np.zeros(n_tasks, np.uint8); sp = np.zeros(n_tasks, np.uint32); ln = np.full(n_tasks, ref_len, np.uint32); sr = np.zeros(n_tasks, np.uint32)
    first = tx_task_begin[:-1][tx_dense].astype(np.int64)
    idx = (first[:, None] + np.arange(nd)[None, :]).ravel()
    code[idx] = np.tile(d_code, first.size); sp[idx] = np.tile(d_sp, first.size); ln[idx] = np.tile(d_ln, first.size); sr[idx] = np.tile(d_sr, first.size)
    alt = AA[rng.integers(0, AA.size, size=int(tx_alt_begin[-1]))]
    hap_tx_begin = (np.arange(n_haps + 1) * tx_per_hap).astype(np.uint64)
    stream = Stream(hap_tx_begin, tx_id.astype(np.uint64) * ref_len, np.full(n_tx, ref_len, np.uint32), np.full(n_tx, ref_len, np.uint32),
                    tx_task_begin, tx_alt_begin, code, sp, ln, sr, alt)
    cols = (k[:, None] * (run + sub) + run + np.arange(sub)[None, :]).ravel()

    def want(h):
        t0, t1 = h * tx_per_hap, (h + 1) * tx_per_hap
        out = proteome2d[tx_id[t0:t1]].copy()
        if dense_h[h]:
            out[:, cols] = alt[int(tx_alt_begin[t0]):int(tx_alt_begin[t1])].reshape(tx_per_hap, pairs * sub)
        return out.ravel()
    return proteome2d.ravel().copy(), stream, want
