#!/usr/bin/env python3
"""Lane-by-lane emulation of stitchw_kernel's LOGIC in plain Python (development tool, CPU only): records, byte map, block owners,
patch table, bulk classification -- and every gather checked against the readable range of its source.  It answers "is the
algorithm right on this image" without a GPU; the parity tests proper compare the kernel itself with the oracle (tests/, -m gpu).

    python tools/emulate_wave.py C3 40 6
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

PAD = 32
NOLIT = 0xFFFF


class Mem:
    """A source space with PAD readable bytes either side; reads outside raise."""

    def __init__(self, data, name):
        self.name, self.n = name, len(data)
        self.buf = np.concatenate([np.full(PAD, 0xEE, np.uint8), np.asarray(data, np.uint8), np.full(PAD, 0xEE, np.uint8)])

    def gather16(self, off):
        if off < -PAD or off + 16 > self.n + PAD:
            raise IndexError(f"gather of 16 bytes at {self.name}[{off}] (length {self.n})")
        return self.buf[off + PAD:off + PAD + 16].copy()


def run_chunk(desc, tb, dn, spaces, out, stats):
    n = (dn >> 48) & 0x7FF
    dst = dn & ((1 << 48) - 1)
    head = dst & 15
    assert n <= 64, n
    recs = []
    pos = head
    for i in range(n):
        d = int(desc[tb + i])
        dlo, dhi = d & 0xFFFFFFFF, d >> 32
        space = dhi >> 30
        snv = (dhi >> 29) == 7
        imm = (not snv) and space == 3
        len1 = ((dlo >> 29) | ((dhi & 0x1FF) << 3)) if snv else (dhi >> 8) & 0x3FFFFF
        nbytes = len1 + 1 + ((dhi >> 9) & 0xFFF) if snv else len1
        src = (dlo & 0x1FFFFFFF) if snv else (((dhi & 0xFF) << 32) | dlo)
        kind = "dots"
        if imm:
            assert len1 <= 5
            kind = "imm"
        elif nbytes and (snv or space != 2):
            kind = "ref" if (snv or space == 0) else "pay"
            assert src + nbytes <= spaces[kind].n, (kind, src, nbytes)
        recs.append(dict(kind=kind, src=src, start=pos, end=pos + nbytes, lit=(pos + len1) if snv else NOLIT, byte=(dhi >> 21) & 0xFF))
        pos += nbytes
    ptotal = pos
    total = ptotal - head
    if total == 0:
        return
    nblk = (ptotal + 15) >> 4
    assert nblk <= 640, nblk
    sent = dict(kind="dots", src=0, start=ptotal, end=0xFFFF, lit=NOLIT, byte=0)
    recs += [sent] * (68 - n)

    def fetch(t, b16):
        if t["kind"] == "imm":
            q = t["start"] - b16
            v = np.zeros(16, np.uint8)
            for k in range(5):
                if 0 <= q + k < 16:
                    v[q + k] = (t["src"] >> (8 * k)) & 0xFF
            return v
        if t["kind"] == "dots":
            return np.full(16, 0x2E, np.uint8)
        v = spaces[t["kind"]].gather16(t["src"] - t["start"] + b16)
        q = t["lit"] - b16
        if t["lit"] != NOLIT and 0 <= q < 16:
            v[q] = t["byte"]
        return v

    def assemble(r, b16):
        hi = min(b16 + 16, ptotal)
        v = fetch(recs[r], b16)
        while recs[r]["end"] < hi:
            r += 1
            t = recs[r]
            ja = t["start"] - b16
            assert 0 <= ja <= 16, (ja, r, b16)
            g = fetch(t, b16)
            v[ja:] = g[ja:]
        return v

    # map: number of records t >= 1 with start <= 16k
    marks = np.zeros(640, np.int64)
    for t in range(1, n):
        kmin = (recs[t]["start"] + 15) >> 4
        if kmin < nblk:
            marks[kmin] += 1
    bmap = np.cumsum(marks)
    patch = {}
    for t in range(n):
        rc = recs[t]
        s, sb16 = rc["start"], rc["start"] & ~15
        prev = recs[t - 1]["start"] if t >= 1 else 0
        owner = t >= 1 and s != sb16 and (prev <= sb16 or (sb16 == 0 and t == 1))
        if owner:
            patch[t] = assemble(t - 1, sb16)
            stats["owners"] += 1
        if rc["lit"] != NOLIT:
            lb16 = rc["lit"] & ~15
            if rc["start"] <= lb16 and rc["end"] >= lb16 + 16:
                patch[64 + t] = fetch(rc, lb16)
                stats["lit_owners"] += 1
    base = dst - head
    for b in range(nblk):
        b16 = b * 16
        r = int(bmap[b])
        t = recs[r]
        if t["end"] < b16 + 16:
            key = r + 1
        elif t["lit"] != NOLIT and 0 <= t["lit"] - b16 < 16:
            key = 64 + r
        else:
            key = None
        if b16 >= head and b16 + 16 <= ptotal:
            if key is None:
                assert t["start"] <= b16 and t["kind"] != "imm", (r, b16, t)
                v = fetch(t, b16)
                stats["plain"] += 1
            else:
                assert key in patch, f"block {b} of chunk at {dst}: patch {key} was never parked (record {r}: {t})"
                v = patch[key]
                stats["patched"] += 1
            out[base + b16:base + b16 + 16] = v
        else:                                   # ragged edge block
            v = assemble(r, b16)
            lo, hi = max(b16, head), min(b16 + 16, ptotal)
            out[base + lo:base + hi] = v[lo - b16:hi - b16]
            stats["ragged"] += 1


def main():
    from gen_util import interpret_image
    from vcf2prot_amd.cohort import Cohort
    preset, h0, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    kw = dict(x.split("=") for x in sys.argv[4:])
    c = Cohort.preset(preset)
    img = c.pack(h0, h0 + n, n_threads=int(kw.pop("threads", 2)), kernel=4, **{k: int(v) for k, v in kw.items()})
    prot = c.proteome()
    want = interpret_image(img.desc, img.chunks, prot, img.payload, img.out_bytes)
    spaces = {"ref": Mem(prot, "proteome"), "pay": Mem(img.payload, "payload")}
    out = np.zeros(img.out_bytes, np.uint8)
    stats = dict(owners=0, lit_owners=0, plain=0, patched=0, ragged=0)
    for tb, dn in img.chunks:
        run_chunk(img.desc, int(tb), int(dn), spaces, out, stats)
    bad = np.nonzero(out != want)[0]
    print(preset, h0, n, "chunks", img.chunks.shape[0], stats, "OK" if bad.size == 0 else f"DIFF at {bad[:10]} ({bad.size} bytes)")
    return 0 if bad.size == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
