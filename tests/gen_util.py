"""Seeded random Task vectors in the reference's shape (test helper, numpy only)."""
from __future__ import annotations

import numpy as np

AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)


def random_tape(rng, n, dtype=np.uint32):
    return AA[rng.integers(0, len(AA), size=n)].astype(dtype)


def random_gir(rng, n_tasks, n_ref, n_alt, mean_len=40, p_zero=0.1, p_gap=0.0, max_len=None, p_alt=0.4):
    """Canonical Task vector: ascending, non-overlapping result ranges; optional gaps.

    Returns dict(code, start_pos, length, start_pos_res, n_res)."""
    length = rng.geometric(1.0 / mean_len, size=n_tasks).astype(np.uint64)
    if max_len:
        length = np.minimum(length, max_len)
    length[rng.random(n_tasks) < p_zero] = 0
    code = (rng.random(n_tasks) < p_alt).astype(np.uint8)
    n_src = np.where(code == 0, n_ref, n_alt).astype(np.uint64)
    length = np.minimum(length, n_src)
    start_pos = (rng.random(n_tasks) * (n_src - length + 1)).astype(np.uint64)
    start_pos = np.minimum(start_pos, n_src - length)
    gaps = np.where(rng.random(n_tasks) < p_gap, rng.integers(1, 20, size=n_tasks), 0).astype(np.uint64)
    ends = np.cumsum(length + gaps)
    start_pos_res = (ends - length).astype(np.uint64)
    tail = int(rng.integers(0, 9)) if p_gap > 0 else 0
    n_res = int(ends[-1]) + tail if n_tasks else tail
    return dict(code=code, start_pos=start_pos, length=length, start_pos_res=start_pos_res, n_res=n_res)


def oracle_run(coracle, g, ref, alt, fill=ord(".")):
    t = coracle.pack_tasks(g["code"], g["start_pos"], g["length"], g["start_pos_res"])
    res = np.full(g["n_res"], fill, dtype=ref.dtype)
    if ref.dtype == np.uint32:
        return coracle.gir_execute(t, ref, alt, res)
    return coracle.gir_execute_u8(t, ref, alt, res)


def _desc_len(d):
    if (d >> 62) == 3 and (d >> 61) & 1:
        return ((d >> 29) & 0xFFF) + 1 + ((d >> 41) & 0xFFF)
    if (d >> 60) == 0xD:
        return ((d >> 29) & 31) + ((d >> 34) & 31) + ((d >> 39) & 31) + 2
    return (d >> 40) & ((1 << 22) - 1)


def interpret_image(desc, chunks, resident, payload, out_bytes):
    """Pure-Python reading of the device image format (vcf2prot_amd/csrc/sir_pack.hpp): what the stitch kernels must write.
    desc: uint64 descriptors; chunks: (n, 2) uint64 {first descriptor, result offset:48 | descriptors:16}."""
    import numpy as np
    out = np.zeros(out_bytes, dtype=np.uint8)
    for tb, dn in chunks:
        nd, dst = (int(dn) >> 48) & 0x7FF, int(dn) & ((1 << 48) - 1)       # the top bits route the chunk to a kernel
        if (int(dn) >> 59) & 1:                   # ROWS image (sir_pack.hpp: CHUNK_CLIP): the chunk starts on a 1 KiB row, skips the bytes of its first descriptor
            tb = int(tb)                          # that belong to the chunk before and drops the bytes of its last one that belong to the next (11 bits each, top of task_begin)
            skip, clip, tb = (tb >> 42) & 0x7FF, tb >> 53, tb & ((1 << 42) - 1)
            assert dst % 1024 == 0
            part = interpret_image(desc[tb:tb + nd], np.array([[0, nd << 48]], dtype=np.uint64), resident, payload, sum(_desc_len(int(d)) for d in desc[tb:tb + nd]))
            part = part[skip:part.size - clip]
            out[dst:dst + part.size] = part
            continue
        for d in desc[int(tb):int(tb) + nd]:
            d = int(d)
            space = d >> 62
            if space == 3 and (d >> 61) & 1:          # fused substitution: reference copy, one literal byte, reference copy one residue on
                src, len1, len2, byte = d & ((1 << 29) - 1), (d >> 29) & 0xFFF, (d >> 41) & 0xFFF, (d >> 53) & 0xFF
                out[dst:dst + len1] = resident[src:src + len1]
                out[dst + len1] = byte
                out[dst + len1 + 1:dst + len1 + 1 + len2] = resident[src + len1 + 1:src + len1 + 1 + len2]
                dst += len1 + 1 + len2
                continue
            if (d >> 60) == 0xD:                      # two substitutions in a row (dense images): copy, byte, copy, byte, copy
                src, l1, l2, l3 = d & ((1 << 29) - 1), (d >> 29) & 31, (d >> 34) & 31, (d >> 39) & 31
                b1, b2 = (d >> 44) & 0xFF, (d >> 52) & 0xFF
                out[dst:dst + l1] = resident[src:src + l1]
                out[dst + l1] = b1
                out[dst + l1 + 1:dst + l1 + 1 + l2] = resident[src + l1 + 1:src + l1 + 1 + l2]
                out[dst + l1 + 1 + l2] = b2
                out[dst + l1 + l2 + 2:dst + l1 + l2 + 2 + l3] = resident[src + l1 + l2 + 2:src + l1 + l2 + 2 + l3]
                dst += l1 + l2 + l3 + 2
                continue
            src, ln = d & ((1 << 40) - 1), (d >> 40) & ((1 << 22) - 1)
            if space == 0:
                out[dst:dst + ln] = resident[src:src + ln]
            elif space == 1:
                out[dst:dst + ln] = payload[src:src + ln]
            elif space == 3:                          # immediate: the source field holds the bytes themselves
                assert 1 <= ln <= 5
                out[dst:dst + ln] = [(src >> (8 * k)) & 0xFF for k in range(ln)]
            else:
                out[dst:dst + ln] = ord(".")
            dst += ln
    return out
