"""Arbitrary sample-column text: the three implementations of the BCSQ bitmask reader -- the Python restatement
(oracle/frontend_oracle.py), the C restatement (oracle/frontend_oracle.c) and the GPU parse kernel -- agree on every
string over the alphabet that matters (digits, ',', ':', '.', '-', '+', ' ', letters), abort for abort."""
import os
import random
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import frontend_oracle as F  # noqa: E402

ALPHABET = "0123456789" * 3 + ",,,::..--+ |ax"
N_CSQ = 40
CSQ = ",".join(f"{'missense' if j % 3 else 'synonymous'}|G|T{j}|protein_coding|+|{j + 1}A>{j + 1}C|1A>C" for j in range(N_CSQ))


def random_field(rng):
    kind = rng.random()
    if kind < 0.25:
        return "".join(rng.choice(ALPHABET) for _ in range(rng.randint(0, 14)))
    if kind < 0.5:                                                   # plausible single words, some too big
        return f"0|1:{rng.choice(['', '+', '-', '0', '00'])}{rng.choice([rng.randrange(0, 64), rng.randrange(0, 1 << 31), rng.randrange(1 << 31, 1 << 33)])}"
    n = rng.randint(2, 4)                                             # word lists
    words = [str(rng.choice([0, 0, rng.randrange(0, 1 << 20), rng.randrange(0, 1 << 32), rng.randrange(1 << 32, 1 << 34)])) for _ in range(n)]
    if rng.random() < 0.15:
        words[rng.randrange(n)] = rng.choice(["", "x", "-3", "+7", " 1"])
    return "1|0:0.5:" + ",".join(words)


def oracle_one(field):
    """('ok', h1 indices, h2 indices) or ('abort', code) for one column of a 40-consequence record"""
    try:
        h1, h2 = F.extract_effect_indices(N_CSQ, F.get_bit_mask(field))
    except F.ReferencePanic as p:
        msg = str(p)
        return "abort", (-20 if "invalid bit mask" in msg else (-21 if "unwrap" in msg else -22))
    keep = [j for j in range(N_CSQ) if j % 3]
    return "ok", [i for i in h1 if i in keep], [i for i in h2 if i in keep]


def row_text(fields):
    head = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"S{i}" for i in range(len(fields))) + "\n"
    return head + f"1\t1\t.\tA\tC\t.\t.\tBCSQ={CSQ}\tGT:X:BCSQ\t" + "\t".join(fields) + "\n"


def test_c_restatement_on_arbitrary_text():
    C = F.CFrontend()
    rng = random.Random(12)
    flat = CSQ.split(",")
    sup = np.array([int(F.get_type(c) in F.SUP_TYPE) for c in flat], dtype=np.uint8)
    n_abort = 0
    for trial in range(4000):
        field = random_field(rng)
        raw = np.frombuffer(field.encode(), dtype=np.uint8)
        rc, hb, ids, _ = C.decode(raw if raw.size else np.zeros(1, np.uint8), np.array([0], np.uint64), np.array([len(field)], np.uint64), 1,
                                  np.array([0, N_CSQ], np.uint32), sup, 1)
        want = oracle_one(field)
        if want[0] == "abort":
            assert rc == {-20: 1, -21: 2, -22: 3}[want[1]], field
            n_abort += 1
        else:
            assert rc == 0 and ids[:int(hb[1])].tolist() == want[1] and ids[int(hb[1]):].tolist() == want[2], field
    assert 200 < n_abort < 3000


@pytest.mark.gpu
def test_gpu_parse_on_arbitrary_text(built, gpu_ctx):
    from vcf2prot_amd import _native as N
    from vcf2prot_amd.frontend import VcfIndex, decode_bitmasks
    rng = random.Random(13)
    n_abort = n_ok = 0
    for trial in range(60):
        # mostly well-formed rows so that whole rows decode; every fourth row is wild
        wild = trial % 4 == 3
        fields = []
        for _ in range(rng.choice([1, 7, 300, 900])):
            f = random_field(rng)
            if not wild and oracle_one(f)[0] == "abort":
                f = "0|0:0"
            fields.append(f)
        each = [oracle_one(f) for f in fields]
        idx = VcfIndex(row_text(fields).encode())
        first_bad = next((i for i, e in enumerate(each) if e[0] == "abort"), None)
        if first_bad is not None:
            with pytest.raises(N.V2PError) as e:
                decode_bitmasks(gpu_ctx, idx)
            assert (e.value.code, e.value.index) == (each[first_bad][1], first_bad), fields[first_bad]
            n_abort += 1
        else:
            got = decode_bitmasks(gpu_ctx, idx)
            for s, e in enumerate(each):
                assert got.of(2 * s).tolist() == e[1] and got.of(2 * s + 1).tolist() == e[2], fields[s]
            n_ok += 1
    assert n_ok >= 30 and n_abort >= 10
