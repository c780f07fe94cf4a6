"""Record index (vcf_index.cpp) against the restatement (readers.rs:96-231, vcf_ds.rs:67-87) on perturbed VCF text:
same records, sample-column ranges and consequence table, or both refuse the file.  CPU only."""
import random

import pytest

from frontend_util import F, oracle_index

CSQS = ["missense|G|T1|protein_coding|+|5A>5C|1A>C", "synonymous|G|T1|protein_coding|+|5A>5A|1A>C", "missense|G|T2|NMD|+|7A>7C", "start_lost|G|T3",
        "*missense|G|T4|lincRNA|+|9A>9C|1A>C", "frameshift|G|T5|protein_coding|+|3ABC*>3AD*|1A>C", "", "@", "stop_gained|G|T6|protein_coding|+|8Q>8*|1A>C|x"]


def random_text(rng):
    n_s = rng.randint(1, 4)
    head = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT" + "".join(f"\tS{i}" for i in range(n_s))
    if rng.random() < 0.1:
        head += "\t"                                                   # trailing tab is popped (readers.rs:128-131)
    if rng.random() < 0.05:
        head = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO" + rng.choice(["", "\tFORMAT"])
    lines = ["##fileformat=VCFv4.2"] if rng.random() < 0.8 else []
    if rng.random() < 0.95:
        lines.append(head)
    for r in range(rng.randint(0, 8)):
        kind = rng.random()
        if kind < 0.05:
            lines.append("#comment in the middle")
            continue
        if kind < 0.08:
            lines.append(rng.choice(["", "1\t2\t3", "x" * 5]))
            continue
        n = rng.randint(1, 4)
        csq = ",".join(rng.choice(CSQS) for _ in range(n))
        info = rng.choice(["", "AC=1;", "AC=1;AF=0.5;", "XBCSQ=1;"]) + rng.choice(["BCSQ=", "BCSQ=", "BCSQ=", "bcsq=", ""]) + csq + rng.choice(["", ";AF=0.1", ";BCSQ=again", ";Z=a=b"])
        cols = ["1", str(r), ".", "A", "C", ".", "PASS", info, "GT:BCSQ"] + [rng.choice(["0|0:0", "0|1:1", ".", "1|1:3", ""]) for _ in range(rng.choice([n_s, n_s, n_s, 0, n_s + 1]))]
        if rng.random() < 0.05:
            cols = cols[:rng.randint(1, 8)]
        lines.append("\t".join(cols))
    nl = rng.choice(["\n", "\n", "\r\n"])
    return nl.join(lines) + (nl if rng.random() < 0.9 else "")


def test_index_on_perturbed_files(built):
    from vcf2prot_amd import _native as N
    from vcf2prot_amd.frontend import VcfIndex
    rng = random.Random(21)
    n_ok = n_refused = 0
    for trial in range(1500):
        text = random_text(rng)
        try:
            names, recs, split, begin = oracle_index(text)
            want = True
            # a supported record without any sample column makes the reference drain(0..9) past the end only if it has
            # fewer than nine columns; with exactly nine it silently mis-pairs columns later: this engine refuses both
            if any(len(r.split("\t")) < 10 for r in recs):
                want = False
        except (F.ReferencePanic, ValueError, IndexError):
            want = False
        try:
            idx = VcfIndex(text.encode())
            got = True
        except N.V2PError:
            got = False
        assert got == want, repr(text)
        if not got:
            n_refused += 1
            continue
        n_ok += 1
        raw = text.encode()
        assert idx.sample_names() == names
        assert [raw[int(b):int(e)].decode() for b, e in zip(idx.row_begin, idx.row_end)] == ["\t".join(r.split("\t")[9:]) for r in recs]
        assert idx.csq_begin.tolist() == begin.tolist()
        flat = [c for x in split for c in x]
        assert [idx.consequence(i) for i in range(idx.n_consequences)] == flat
        assert idx.csq_supported.tolist() == [int(F.get_type(c) in F.SUP_TYPE) for c in flat]
    assert n_ok > 300 and n_refused > 300
