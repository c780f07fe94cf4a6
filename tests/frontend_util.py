"""Helpers of the front-end tests: random VCF text and the oracle's answers in id space."""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import frontend_oracle as F  # noqa: E402

AA = "ACDEFGHIKLMNPQRSTVWY"
HEADER = "##fileformat=VCFv4.2\n##INFO=<ID=BCSQ,Number=.,Type=String,Description=\"synthetic\">\n"
KINDS = ["missense", "*missense", "synonymous", "missense", "stop_gained", "frameshift", "5_prime_utr", "inframe_insertion",
         "missense&inframe_altering", "start_lost", "splice_region", "missense"]


def random_vcf(seed, n_records, n_samples, max_csq=40, n_tx=50, p_zero=0.5, fmt_extra=True, unique_positions=True):
    """VCF text whose masks only select consequences that exist (no aborts).  Returns the text."""
    rng = random.Random(seed)
    tx = [f"ENST{i:011d}" for i in range(n_tx)]
    used = set()
    out = [HEADER, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"S{i}" for i in range(n_samples)) + "\n"]
    for r in range(n_records):
        n = rng.choice([1, 1, 1, 2, 3, 5, 15, 16, 17, 30, 31, max_csq])
        n = min(n, max_csq)
        csq = []
        for j in range(n):
            kind = rng.choice(KINDS)
            bio = rng.choice(["protein_coding"] * 6 + ["NMD", "lincRNA"])
            while True:
                t, pos = rng.choice(tx), rng.randrange(1, 500)
                if not unique_positions or (t, pos) not in used:
                    used.add((t, pos))
                    break
            csq.append(f"{kind}|G{j}|{t}|{bio}|+|{pos}{rng.choice(AA)}>{pos}{rng.choice(AA)}|{pos}A>T")
        if not any(F.is_supported_csq(c) for c in csq):
            csq[0] = f"missense|G|{tx[r % n_tx]}|protein_coding|+|{600 + r}A>{600 + r}C|1A>T"
        info = rng.choice(["", "AC=2;AN=4;"]) + "BCSQ=" + ",".join(csq) + rng.choice(["", ";AF=0.25"])
        cols = []
        n_words = (n + 14) // 15
        for s in range(n_samples):
            u = rng.random()
            if u < p_zero:
                mask = rng.choice(["0", "0", "0", ".", "0,0" if n_words > 1 else "0"])
            elif n <= 16 and rng.random() < 0.8:
                m = 0
                for j in range(min(n, 16)):
                    if rng.random() < 0.3:
                        m |= rng.randrange(1, 4) << (2 * j)
                m &= (1 << 31) - 1
                mask = str(m)
            else:
                w = [0] * n_words
                for j in range(n):
                    if rng.random() < 0.2:
                        w[j // 15] |= rng.randrange(1, 4) << (2 * (j % 15))
                if rng.random() < 0.2:
                    w += [0] * rng.randrange(1, 3)
                mask = ",".join(str(x) for x in w)
            gt = rng.choice(["0|0", "0|1", "1|0", "1|1", "./."])
            extra = rng.choice(["", f":{rng.random():.4f}:{rng.randrange(99)},{rng.randrange(99)}:PASS"]) if fmt_extra else ""
            cols.append(f"{gt}{extra}:{mask}")
        out.append(f"7\t{1000 + r}\tv{r}\tC\tT\t100\tPASS\t{info}\tGT:BCSQ\t" + "\t".join(cols) + "\n")
    return "".join(out)


def oracle_index(text):
    """(names, records, consequences split per record, csq_begin) by the restatement."""
    names, recs = F.read_vcf_text(text)
    split = [F.consequences_of(r).split(",") for r in recs]
    begin = np.concatenate([[0], np.cumsum([len(x) for x in split])]).astype(np.int64)
    return names, recs, split, begin


def oracle_lists(text):
    """(names, records, consequences split per record, csq_begin, [ids per haplotype list 2s+h]) by the restatement."""
    names, recs, split, begin = oracle_index(text)
    _, per = F.get_csq_per_patient(recs, len(names))
    lists = []
    for h1, h2 in per:
        lists.append([int(begin[r]) + i for r, i in h1])
        lists.append([int(begin[r]) + i for r, i in h2])
    return names, recs, split, begin, lists


def lists_to_arrays(lists):
    hap_begin = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.uint64)
    ids = np.array([i for x in lists for i in x], dtype=np.uint32)
    return hap_begin, ids
