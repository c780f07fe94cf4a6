#!/usr/bin/env python3
"""Harvest golden vectors for the VCF front end (BCSQ bitmask decode + grouping per transcript,
SURVEY section 8f rank 4) from the REAL reference.

Build container only (needs /root/reference): handcrafted VCFs go through the reference's prebuilt
binary ``bins/Linux/vcf2prot`` (v0.1.2, -g st, and -g mt where it has enough records).  Every
consequence is a missense on a known 60-residue transcript, so the FASTA the reference writes
shows exactly which consequences it decoded onto which haplotype and how it grouped them.
A case where the reference aborts is recorded as {"panics": true}.

Written to tests/golden/decode_cases.json: the VCF text, the reference FASTA text (inputs) and
the reference's FASTA records per sample or the panic flag (expected outputs).  Data only.

usage: python oracle/make_decode_golden.py [--out tests/golden]
"""
from __future__ import annotations

import argparse
import json
import os
import random
import subprocess
import sys
import tempfile

REF_ROOT = "/root/reference"
BIN = os.path.join(REF_ROOT, "bins/Linux/vcf2prot")
AA = "ACDEFGHIKLMNPQRSTVWY"
HEADER = "##fileformat=VCFv4.2\n##INFO=<ID=BCSQ,Number=.,Type=String,Description=\"synthetic\">\n"


def run_reference(vcf, fasta, outdir, engine):
    env = dict(os.environ)
    for k in ("DEBUG_CPU_EXEC", "INSPECT_TXP", "INSPECT_INS_GEN", "PANIC_INSPECT_ERR", "DEBUG_TXP", "DEBUG_GPU"):
        env.pop(k, None)
    p = subprocess.run([BIN, "-f", vcf, "-r", fasta, "-o", outdir, "-g", engine], env=env, capture_output=True, text=True, timeout=300)
    return p.returncode, p.stdout + p.stderr


def read_fasta_records(path):
    with open(path) as f:
        lines = f.read().split("\n")
    recs, i = [], 0
    while i < len(lines):
        if lines[i].startswith(">"):
            recs.append([lines[i][1:], lines[i + 1] if i + 1 < len(lines) else ""])
            i += 2
        else:
            i += 1
    return sorted(recs)


class Proteome:
    def __init__(self, names, seed):
        rng = random.Random(seed)
        self.seqs = {n: "M" + "".join(rng.choice(AA) for _ in range(59)) for n in names}

    def missense(self, tx, pos, kind="missense", biotype="protein_coding", alt=None):
        """csq string for a substitution at 1-based pos of tx."""
        ref = self.seqs[tx][pos - 1]
        if alt is None:
            alt = AA[(AA.index(ref) + 1 + pos % 7) % 20]
            if alt == ref:
                alt = AA[(AA.index(ref) + 1) % 20]
        return f"{kind}|GENE|{tx}|{biotype}|+|{pos}{ref}>{pos}{alt}|{100 + pos}A>T"

    def fasta(self):
        return "".join(f">{k}\n{v}\n" for k, v in self.seqs.items())


def vcf_text(samples, rows):
    """rows: list of (INFO column text, [sample column text per sample])."""
    out = [HEADER, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples) + "\n"]
    for i, (info, cols) in enumerate(rows):
        out.append(f"7\t{1000 + i}\tv{i}\tC\tT\t100\tPASS\t{info}\tGT:BCSQ\t" + "\t".join(cols) + "\n")
    return "".join(out)


def gt(mask_text, prefix="0|1"):
    return f"{prefix}:{mask_text}"


def words_for(indices_h1, indices_h2, n_words):
    """BCSQ words the way bcftools csq writes them for more than 15 consequences... as the reference reads
    them: word k covers indices 15k..15k+14 (MaskDecoder.rs:123-153)."""
    w = [0] * n_words
    for i in indices_h1:
        w[i // 15] |= 1 << (2 * (i % 15))
    for i in indices_h2:
        w[i // 15] |= 1 << (2 * (i % 15) + 1)
    return w


def build_cases():
    cases = []
    rng = random.Random(2024)

    # 1. one consequence per record, all four masks, plus quirky sample columns that decode to "nothing"
    names = [f"ENST{i:011d}" for i in range(1, 13)]
    P = Proteome(names, 1)
    samples = [f"S{i}" for i in range(6)]
    rows = []
    for i, tx in enumerate(names):
        masks = [(i + s) % 4 for s in range(6)]
        cols = [gt(str(m), f"{m & 1}|{m >> 1}") for m in masks]
        rows.append((f"AC=1;BCSQ={P.missense(tx, 5 + i)}", cols))
    quirks = ["0|1", "0|1:.", "0|1:", "0|1:abc", "0|1:+1", "0|1:003", "0|1:4000000000", "0|1:4000000000,0",
              "0|1:0,0", "0|1:1,0", "0|1:2,0,0", "0|1:0.5:7:3", "1|1:3 ", "1|1: 3", "1|1:3:", "./.:.:1", "0|1:00", "0|1:+0"]
    tx_q = [f"ENSTQ{i:010d}" for i in range(len(quirks) // 6 + 1)]
    P.seqs.update(Proteome(tx_q, 2).seqs)
    for k in range(0, len(quirks), 6):
        cols = (quirks[k:k + 6] + ["0|0:0"] * 6)[:6]
        rows.append((f"BCSQ={P.missense(tx_q[k // 6], 9)};AF=0.5", cols))       # text after BCSQ= keeps ';AF=0.5' (vcf_ds.rs:78)
    cases.append(dict(name="single_consequence_and_quirks", samples=samples, rows=rows, proteome=P))

    # 2. up to 16 consequences in one word, unsupported and non-coding consequences in between
    names = [f"ENST{i:011d}" for i in range(100, 140)]
    P = Proteome(names, 3)
    samples = [f"P{i}" for i in range(8)]
    rows = []
    for r in range(10):
        n = [1, 2, 3, 7, 15, 16, 16, 11, 5, 16][r]
        csq = []
        for j in range(n):
            tx = names[(r * 3 + j) % len(names)]
            kind, bio = "missense", "protein_coding"
            if j % 5 == 3:
                kind = "synonymous"                        # dropped by the SUP_TYPE filter (vcf_ds.rs:272)
            if j % 7 == 5:
                bio = "lincRNA"                            # passes the filter, dropped by split_csq_string (text_parser.rs:38)
            if j % 11 == 9:
                kind = "*missense"
            # (biotype "NMD" is accepted by the source, text_parser.rs:38, but not by the 0.1.2 binary: version
            #  skew, so it is pinned by the source's own test vector in tests/test_frontend_oracle.py instead)
            csq.append(P.missense(tx, 3 + r + 3 * (j % 17), kind, bio))
        cols = []
        for s in range(8):
            m = 0
            for j in range(n):
                m |= rng.randrange(4) << (2 * j) if rng.random() < 0.5 else 0
            m &= (1 << 31) - 1                             # a single word goes through parse::<i32> (text_parser.rs:207)
            if n == 16 and s == 0:
                m = 1 << 30                                # highest haplotype-1 bit of a single word: index 15
            cols.append(gt(str(m)))
        rows.append(("BCSQ=" + ",".join(csq), cols))
    cases.append(dict(name="single_word_multi_consequence", samples=samples, rows=rows, proteome=P))

    # 3. several words: 15 indices per word, 16th pair of word k aliases index 0 of word k+1, zero words
    names = [f"ENST{i:011d}" for i in range(200, 260)]
    P = Proteome(names, 4)
    samples = [f"M{i}" for i in range(8)]
    rows = []
    for r in range(9):
        n = [17, 30, 31, 45, 46, 20, 33, 60, 16][r]
        csq = [P.missense(names[j], 4 + r + (j % 40)) for j in range(n)]
        n_words = (n + 14) // 15
        cols = []
        for s in range(8):
            h1 = [j for j in range(n) if rng.random() < 0.25]
            h2 = [j for j in range(n) if rng.random() < 0.25]
            w = words_for(h1, h2, n_words)
            if s == 1:
                w = w + [0, 0]                             # trailing zero words are stripped (text_parser.rs:236-239)
            if s == 2 and n_words >= 3:
                w[1] = 0                                   # a zero word in the middle stays
            if s == 3:
                w = [0] * len(w)                           # "0,0,0" -> nothing
            if s == 4 and n > 15:
                w[0] |= 1 << 30                            # pair 15 of word 0 == index 15 == pair 0 of word 1 ...
                w[1] &= ~1                                 # ... but not both: the 0.1.2 binary has no drop_replicate (vcf_ds.rs:387-420)
            if s == 5:
                w = [w[0]] + [0] * (len(w) - 1)            # "x,0" collapses to the single-word path
            cols.append(gt(",".join(str(x) for x in w)))
        rows.append(("BCSQ=" + ",".join(csq), cols))
    cases.append(dict(name="multi_word", samples=samples, rows=rows, proteome=P))

    # 4. grouping: several mutations per transcript out of order, duplicates, a transcript id that is a
    #    substring of another one (vcf_tools.rs:91 filters with str::contains).  No duplicate mutations: the 0.1.2
    #    binary predates drop_replicate (vcf_ds.rs:387-420) and writes garbage for them (version skew)
    names = ["ENSTA", "ENSTAB", "ENSTC", "XENSTC", "ENSTD"]
    P = Proteome(names, 5)
    samples = ["G0", "G1", "G2", "G3"]
    rows = []
    plan = [("ENSTA", 40), ("ENSTC", 12), ("ENSTA", 7), ("ENSTAB", 22), ("ENSTA", 21), ("ENSTC", 13), ("XENSTC", 30),
            ("ENSTD", 2), ("ENSTD", 60), ("ENSTD", 33), ("ENSTC", 50), ("ENSTAB", 5)]
    for i, (tx, pos) in enumerate(plan):
        masks = [3, 1, 2, (i * 7) % 4]
        rows.append((f"BCSQ={P.missense(tx, pos)}", [gt(str(m)) for m in masks]))
    cases.append(dict(name="grouping_order_and_substring_names", samples=samples, rows=rows, proteome=P))

    # 5. the aborts
    names = [f"ENST{i:011d}" for i in range(300, 304)]
    P = Proteome(names, 6)
    base = [(f"BCSQ={P.missense(names[0], 5)}", ["0|1:1", "0|0:0"]), (f"BCSQ={P.missense(names[1], 6)}", ["0|1:2", "0|1:1"])]
    for tag, bad in (("negative_single", "0|1:-1"), ("negative_in_list", "0|1:1,-2"), ("index_out_of_range", "0|1:4"),
                     ("empty_word", "0|1:1,,2"), ("non_numeric_word", "0|1:1,x"), ("minus_zero", "0|1:-0"),
                     ("word_above_u32", "0|1:1,4294967296"), ("index_out_of_range_second_word", "0|1:1,4")):
        rows = list(base) + [(f"BCSQ={P.missense(names[2], 7)}", [bad, "0|0:0"])]
        cases.append(dict(name="abort_" + tag, samples=["A0", "A1"], rows=rows, proteome=P))
    rows = list(base) + [(f"BCSQ={P.missense(names[0], 5, alt='W')}", ["0|1:1", "0|0:0"])]     # two different mutations, one position
    cases.append(dict(name="abort_two_mutations_one_position", samples=["A0", "A1"], rows=rows, proteome=P))
    return cases


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
    a = ap.parse_args()
    if not os.path.exists(BIN):
        sys.exit("reference binary not found: this script only runs in the build container")
    out_cases = []
    with tempfile.TemporaryDirectory() as tmp:
        for c in build_cases():
            vcf, fa = vcf_text(c["samples"], c["rows"]), c["proteome"].fasta()
            vp, fp = os.path.join(tmp, "in.vcf"), os.path.join(tmp, "ref.fasta")
            open(vp, "w").write(vcf)
            open(fp, "w").write(fa)
            entry = dict(name=c["name"], samples=c["samples"], vcf=vcf, reference_fasta=fa, oracle_binary="vcf2prot 0.1.2 (bins/Linux)")
            per_engine = {}
            for engine in ("st", "mt"):
                if engine == "mt" and len(c["rows"]) < 2 * (os.cpu_count() or 8):
                    continue                               # vcf_ds.rs:155-156: chunks(len / num_cpus) needs enough records
                od = os.path.join(tmp, "out_" + engine)
                os.makedirs(od, exist_ok=True)
                for f in os.listdir(od):
                    os.remove(os.path.join(od, f))
                rc, log = run_reference(vp, fp, od, engine)
                if rc != 0:
                    per_engine[engine] = dict(panics=True, message=[ln for ln in log.split("\n") if "panicked" in ln][:1])
                else:
                    per_engine[engine] = dict(panics=False, fasta={s: read_fasta_records(os.path.join(od, s + ".fasta")) if os.path.exists(os.path.join(od, s + ".fasta")) else None
                                                                   for s in c["samples"]})
            if "mt" in per_engine and per_engine["mt"] != per_engine["st"]:
                sys.exit(f"{c['name']}: -g st and -g mt disagree")
            entry["engines"] = sorted(per_engine)
            entry.update(per_engine["st"])
            out_cases.append(entry)
            n = sum(len(v or []) for v in entry.get("fasta", {}).values()) if not entry["panics"] else 0
            print(f"{c['name']:45s} panics={entry['panics']}  fasta records={n}  engines={entry['engines']}  {entry.get('message', '')}")
    with open(os.path.join(a.out, "decode_cases.json"), "w") as f:
        json.dump(dict(generator="oracle/make_decode_golden.py", cases=out_cases), f, indent=1)


if __name__ == "__main__":
    main()
