#!/usr/bin/env python3
"""Write a synthetic cohort (vcf2prot_amd.cohort, C2-style: every haplotype carries one missense in every transcript) as
a VCF file + reference FASTA, the way BCFtools/csq would annotate it: one record per distinct consequence, GT:BCSQ sample
columns.  Deterministic: the same arguments give the same bytes on any machine (used on both sides of
tests/golden/e2e_200x2000_digests.json).

    python tools/e2e_cohort_vcf.py <samples> <transcripts> <out prefix>
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def write_cohort(n_samples: int, n_transcripts: int, prefix: str, preset: str = "C2", **overrides) -> dict:
    from vcf2prot_amd.cohort import Cohort
    t0 = time.time()
    c = Cohort.preset(preset, n_samples=n_samples, n_transcripts=n_transcripts, **overrides)
    samples = [f"SAMPLE{s:04d}" for s in range(n_samples)]
    prot, off = c.proteome(), c.tx_offsets()
    with open(prefix + "_reference.fasta", "w") as f:
        for t in range(c.n_transcripts):
            f.write(f">{c.tx_name(t)}\n{prot[int(off[t]):int(off[t + 1])].tobytes().decode()}\n")
    records = {}
    n_alt = 0
    for h in range(c.n_haplotypes):
        for t, kind, aa in c.describe(h):
            key = (t, kind, aa)
            m = records.get(key)
            if m is None:
                m = records[key] = bytearray(n_samples)
            m[h // 2] |= 1 << (h % 2)
            n_alt += 1
    cell = ["0|0:0", "1|0:1", "0|1:2", "1|1:3"]
    with open(prefix + ".vcf", "w") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples) + "\n")
        for i, ((t, kind, aa), m) in enumerate(sorted(records.items())):
            f.write(f"7\t{1000 + i}\tv{i}\tC\tT\t100\tPASS\tAC=1;BCSQ={kind}|GENE{t}|{c.tx_name(t)}|protein_coding|+|{aa}|{1000 + t}A>T\tGT:BCSQ\t"
                    + "\t".join(cell[x] for x in m) + "\n")
    return dict(samples=samples, records=len(records), alterations=n_alt, vcf_bytes=os.path.getsize(prefix + ".vcf"), seconds=time.time() - t0)


def sample_digest(path: str) -> str:
    """sha256 over the sorted '>header\\nsequence' records of one FASTA file (the reference writes them in HashMap order)."""
    import hashlib
    lines = open(path).read().split("\n")
    recs = sorted(lines[i] + "\n" + lines[i + 1] for i in range(0, len(lines) - 1, 2))
    return hashlib.sha256("\n".join(recs).encode()).hexdigest()


if __name__ == "__main__":
    print(write_cohort(int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]))
