#!/usr/bin/env python3
"""Random multi-sample VCFs answered by the REAL reference (build container only): every consequence kind of
tests/frontend_util.random_vcf, several consequences per record, multi-word masks, 2-5 samples, 20 transcripts of 700
residues.  Kept: the first N files the reference binary (v0.1.2, -g st) gets through, with the FASTA records it wrote.
(Files it aborts on are skipped: with arbitrary amino-acid fields its haplotype-level capacity sums wrap --
haplotype_instruction.rs:161-198 -- which this engine does not reproduce, see DESIGN.md.)

-> tests/golden/random_vcfs.json (data only).   usage: python oracle/make_random_vcf_golden.py [--n 24]
"""
import argparse
import json
import os
import random
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import frontend_util as U  # noqa: E402
import make_decode_golden as G  # noqa: E402

AA = "ACDEFGHIKLMNPQRSTVWY"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=24)
    a = ap.parse_args()
    if not os.path.exists(G.BIN):
        sys.exit("reference binary not found: this script only runs in the build container")
    cases, tried = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        while len(cases) < a.n and tried < 400:
            trial = tried
            tried += 1
            rng = random.Random(1000 + trial)
            n_s = 2 + trial % 4
            # NMD is rejected by the 0.1.2 binary (version skew); start_lost away from position 1 does not occur in csq output
            text = U.random_vcf(700 + trial, 30 + trial % 30, n_s, max_csq=5, n_tx=20).replace("|NMD|", "|protein_coding|").replace("start_lost|", "missense|")
            ref = "".join(f">ENST{i:011d}\n{'M' + ''.join(rng.choice(AA) for _ in range(699))}\n" for i in range(20))
            vp, fp, od = os.path.join(tmp, "in.vcf"), os.path.join(tmp, "ref.fa"), os.path.join(tmp, "o")
            open(vp, "w").write(text)
            open(fp, "w").write(ref)
            os.makedirs(od, exist_ok=True)
            for f in os.listdir(od):
                os.remove(os.path.join(od, f))
            rc, _ = G.run_reference(vp, fp, od, "st")
            if rc != 0:
                continue
            names = [f"S{i}" for i in range(n_s)]
            import hashlib
            recs = {s: (G.read_fasta_records(os.path.join(od, s + ".fasta")) if os.path.exists(os.path.join(od, s + ".fasta")) else []) for s in names}
            # the reference FASTA is regenerated from reference_seed (random.Random(seed), 20 x ('M' + 699 residues of AA));
            # a record is kept as [header, length, first 16 hex digits of sha256(sequence)]
            cases.append(dict(name=f"random_vcf_{trial}", samples=names, vcf=text, reference_seed=1000 + trial, oracle_binary="vcf2prot 0.1.2 (bins/Linux), -g st",
                              fasta={s: [[h, len(q), hashlib.sha256(q.encode()).hexdigest()[:16]] for h, q in v] for s, v in recs.items()}))
    with open(os.path.join(HERE, "..", "tests", "golden", "random_vcfs.json"), "w") as f:
        json.dump(dict(generator="oracle/make_random_vcf_golden.py", tried=tried, cases=cases), f, indent=0)
    print(f"{len(cases)} files kept of {tried} tried; {sum(len(v) for c in cases for v in c['fasta'].values())} FASTA records")


if __name__ == "__main__":
    main()
