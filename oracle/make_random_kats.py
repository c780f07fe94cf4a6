#!/usr/bin/env python3
"""Random single-transcript known-answer tests from the REAL reference (build container only).

Random mutation sets on random reference proteins go through the reference's prebuilt binary (v0.1.2, -g st) with
DEBUG_TXP, exactly like oracle/make_golden.py does for the unit-test cases: per case the Instruction list, the Vec<Task>
dump and the FASTA record the reference produced -- or the fact that it aborted / wrote no record.  They widen the
pinning of steps 4a and 4b (tests/test_random_kats.py) beyond the 36 hand-made cases.

Written to tests/golden/kat_random.json.  Data only (inputs + the reference's outputs).

usage: python oracle/make_random_kats.py [--n 400] [--seed 1]
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as G  # noqa: E402

AA = "ACDEFGHIKLMNPQRSTVWY"


def rseq(rng, n):
    return "".join(rng.choice(AA) for _ in range(n))


def random_case(rng, idx):
    L = rng.randint(30, 110)
    ref = "M" + rseq(rng, L - 1)
    tx = f"ENST{90000000000 + idx:011d}"
    k = rng.choice([1, 1, 2, 2, 3, 4, 6])
    # strictly increasing 1-based positions with room for deletions
    pos, p = [], rng.choice([1, 1, 2, 3, 4, 5, 6])
    tight = idx % 2 == 1                                   # every other case: neighbouring and colliding ranges
    for _ in range(k):
        if p > L - 8:
            break
        pos.append(p)
        p += rng.randint(1, 4) if tight else rng.randint(1, 12) + 6
    muts, terminal = [], False
    for j, P in enumerate(pos):
        last = j == len(pos) - 1
        x = ref[P - 1]
        kinds = ["missense"] * 5 + ["*missense", "inframe_insertion", "*inframe_insertion", "inframe_deletion", "*inframe_deletion"]
        if last:
            kinds += ["frameshift", "*frameshift", "stop_gained", "*stop_gained", "stop_lost", "frameshift&stop_retained",
                      "stop_gained&inframe_altering", "*stop_gained&inframe_altering", "*missense&inframe_altering",
                      "missense&inframe_altering", "inframe_deletion&stop_retained", "stop_lost&frameshift"] * 1
        if j == 0 and len(pos) == 1:
            kinds += ["start_lost", "start_lost&splice_region"]
        kind = rng.choice(kinds)
        base = kind.lstrip("*")
        if base == "missense":
            y = rng.choice([a for a in AA if a != x])
            aa = f"{P}{x}>{P}{y}"
        elif base == "inframe_insertion":
            aa = f"{P}{x}>{P}{x}{rseq(rng, rng.randint(1, 5))}"
        elif base == "inframe_deletion":
            d = rng.randint(1, 5)
            aa = f"{P}{ref[P - 1:P + d]}>{P}{x}"
        elif base == "frameshift":
            aa = f"{P}{ref[P - 1:]}*>{P}{x}{rseq(rng, rng.randint(0, 15))}*"
        elif base == "stop_gained":
            aa = f"{P}{x}>{P}*"
        elif base == "stop_lost":
            aa = f"{L + 1}*>{L + 1}{rseq(rng, rng.randint(1, 9))}"
        elif base == "frameshift&stop_retained":
            aa = f"{P}{ref[P - 1:]}*>{P}{x}{rseq(rng, rng.randint(1, 9))}*"
        elif base == "stop_gained&inframe_altering":
            aa = f"{P}{ref[P - 1:P + rng.randint(1, 6)]}>{P}*"
        elif base == "missense&inframe_altering":
            n = rng.randint(2, 5)
            m = n if rng.random() < 0.6 else rng.randint(2, 6)
            aa = f"{P}{ref[P - 1:P - 1 + n]}>{P}{rseq(rng, m)}"
        elif base == "inframe_deletion&stop_retained":
            aa = f"{P}{ref[P - 1:]}*>{P}*"
        elif base == "stop_lost&frameshift":
            aa = f"{L + 1}*>{L + 1}{rseq(rng, rng.randint(1, 6))}"
        elif base in ("start_lost", "start_lost&splice_region"):
            aa = f"1M>1{rng.choice('KIVT')}"
        else:
            raise AssertionError(kind)
        muts.append(f"{kind}|GENE|{tx}|protein_coding|+|{aa}|{100 + P}A>T")
    return tx, ref, muts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
    a = ap.parse_args()
    if not os.path.exists(G.BIN):
        sys.exit("reference binary not found: this script only runs in the build container")
    rng = random.Random(a.seed)
    cases = []
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(a.n):
            tx, ref, muts = random_case(rng, i)
            r = G.harvest_single_transcript(tx, ref, muts, tmp)
            c = {"name": f"random_{i}", "transcript": tx, "ref": ref, "mutations": muts, "oracle_binary": "vcf2prot 0.1.2 (bins/Linux), -g st"}
            if r["rc"] != 0:
                c["panics"] = True
                msg = [r["panic_line"]] if r.get("panic_line") else []
                c["message"] = msg
                # where the reference gave up: in the executor (task.rs:38-50, the slice bounds of Task::execute) the
                # Instruction list and the Vec<Task> were already printed and are part of the vector
                c["panic_in_executor"] = bool(msg) and "task.rs" in msg[0]
                if c["panic_in_executor"]:
                    c["instructions"], c["tasks"] = r["instructions"], r["tasks"]
                    c["res_len"] = r.get("computed_size")
                    c["alt"] = G.alt_from_instructions(r["instructions"]) if r["tasks"] else ""
            else:
                c["panics"] = False
                seq = dict(r["fasta"] or []).get(f"{tx}_1")
                c["instructions"], c["tasks"] = r["instructions"], r["tasks"]
                c["record"] = seq                       # None: the reference wrote no record for the transcript
                if seq is not None and r["tasks"]:
                    c["alt"] = G.alt_from_instructions(r["instructions"])
            cases.append(c)
    n_p = sum(c["panics"] for c in cases)
    n_none = sum((not c["panics"]) and c["record"] is None for c in cases)
    print(f"{len(cases)} cases: {n_p} aborts, {n_none} without a record, {len(cases) - n_p - n_none} with a record")
    with open(os.path.join(a.out, "kat_random.json"), "w") as f:
        json.dump({"generator": "oracle/make_random_kats.py", "seed": a.seed, "cases": cases}, f, indent=0)


if __name__ == "__main__":
    main()
