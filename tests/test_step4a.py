"""Step 4a restated in C++ (vcf2prot_amd/csrc/host/instructions.cpp, include/v2p_step4a.h) against the Instruction lists
the reference binary printed for the golden transcripts, the vectors of the reference's unit tests that its current
source satisfies, and the Python restatement on random mutation sets.  CPU only."""
import os
import random
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import frontend_oracle as F  # noqa: E402


def as_tuples(muts):
    return [(m.mut_type, m.ref_aa_position, m.mut_aa_position, m.ref_aa, m.mut_aa) for m in muts]


def both(muts, inspect=True, panic=True):
    """(oracle result, product result); result = list of dicts | 'skip' | 'panic'"""
    from vcf2prot_amd import step4a
    muts = sorted(muts, key=lambda m: m.mut_aa_position)
    try:
        o = F.transcript_instructions("TX", muts, inspect, panic)
        o = "skip" if o is None else [i.as_dict() for i in o]
    except F.ReferencePanic:
        o = "panic"
    rc, ins = step4a.transcript_instructions(as_tuples(muts), (1 if inspect else 0) | (2 if panic else 0))
    p = {0: ins, 1: "skip", 2: "panic"}[rc]
    return o, p


def test_golden_transcripts_give_the_reference_instruction_lists(built, golden):
    for c in golden["cases"]:
        muts = []
        for m in c["mutations"]:
            f = m.split("|")
            f[2] = c["transcript"]
            muts.append("|".join(f))
        groups = F.group_muts_per_transcript(muts)
        assert len(groups) == 1
        o, p = both(groups[0][1])
        assert o == c["instructions"], c["name"]          # the restatement against the reference binary
        assert p == c["instructions"], c["name"]          # the product against the reference binary


def test_reference_unit_vectors(built):
    """instruction.rs test module: the vectors its current source satisfies (code, s_state, positions, length, payload)."""
    def one(kind, aa):
        m = F.mutation_new(f"{kind}|G|TX|protein_coding|+|{aa}|1A>T")
        o, p = both([m])
        assert o == p
        return p[0] if isinstance(p, list) else p
    assert one("missense", "32Q>32R") == dict(code="M", s_state=False, pos_ref=31, pos_res=31, len=1, data="R")
    assert one("*missense", "32Q>32R") == dict(code="N", s_state=True, pos_ref=31, pos_res=31, len=1, data="R")
    assert one("*missense", "32Q>32*") == "panic"
    assert one("inframe_insertion", "125Y>125YRR") == dict(code="I", s_state=False, pos_ref=124, pos_res=124, len=3, data="YRR")
    assert one("*inframe_insertion", "125Y>125YRR")["code"] == "J"
    assert one("inframe_deletion", "115SL>115S") == dict(code="D", s_state=False, pos_ref=114, pos_res=114, len=1, data="S")
    assert one("*inframe_deletion", "115SL>115S")["code"] == "C"
    assert one("frameshift", "40VGLHFWTM*>40VDSTFGQC") == dict(code="F", s_state=False, pos_ref=39, pos_res=39, len=8, data="VDSTFGQC")
    assert one("*frameshift", "40VGLHFWTM*>40VDSTFGQC")["code"] == "R"
    assert one("stop_gained", "217E>217*") == dict(code="G", s_state=False, pos_ref=216, pos_res=216, len=0, data="")
    assert one("stop_lost", "489*>489S") == dict(code="L", s_state=False, pos_ref=488, pos_res=488, len=1, data="S")
    assert one("start_lost", "1M>1K")["code"] == "0"
    assert one("*stop_gained", "217E>217*") == dict(code="X", s_state=True, pos_ref=216, pos_res=216, len=0, data="")
    assert one("*missense&inframe_altering", "188LAY>188LQS") == dict(code="K", s_state=True, pos_ref=187, pos_res=187, len=3, data="LQS")
    assert one("*stop_gained&inframe_altering", "1273KEED>1273")["code"] == "A"
    assert one("frameshift&stop_retained", "65IEREF*>65IENLKTFISKT*") == dict(code="B", s_state=False, pos_ref=64, pos_res=64, len=11, data="IENLKTFISKT")
    assert one("inframe_deletion&stop_retained", "733S*>733*") == dict(code="P", s_state=False, pos_ref=732, pos_res=732, len=1, data="")
    assert one("stop_gained&inframe_altering", "22LESV>22*")["code"] == "T"
    assert one("start_lost&splice_region", "1M>1I")["code"] == "U"
    assert one("inframe_insertion&stop_retained", "192*>192*") == "skip"          # the phi instruction: nothing left


AA = "ACDEFGHIKLMNPQRSTVWY"


def random_mutation(rng, pos):
    kind = rng.choice(F.SUP_TYPE)

    def seq(lo, hi):
        return "".join(rng.choice(AA) for _ in range(rng.randint(lo, hi)))
    ref = rng.choice([seq(1, 1), seq(1, 1), seq(2, 6), seq(1, 4) + "*", "*", ""])
    mut = rng.choice([seq(1, 1), seq(1, 1), seq(2, 6), seq(1, 4) + "*", "*", ""])
    mpos = pos if rng.random() < 0.9 else pos + rng.randint(0, 2)
    return F.mutation_new(f"{kind}|G|TX|protein_coding|+|{pos}{ref}>{mpos}{mut}|1A>T")


def test_random_mutation_sets(built):
    rng = random.Random(4)
    outcomes = {"list": 0, "skip": 0, "panic": 0}
    codes = set()
    for trial in range(3000):
        n = rng.choice([1, 1, 2, 3, 5])
        positions = sorted(rng.sample(range(1, 60), n)) if rng.random() < 0.8 else sorted(rng.choices(range(1, 12), k=n))
        muts = [m for m in (random_mutation(rng, p) for p in positions) if m is not None]
        if not muts:
            continue
        for inspect, panic in ((True, True), (True, False), (False, False)):
            o, p = both(muts, inspect, panic)
            assert o == p, (trial, [m.source for m in muts], inspect, panic)
        outcomes["list" if isinstance(p, list) else p] += 1
        if isinstance(p, list):
            codes |= {i["code"] for i in p}
    assert min(outcomes.values()) > 50
    assert {"M", "N", "I", "J", "D", "C", "F", "R", "G", "X", "L", "0", "U", "K", "Q", "A", "B", "P", "T", "2", "3"} <= codes
