"""The front-end restatement (oracle/frontend_oracle.py: BCSQ bitmask decode + grouping per transcript,
SURVEY section 8f rank 4) against (a) the reference's own unit-test vectors and (b) FASTA written by the
reference binary for handcrafted VCFs (tests/golden/decode_cases.json).  CPU only."""
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import frontend_oracle as F  # noqa: E402


@pytest.fixture(scope="module")
def decode_cases():
    with open(os.path.join(HERE, "golden", "decode_cases.json")) as f:
        return json.load(f)["cases"]


def test_get_bit_mask_vectors():
    """text_parser.rs:509-625 (test_remove_leading_zeros_*, test_parse_fields*, test_get_bit_mask1..10)."""
    assert F.remove_leading_zeros("3,4,0") == "3,4"
    assert F.remove_leading_zeros("3,4,0,1,0") == "3,4,0,1"
    assert F.remove_leading_zeros("0,0") == ""
    assert F.parse_fields("0") == "0$" and F.parse_fields("6") == "6$" and F.parse_fields("6,3") == ""
    for field, want in (("0|0", ""), ("0|0:.:79,0:79:99:.:.:.:0", "0$"), ("0|0:.:37,0:37:72:.:.:.:0", "0$"), ("0|0:0", "0$"),
                        ("0|1:0.541667:26,22:48:PASS:99:577,0,683:..:0.3336:2", "2$"), ("0|1:10", "10$"),
                        ("0|1:0.432432:16,21:37:PASS:99:634,0,417:..:0.1989:10922", "10922$"),
                        ("1|1:.:4,87:91:99:3000,249,0:..:0.4777:15", "15$"),
                        ("1|1:.:4,87:91:99:3000,249,0:..:0.4777:15,32,14", "15,32,14"),
                        ("1|1:.:4,87:91:99:3000,249,0:..:0.4777:15,32,14,0,0,0", "15,32,14")):
        assert F.get_bit_mask(field) == want, field


def test_bitmask_vectors():
    """MaskDecoder.rs:14-31,55-77,161-399."""
    assert F.bitmask_from_string("") is None and F.bitmask_from_string("0$") is None
    assert F.bitmask_from_string("2$") == [2] and F.bitmask_from_string("1024$") == [1024] and F.bitmask_from_string("10922$") == [10922]
    assert F.bitmask_from_string("1024,2048,4096") == [1024, 2048, 4096]
    assert F.bitmask_from_string("1024,0,4096") == [1024, 0, 4096]
    with pytest.raises(F.ReferencePanic):
        F.bitmask_from_string("-1,0,4096")
    assert F.get_indices(F.bitmask_from_string("0$")) is None
    assert F.get_indices([1]) == ([0], [])
    assert F.get_indices([3]) == ([0], [0])
    assert F.get_indices([1024]) == ([5], [])
    assert F.get_indices([1, 1]) == ([0, 15], [])
    assert F.get_indices([3, 3]) == ([0, 15], [0, 15])
    assert F.get_indices([3, 3, 3, 3]) == ([0, 15, 30, 45], [0, 15, 30, 45])


def test_extract_effects_vectors():
    """vcf_ds.rs:596-614."""
    csq = "effect1,effect2,effect3,effect4,effect5,effect6".split(",")
    h1, h2 = F.extract_effect_indices(len(csq), "1")          # the unit test passes the bare number
    assert [csq[i] for i in h1] == ["effect1"] and h2 == []
    h1, h2 = F.extract_effect_indices(len(csq), "3")
    assert [csq[i] for i in h1] == ["effect1"] and [csq[i] for i in h2] == ["effect1"]


def test_csq_and_amino_acid_field_vectors():
    """text_parser.rs:266-300 (split_csq_string), :302-372 (parse_amino_acid_field), mutation_ds.rs:186-."""
    assert F.split_csq_string("stop_gained|RABGEF1|ENST00000484547|NMD|+|32Q>32*|66771993C>T") == ["stop_gained", "ENST00000484547", "32Q>32*"]
    assert F.split_csq_string("5_prime_utr|RABGEF1|ENST00000437078|protein_coding") is None
    assert F.get_type("*missense|ITPRID1|ENST00000409210|protein_coding|+|717C>717Y|31643796G>A") == "*missense"
    m = F.mutation_new("stop_gained|RABGEF1|ENST00000484547|NMD|+|32Q>32*|66771993C>T")
    assert (m.ref_aa_position, m.mut_aa_position, m.ref_aa, m.mut_aa) == (31, 31, "Q", "*")
    m = F.mutation_new("stop_gained|G|T|protein_coding|+|32QK>32NMKLOPLMNBJK*|1C>T")
    assert (m.ref_aa, m.mut_aa) == ("QK", "NMKLOPLMNBJK*")
    assert F.parse_amino_acid_seq_position("32Q") == (32, "Q")
    for i in range(100):
        for j in range(1, 24):
            seq = "ABCDEFGHIJKLMNOPQRSTUVWXYZ"[:j]
            assert F.parse_amino_acid_seq_position(f"{i}{seq}") == (i, seq)
    assert F.parse_amino_acid_seq_position("Test") is None and F.parse_amino_acid_seq_position("-5Q") is None


def test_grouping_vectors():
    """vcf_tools.rs:36-81,103-131,179-225 and vcf_ds.rs:423-470."""
    muts = ["*missense|MAD1L1|Transcript1|protein_coding|-|1R>1H|1936821C>T", "*missense|MAD1L1|Transcript1|protein_coding|-|10R>10H|1936821C>T",
            "*missense|MAD1L1|Transcript2|protein_coding|-|100R>100H|1936821C>T", "*missense|MAD1L1|Transcript2|protein_coding|-|1000R>1000H|1936821C>T",
            "*missense|MAD1L1|Transcript3|protein_coding|-|18R>18H|1936821C>T", "*missense|MAD1L1|Transcript3|protein_coding|-|1993R>1993H|1936821C>T"]
    assert F.get_unique_transcript(muts) == ["Transcript1", "Transcript2", "Transcript3"]
    g = F.group_muts_per_transcript(muts)
    assert [t for t, _ in g] == ["Transcript1", "Transcript2", "Transcript3"]
    assert [[m.ref_aa_position for m in ms] for _, ms in g] == [[0, 9], [99, 999], [17, 1992]]
    t = ["*missense|MAD1L1|ENST00000406869|protein_coding|-|%s|1936821C>T" % a for a in ("1R>1H", "10R>10H", "100R>100H", "1000R>1000H", "200L>200H")]
    got = F.drop_replicate("ENST00000406869", [F.mutation_new(x) for x in t])
    assert [m.ref_aa_position for m in got] == [0, 9, 99, 199, 999]


def apply_missense(seq, muts):
    s = list(seq)
    for m in muts:
        assert m.mut_type in ("missense", "*missense") and len(m.mut_aa) == 1
        s[m.mut_aa_position] = m.mut_aa
    return "".join(s)


def predicted_fasta(case):
    """What the reference should write for the case according to the restatement."""
    ref = {}
    lines = case["reference_fasta"].split("\n")
    for i in range(0, len(lines) - 1, 2):
        ref[lines[i][1:]] = lines[i + 1]
    out = {}
    for name, g1, g2 in F.parse_vcf(case["vcf"]):
        recs = []
        for h, groups in ((1, g1), (2, g2)):
            for tx, muts in groups:
                recs.append([f"{tx}_{h}", apply_missense(ref[tx], muts)])
        out[name] = sorted(recs)
    return out


def test_reference_binary_fasta(decode_cases):
    seen_panics = seen_ok = 0
    for case in decode_cases:
        if case["panics"]:
            with pytest.raises((F.ReferencePanic,)):
                F.parse_vcf(case["vcf"])
            seen_panics += 1
            continue
        want = {s: (v or []) for s, v in case["fasta"].items()}
        got = predicted_fasta(case)
        for s in case["samples"]:
            assert got[s] == want[s], (case["name"], s)
        seen_ok += 1
    assert seen_ok >= 4 and seen_panics >= 9


def test_drop_replicate_semantics_from_source():
    """vcf_ds.rs:387-420 (no vector in the reference's tests and not in the 0.1.2 binary: restated from the source).
    Identical mutations collapse, different mutations on one reference position abort."""
    a = "missense|G|TX|protein_coding|+|12C>12I|1A>T"
    b = "missense|G|TX|protein_coding|+|12C>12W|1A>T"
    c = "missense|G|TX|protein_coding|+|40K>40R|1A>T"
    g = F.group_muts_per_transcript([c, a, a, a])
    assert [(m.ref_aa_position, m.mut_aa) for m in g[0][1]] == [(11, "I"), (39, "R")]
    with pytest.raises(F.ReferencePanic):
        F.group_muts_per_transcript([a, b, c])
    # the 16th pair of word k and the first pair of word k+1 name the same consequence (MaskDecoder.rs:123-153)
    assert F.get_indices([1 << 30, 1]) == ([15, 15], [])


def test_c_restatement_agrees_with_the_python_one(decode_cases):
    """oracle/frontend_oracle.c (the CPU baseline of the decode bench) on the golden VCFs and random ones."""
    import numpy as np
    from frontend_util import oracle_index, oracle_lists, random_vcf
    C = F.CFrontend()
    texts = [c["vcf"] for c in decode_cases] + [random_vcf(s, 80, 11, unique_positions=False) for s in range(4)]
    n_abort = 0
    for text in texts:
        names, recs, split, begin = oracle_index(text)
        raw = text.encode()
        rb, re_, pos = [], [], 0
        for rec in recs:                                  # sample-column ranges of the supported records
            at = raw.index(rec.encode(), pos)
            cols = at + len("\t".join(rec.split("\t")[:9]).encode()) + 1
            rb.append(cols)
            re_.append(at + len(rec.encode()))
            pos = at + 1
        flat = [c for x in split for c in x]
        sup = np.array([int(F.get_type(c) in F.SUP_TYPE) for c in flat], dtype=np.uint8)
        for threads in (1, 3):
            rc, hb, ids, err = C.decode(np.frombuffer(raw, dtype=np.uint8), np.array(rb, dtype=np.uint64), np.array(re_, dtype=np.uint64),
                                        len(names), begin.astype(np.uint32), sup, threads)
            try:
                want = oracle_lists(text)[4]
            except F.ReferencePanic:
                assert rc in (1, 2, 3)
                n_abort += 1
                continue
            assert rc == 0
            assert hb.tolist() == np.concatenate([[0], np.cumsum([len(x) for x in want])]).tolist()
            assert ids.tolist() == [i for x in want for i in x]
    assert n_abort >= 16
