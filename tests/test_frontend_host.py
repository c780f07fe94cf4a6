"""Host pieces of include/v2p_frontend.h against the restatement (oracle/frontend_oracle.py) -- CPU only:
the record index (vcf_index.cpp) and the grouping per transcript (group_muts.cpp)."""
import json
import os

import numpy as np
import pytest

from frontend_util import F, lists_to_arrays, oracle_index, oracle_lists, random_vcf

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def decode_cases():
    with open(os.path.join(HERE, "golden", "decode_cases.json")) as f:
        return json.load(f)["cases"]


def check_index(text):
    from vcf2prot_amd.frontend import VcfIndex
    names, recs, split, begin = oracle_index(text)
    idx = VcfIndex(text.encode())
    assert idx.sample_names() == names
    assert idx.n_records == len(recs)
    raw = text.encode()
    for r, rec in enumerate(recs):
        assert raw[int(idx.row_begin[r]):int(idx.row_end[r])].decode() == "\t".join(rec.split("\t")[9:])
    assert idx.csq_begin.tolist() == begin.tolist()
    flat = [c for x in split for c in x]
    assert [idx.consequence(i) for i in range(idx.n_consequences)] == flat
    assert idx.csq_supported.tolist() == [int(F.get_type(c) in F.SUP_TYPE) for c in flat]
    return idx


def test_index_on_the_golden_vcfs(built, decode_cases):
    for c in decode_cases:
        check_index(c["vcf"])          # (the aborts of these cases happen later, in the decode or the grouping)


def test_index_record_filter_and_quirks(built):
    """readers.rs:185-231: only BCSQ entries with six separators and a supported type keep a record;
    vcf_ds.rs:78: the consequence text runs to the next "BCSQ=" or the end of INFO."""
    head = "##x\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\n"
    rows = ["1\t1\t.\tA\tC\t.\t.\tAC=1\tGT\t0|0\t0|1",                                                   # no BCSQ
            "1\t2\t.\tA\tC\t.\t.\tBCSQ=synonymous|G|T1|protein_coding|+|5A>5A|1A>C\tGT:BCSQ\t0|0:0\t0|1:1",  # unsupported type
            "1\t3\t.\tA\tC\t.\t.\tBCSQ=missense|G|T1|protein_coding|+|5A>5C\tGT:BCSQ\t0|0:0\t0|1:1",         # five separators
            "1\t4\t.\tA\tC\t.\t.\tXBCSQ=1;BCSQ=synonymous|G|T1|protein_coding|+|5A>5A|1A>C,missense|G|T2|protein_coding|+|7A>7C|1A>C;AF=0.5\tGT:BCSQ\t0|0:0\t0|1:4",
            "1\t5\t.\tA\tC\t.\t.\tBCSQ=missense|G|T3|protein_coding|+|9A>9C|1A>C\tGT:BCSQ\t0|1:1\t1|1:3\r"]
    text = head + "\n".join(rows) + "\n"
    idx = check_index(text)
    assert idx.n_records == 2 and idx.n_consequences == 2
    # row 4: INFO contains "XBCSQ=1;BCSQ=": split("BCSQ=")[1] is the text between the two occurrences
    assert idx.consequence(0) == "1;"
    from vcf2prot_amd import _native as N
    from vcf2prot_amd.frontend import VcfIndex
    for bad in ("", "##only meta\n", head, head + "1\t2\t3\n", head.replace("\tA\tB", "") + rows[4] + "\n"):
        with pytest.raises(N.V2PError):
            VcfIndex(bad.encode())


def compare_groups(text, n_threads=3):
    from vcf2prot_amd.frontend import Groups, HaplotypeLists, VcfIndex
    names, recs, split, begin, lists = oracle_lists(text)
    flat = [c for x in split for c in x]
    idx = VcfIndex(text.encode())
    hb, ids = lists_to_arrays(lists)
    g = Groups(idx, HaplotypeLists(hb, ids), n_threads)
    for h, lst in enumerate(lists):
        want = [(t, [m.source for m in ms]) for t, ms in F.group_muts_per_transcript([flat[i] for i in lst])]
        got = [(t, [flat[i] for i in members]) for t, members in g.of(h)]
        assert got == want, h
    return g


def test_grouping_on_the_golden_vcfs(built, decode_cases):
    from vcf2prot_amd import _native as N
    n_ok = n_abort = 0
    for c in decode_cases:
        try:
            oracle_lists(c["vcf"])
        except F.ReferencePanic:
            continue                                      # decode abort: GPU tests
        try:
            F.parse_vcf(c["vcf"])
        except F.ReferencePanic:
            with pytest.raises(N.V2PError) as e:          # vcf_ds.rs:411
                compare_groups(c["vcf"])
            assert e.value.code == -27
            n_abort += 1
            continue
        compare_groups(c["vcf"])
        n_ok += 1
    assert n_ok >= 4 and n_abort >= 1


def test_grouping_reference_vectors(built):
    """vcf_tools.rs:179-225 (test_group_muts_per_transcript) through the C ABI."""
    from vcf2prot_amd.frontend import Groups, HaplotypeLists, VcfIndex
    muts = ["*missense|MAD1L1|Transcript1|protein_coding|-|1R>1H|1936821C>T", "*missense|MAD1L1|Transcript1|protein_coding|-|10R>10H|1936821C>T",
            "*missense|MAD1L1|Transcript2|protein_coding|-|100R>100H|1936821C>T", "*missense|MAD1L1|Transcript2|protein_coding|-|1000R>1000H|1936821C>T",
            "*missense|MAD1L1|Transcript3|protein_coding|-|18R>18H|1936821C>T", "*missense|MAD1L1|Transcript3|protein_coding|-|1993R>1993H|1936821C>T"]
    text = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\n" + "".join(f"1\t{i}\t.\tA\tC\t.\t.\tBCSQ={m}\tGT:BCSQ\t0|1:1\n" for i, m in enumerate(muts))
    idx = VcfIndex(text.encode())
    g = Groups(idx, HaplotypeLists(np.array([0, 6, 6], dtype=np.uint64), np.arange(6, dtype=np.uint32)))
    got = g.of(0)
    assert [t for t, _ in got] == ["Transcript1", "Transcript2", "Transcript3"]
    assert [[int(g.mutations[i]["ref_aa_position"]) for i in m] for _, m in got] == [[0, 9], [99, 999], [17, 1992]]
    assert g.of(1) == []


def test_grouping_duplicates_substrings_and_invalid_fields(built):
    """drop_replicate (vcf_ds.rs:387-420), str::contains matching (vcf_tools.rs:91), Mutation::new failures."""
    rows = ["missense|G|TXA|protein_coding|+|40K>40R|1A>T", "missense|G|TXAB|protein_coding|+|7K>7R|1A>T", "missense|G|TXA|protein_coding|+|12C>12I|1A>T",
            "missense|G|TXA|protein_coding|+|12C>12I|1A>T", "missense|TXA|TXC|protein_coding|+|3K>3R|1A>T",       # gene field spells another transcript
            "missense|G|TXD|protein_coding|+|5K5R|1A>T",                                                     # no '>': Mutation::new fails, the group stays
            "missense|G|TXE|protein_coding|+|-5K>5R|1A>T", "start_lost|G|TXF", "missense|G|TXG|lincRNA|+|5K>5R|1A>T",
            "stop_gained|G|TXA|NMD|+|300Q>300*|1A>T", "missense|G|TXA|protein_coding|+|0K>0R|1A>T"]               # position 0 wraps to 65535
    text = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\n" + "".join(
        f"1\t{i}\t.\tA\tC\t.\t.\tBCSQ={m}|||||\tGT:BCSQ\t0|1:{1 + (i % 3)}\t1|1:3\n" if m.count("|") < 6 and not m.startswith("start_lost") else
        f"1\t{i}\t.\tA\tC\t.\t.\tBCSQ={m},missense|G|TZ{i}|protein_coding|+|1M>1V|1A>T\tGT:BCSQ\t0|1:{1 + (i % 3)}\t1|1:3\n" for i, m in enumerate(rows))
    compare_groups(text)


def test_grouping_random(built):
    for seed in range(6):
        compare_groups(random_vcf(100 + seed, 60, 9, n_tx=12 if seed % 2 else 50), n_threads=1 + seed % 4)


def test_grouping_large_table_uses_the_worker_threads(built):
    """More than 8 192 consequences: the per-file table (parse + substring matches) is built on several threads."""
    text = random_vcf(77, 1200, 5, n_tx=400)
    from frontend_util import oracle_index
    assert oracle_index(text)[3][-1] > 3 * 4096
    compare_groups(text, n_threads=7)


def test_grouping_abort_on_two_mutations_one_position(built):
    from vcf2prot_amd import _native as N
    a = "missense|G|TX|protein_coding|+|12C>12I|1A>T"
    b = "missense|G|TX|protein_coding|+|12C>12W|1A>T"
    text = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\n" + "".join(
        f"1\t{i}\t.\tA\tC\t.\t.\tBCSQ={m}\tGT:BCSQ\t0|0:0\t0|1:1\n" for i, m in enumerate([a, b]))
    with pytest.raises(N.V2PError) as e:
        compare_groups(text)
    assert e.value.code == -27 and e.value.index == 2          # haplotype 1 of sample B
