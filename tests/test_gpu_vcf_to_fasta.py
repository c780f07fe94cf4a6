"""BASELINE config 0 without the Rust host: VCF text + reference FASTA through the whole stack -- record index, GPU bitmask
decode, grouping, step 4a, step 4b, step 5 in the image builder, step 6 + FASTA emit on the GPU -- must give the files the
reference binary wrote (`-g st` == `-g mt`), compared as sets of records per proband.

flags=0: the 0.1.2 binary runs the INSPECT_INS_GEN / PANIC_INSPECT_ERR checks only when they are exported and the golden
generator unsets them -- the NO_TEST=1 behaviour of the current source (cli.rs:275-335).  With the checks on, the current
source aborts on the C1 example (an insertion's payload counts as overlapping the next mutation,
transcript_instructions.rs:99); tests/test_step4a.py covers the checks against the restatement."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")


def records(text: bytes):
    lines = text.decode().split("\n")
    assert lines[-1] == ""
    lines = lines[:-1]
    assert len(lines) % 2 == 0
    out = []
    for i in range(0, len(lines), 2):
        assert lines[i].startswith(">")
        out.append([lines[i][1:], lines[i + 1]])
    return sorted(out)


def cohort_examples():
    return sorted(f[:-5] for f in os.listdir(GOLDEN) if f.endswith(".json") and os.path.exists(os.path.join(GOLDEN, f[:-5] + ".vcf")))


@pytest.mark.parametrize("stem", cohort_examples())
def test_cohort_example_vcf_gives_the_reference_fasta(built, gpu_ctx, stem):
    from vcf2prot_amd.pipeline import vcf_to_fasta
    want = json.load(open(os.path.join(GOLDEN, stem + ".json")))
    vcf = open(os.path.join(GOLDEN, stem + ".vcf"), "rb").read()
    ref = open(os.path.join(GOLDEN, stem + "_reference.fasta")).read()
    got = vcf_to_fasta(gpu_ctx, vcf, ref, flags=0)
    assert sorted(got) == sorted(want["fasta"])
    n = 0
    for sample, recs in want["fasta"].items():
        assert records(got[sample]) == sorted(recs), (stem, sample)
        n += len(recs)
    assert n >= 100


def test_decode_golden_vcfs_give_the_reference_fasta(built, gpu_ctx):
    from vcf2prot_amd import _native as N
    from vcf2prot_amd.pipeline import vcf_to_fasta
    cases = json.load(open(os.path.join(GOLDEN, "decode_cases.json")))["cases"]
    n_abort = 0
    for c in cases:
        if c["panics"]:
            with pytest.raises(N.V2PError):
                vcf_to_fasta(gpu_ctx, c["vcf"].encode(), c["reference_fasta"], flags=0)
            n_abort += 1
            continue
        got = vcf_to_fasta(gpu_ctx, c["vcf"].encode(), c["reference_fasta"], flags=0)
        for sample in c["samples"]:
            assert records(got[sample]) == (c["fasta"][sample] or []), (c["name"], sample)
    assert n_abort >= 9


@pytest.mark.parametrize("stem", cohort_examples())
def test_write_all_proteins(built, gpu_ctx, stem):
    """-a / --write_all_proteins (personalized_genome.rs:118-204): every transcript of the reference per haplotype."""
    from vcf2prot_amd.pipeline import vcf_to_fasta
    want = json.load(open(os.path.join(GOLDEN, stem + ".json")))["fasta_write_all"]
    vcf = open(os.path.join(GOLDEN, stem + ".vcf"), "rb").read()
    ref = open(os.path.join(GOLDEN, stem + "_reference.fasta")).read()
    got = vcf_to_fasta(gpu_ctx, vcf, ref, flags=0, write_all=True)
    for sample, recs in want.items():
        assert records(got[sample]) == sorted(recs), (stem, sample)


def test_random_vcfs_answered_by_the_reference_binary(built, gpu_ctx):
    """24 random multi-sample VCFs (every consequence kind, several consequences per record, multi-word masks) that the
    reference binary gets through (oracle/make_random_vcf_golden.py): same records per proband, by length and digest."""
    import hashlib
    import random
    from vcf2prot_amd.pipeline import vcf_to_fasta
    cases = json.load(open(os.path.join(GOLDEN, "random_vcfs.json")))["cases"]
    aa = "ACDEFGHIKLMNPQRSTVWY"
    n = 0
    for c in cases:
        rng = random.Random(c["reference_seed"])
        ref = "".join(f">ENST{i:011d}\n{'M' + ''.join(rng.choice(aa) for _ in range(699))}\n" for i in range(20))
        got = vcf_to_fasta(gpu_ctx, c["vcf"].encode(), ref, flags=0)
        for sample in c["samples"]:
            mine = [[h, len(q), hashlib.sha256(q.encode()).hexdigest()[:16]] for h, q in records(got[sample])]
            assert mine == c["fasta"][sample], (c["name"], sample)
            n += len(mine)
    assert n > 1000
