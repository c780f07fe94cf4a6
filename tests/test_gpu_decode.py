"""BCSQ bitmask decode on the GPU (include/v2p_frontend.h part 2, decode_kernels.hip) against the restatement
(oracle/frontend_oracle.py), bit-exact, through the C ABI.  SURVEY section 8f rank 4."""
import json
import os

import numpy as np
import pytest

from frontend_util import F, oracle_lists, random_vcf

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def decode_cases():
    with open(os.path.join(HERE, "golden", "decode_cases.json")) as f:
        return json.load(f)["cases"]


def gpu_lists(gpu_ctx, text):
    from vcf2prot_amd.frontend import VcfIndex, decode_bitmasks
    idx = VcfIndex(text.encode())
    return idx, decode_bitmasks(gpu_ctx, idx)


def assert_same_lists(gpu_ctx, text):
    _, _, _, _, want = oracle_lists(text)
    idx, got = gpu_lists(gpu_ctx, text)
    assert got.n_haplotypes == len(want)
    assert got.hap_begin.tolist() == np.concatenate([[0], np.cumsum([len(x) for x in want])]).tolist()
    for h, w in enumerate(want):
        assert got.of(h).tolist() == w, f"haplotype list {h}"
    return idx, got


PANIC_CODE = (("An invalid bit mask", -20), ("unwrap on parse", -21), ("index out of bounds", -22))


def test_golden_vcfs_decode_like_the_reference(built, gpu_ctx, decode_cases):
    from vcf2prot_amd import _native as N
    n_abort = 0
    for c in decode_cases:
        try:
            oracle_lists(c["vcf"])
        except F.ReferencePanic as p:
            code = [k for m, k in PANIC_CODE if m in str(p)][0]
            with pytest.raises(N.V2PError) as e:
                gpu_lists(gpu_ctx, c["vcf"])
            assert e.value.code == code, c["name"]
            assert c["panics"]
            n_abort += 1
            continue
        assert_same_lists(gpu_ctx, c["vcf"])
    assert n_abort >= 8


def test_vcf_to_fasta_records_of_the_reference_binary(built, gpu_ctx, decode_cases):
    """index -> GPU decode -> grouping; every consequence of these VCFs is a missense, so the FASTA the reference
    binary wrote follows by substitution."""
    from vcf2prot_amd.frontend import group_per_transcript
    n = 0
    for c in decode_cases:
        if c["panics"]:
            continue
        ref = {}
        lines = c["reference_fasta"].split("\n")
        for i in range(0, len(lines) - 1, 2):
            ref[lines[i][1:]] = lines[i + 1]
        idx, lists = gpu_lists(gpu_ctx, c["vcf"])
        g = group_per_transcript(idx, lists)
        for s, name in enumerate(idx.sample_names()):
            recs = []
            for h in (0, 1):
                for tx, members in g.of(2 * s + h):
                    seq = list(ref[tx])
                    for i in members:
                        m = F.mutation_new(idx.consequence(i))
                        seq[m.mut_aa_position] = m.mut_aa
                    recs.append([f"{tx}_{h + 1}", "".join(seq)])
            assert sorted(recs) == (c["fasta"][name] or []), (c["name"], name)
            n += len(recs)
    assert n > 700


@pytest.mark.parametrize("seed,n_records,n_samples,p_zero", [(1, 300, 70, 0.5), (2, 40, 700, 0.3), (3, 700, 3, 0.1), (4, 257, 33, 0.9),
                                                              (5, 5, 2500, 0.5), (6, 1, 1, 0.0), (7, 513, 65, 0.0)])
def test_random_vcfs(built, gpu_ctx, seed, n_records, n_samples, p_zero):
    assert_same_lists(gpu_ctx, random_vcf(seed, n_records, n_samples, p_zero=p_zero, unique_positions=False))


def test_more_haplotypes_than_one_cursor_range_and_dense_records(built, gpu_ctx):
    """count / emit keep 6 144 haplotype cursors in LDS per workgroup: 3 300 samples take two ranges; with no empty column a record
    has more carriers than the 256 an emit step fetches ahead."""
    assert_same_lists(gpu_ctx, random_vcf(11, 70, 3300, p_zero=0.7, unique_positions=False))
    assert_same_lists(gpu_ctx, random_vcf(12, 130, 900, p_zero=0.0, unique_positions=False))


def test_record_block_with_more_than_65536_consequences(built, gpu_ctx):
    """The staged emit kernel keeps a block's ids as 16-bit offsets from the block's first consequence; a block whose records list
    more than 65 535 consequences between them goes through the direct kernel."""
    from frontend_util import HEADER
    rng = np.random.default_rng(5)
    n_rec, n_smp, n_csq = 70, 6, 1100
    rows = [HEADER, "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"S{i}" for i in range(n_smp)) + "\n"]
    for r in range(n_rec):
        csq = ",".join(f"missense|G{j}|ENST{(r * 7 + j) % 90:011d}|protein_coding|+|{1 + j % 400}A>{1 + j % 400}C|{r}A>T" for j in range(n_csq))
        cols = [f"0|1:{int(rng.integers(0, 64)) if rng.random() < 0.7 else 0}" for _ in range(n_smp)]
        rows.append(f"1\t{100 + r}\t.\tA\tT\t.\tPASS\tBCSQ={csq}\tGT:BCSQ\t" + "\t".join(cols) + "\n")
    idx, _ = assert_same_lists(gpu_ctx, "".join(rows))
    assert int(idx.csq_begin[64]) - int(idx.csq_begin[0]) > 0xFFFF


def test_64_bit_cursors(built, gpu_ctx):
    """Calls with 2^32 ids or more switch emit to 64-bit cursors; V2P_DECODE_CURSOR64 forces that kernel for a small call."""
    import os
    os.environ["V2P_DECODE_CURSOR64"] = "1"
    try:
        assert_same_lists(gpu_ctx, random_vcf(13, 200, 300, p_zero=0.4, unique_positions=False))
        assert_same_lists(gpu_ctx, random_vcf(14, 66, 3300, p_zero=0.8, unique_positions=False))
    finally:
        del os.environ["V2P_DECODE_CURSOR64"]


def test_columns_longer_than_a_tile_and_rows_at_every_alignment(built, gpu_ctx):
    """Sample columns with kilobytes of FORMAT text before the mask, so that columns straddle the 4 KiB parse tiles."""
    head = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tA\tB\tC\tD\tE\n"
    rows = []
    for r in range(40):
        pad = ["x" * ((r * 577 + k * 1201) % 3900) for k in range(5)]
        cols = [f"0|1:{pad[k]}:{(r + k) % 4}" for k in range(5)]
        cols[r % 5] = "0|1:" + "y" * ((r * 131) % 4000)               # no mask after the last ':' -> nothing
        rows.append(f"1\t{r}\t.{'z' * (r % 17)}\tA\tC\t.\t.\tBCSQ=missense|G|T{r}|protein_coding|+|9A>9C|1A>C\tGT:X:BCSQ\t" + "\t".join(cols))
    assert_same_lists(gpu_ctx, head + "\n".join(rows) + "\n")


@pytest.mark.parametrize("bad,code", [("0|1:-7", -20), ("0|1:3,-1", -20), ("0|1:-0", -21), ("0|1:1,,1", -21), ("0|1:1,a", -21),
                                      ("0|1:5,4294967296", -21), ("0|1:64", -22), ("0|1:1,16", -22)])
def test_abort_is_located(built, gpu_ctx, bad, code):
    """v2p_last_error_index = record * n_samples + sample of the first offending column (README.md:156-157 style)."""
    from vcf2prot_amd import _native as N
    text = random_vcf(11, 90, 40, max_csq=3)
    lines = text.split("\n")
    first = next(i for i, ln in enumerate(lines) if ln and not ln.startswith("#"))
    for rec, smp in ((0, 0), (57, 23), (89, 39)):
        ln = lines[first + rec].split("\t")
        ln[9 + smp] = bad
        mod = lines[:first + rec] + ["\t".join(ln)] + lines[first + rec + 1:]
        with pytest.raises(N.V2PError) as e:
            gpu_lists(gpu_ctx, "\n".join(mod))
        assert e.value.code == code and e.value.index == rec * 40 + smp
        with pytest.raises(F.ReferencePanic):
            oracle_lists("\n".join(mod))


def test_wrong_number_of_columns_is_refused(built, gpu_ctx):
    from vcf2prot_amd import _native as N
    text = random_vcf(12, 30, 8, max_csq=2)
    lines = text.split("\n")
    first = next(i for i, ln in enumerate(lines) if ln and not ln.startswith("#"))
    for rec, edit in ((4, lambda c: c[:-1]), (17, lambda c: c + ["0|0:0"]), (29, lambda c: c[:10])):
        mod = list(lines)
        mod[first + rec] = "\t".join(edit(lines[first + rec].split("\t")))
        with pytest.raises(N.V2PError) as e:
            gpu_lists(gpu_ctx, "\n".join(mod))
        assert e.value.code == -23 and e.value.index // 8 == rec


def test_cohort_scale_counts(built, gpu_ctx):
    """20 000 records x 3 000 samples (360 MB of text): list lengths and contents against numpy."""
    from vcf2prot_amd.frontend import VcfIndex, decode_bitmasks
    R, S = 20000, 3000
    rng = np.random.default_rng(5)
    m = rng.integers(0, 4, size=(R, S), dtype=np.uint8)
    m[rng.random((R, S)) < 0.6] = 0
    cell = np.frombuffer(b"0|1:0\t", dtype=np.uint8)
    body = np.tile(cell, (R, S, 1))
    body[:, :, 4] = m + ord("0")
    body[:, -1, 5] = ord("\n")
    pre = [f"1\t{r}\t.\tA\tC\t.\t.\tBCSQ=missense|G|T{r % 977}|protein_coding|+|{1 + r // 977}A>{1 + r // 977}C|1A>C\tGT:BCSQ\t".encode() for r in range(R)]
    head = ("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"S{i}" for i in range(S)) + "\n").encode()
    text = head + b"".join(p + body[r].tobytes() for r, p in enumerate(pre))
    idx = VcfIndex(text)
    assert (idx.n_records, idx.n_samples, idx.n_consequences) == (R, S, R)
    got = decode_bitmasks(gpu_ctx, idx)
    for h in (0, 1):
        sel = ((m >> h) & 1).astype(bool)
        assert (np.diff(got.hap_begin.astype(np.int64))[h::2] == sel.sum(axis=0)).all()
    for s in (0, 1, 1499, 2999):
        for h in (0, 1):
            assert got.of(2 * s + h).tolist() == np.nonzero((m[:, s] >> h) & 1)[0].tolist()
    assert int(got.hap_begin[-1]) == int(((m & 1) != 0).sum() + ((m & 2) != 0).sum())


def test_many_multi_word_masks_force_the_capacity_retry(built, gpu_ctx):
    """Every column carries a four-word mask: the side list outgrows its first allocation and v2p_decode_run runs again
    with the exact size."""
    import random
    rng = random.Random(8)
    S, R, n = 64, 330, 50
    head = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(f"S{i}" for i in range(S)) + "\n"
    rows = []
    for r in range(R):
        csq = ",".join(f"missense|G|T{(r * n + j) % 997}|protein_coding|+|{1 + j}A>{1 + j}C|1A>C" for j in range(n))
        cols = []
        for s in range(S):
            w = [rng.randrange(1, 1 << 30) for _ in range(3)] + [rng.randrange(1, 1 << 10)]      # indices 45..49 in the last word
            cols.append("0|1:" + ",".join(map(str, w)))
        rows.append(f"1\t{r}\t.\tA\tC\t.\t.\tBCSQ={csq}\tGT:BCSQ\t" + "\t".join(cols))
    assert R * S * 5 > R * S // 4 + 65536
    assert_same_lists(gpu_ctx, head + "\n".join(rows) + "\n")
