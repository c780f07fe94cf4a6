"""TILE images (csrc/dense_pieces.h, round 6): what the one call builds for DEEP Task vectors (runs of substitutions:
transcript_instructions.rs:508-629,654-663).  The one-pass parse writes PIECES -- <= 16 result bytes of one source, their offset inside
the tile's result, at most one substituted residue -- straight into the slots of its tile of transcripts, and the tile is the executor's
work item: no dense image first, no compaction, no row map, no cutter, no chunk table, nothing to re-write at a second execute.
Semantics are the reference's (task.rs:38-50 per Task, '.' for cells nothing covers: haplotype_instruction.rs:78): every arena is
compared with the oracle's tapes, with the dense rows image of the same stream (kernel 7) and with plain numpy expectations."""
import numpy as np
import pytest

from stream_util import random_stream, regular_stream
from test_gpu_oneshot import oracle_hap

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("preset,h0,n,kernel", [("C5", 50, 1500, 0), ("C5", 11, 300, 9), ("C3", 100, 300, 9), ("C1", 0, 8, 9), ("C2", 0, 40, 9)])
def test_tile_image_of_a_cohort(built, gpu_ctx, coracle, preset, h0, n, kernel):
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset(preset)
    gpu_ctx.upload_proteome(c.proteome())
    stream = c.txstream(h0, h0 + n, n_threads=4)
    rs = gpu_ctx.upload_stream(stream)
    stream.close()
    d = gpu_ctx.batch()
    d.build_and_execute(rs, 7, 0); d.sync()                 # the dense rows image of the same stream
    want_dig = d.digests()
    d.close()
    b = gpu_ctx.batch()
    b.build_and_execute(rs, kernel, 0)                      # kernel 0: the routing rule must pick a tile image for C5
    b.sync()
    info = b.oneshot_info()
    assert info["kernel"] == 9 and b.image_form()["tiles"], (info, b.image_form())
    cn = b.counts()
    assert cn["n_haps"] == n and cn["out_bytes"] == int(c.result_sizes(h0, h0 + n).sum()) and cn["n_desc"] > 0 and cn["n_chunks"] > 0
    assert np.array_equal(b.digests(), want_dig)
    for i in range(0, n, max(1, n // 25)):
        assert np.array_equal(b.download_hap(i), oracle_hap(c, coracle, h0 + i)), (preset, h0 + i)
    for _ in range(2):                                       # executed again: the same image, the same kernel -- and it must WRITE the arena
        b.scribble(0x5A); b.execute(); b.sync()
        assert np.array_equal(b.digests(), want_dig)
    # the two-call form, and a recycled batch
    b.reset()
    ms = b.build_from_stream(rs, kernel)
    assert ms > 0 and b.image_form()["tiles"]
    b.execute(); b.sync()
    assert np.array_equal(b.digests(), want_dig)
    from vcf2prot_amd._native import V2PError
    with pytest.raises(V2PError) as e:
        b.download_image()
    assert e.value.code == -10
    b.close(); rs.close()


def test_long_tailed_transcripts_are_not_a_tile_image(built, gpu_ctx, coracle):
    """C4's transcripts (log-normal lengths, some of several KiB): no K keeps a tile inside the executor's LDS image with six standard
    deviations to spare -- by number the form is refused, the rule never picks it (C4 is a wave image anyway)."""
    from vcf2prot_amd._native import V2PError
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset("C4")
    gpu_ctx.upload_proteome(c.proteome())
    stream = c.txstream(7, 19, n_threads=4)
    rs = gpu_ctx.upload_stream(stream)
    stream.close()
    b = gpu_ctx.batch()
    with pytest.raises(V2PError) as e:
        b.build_and_execute(rs, 9, 0)
    assert e.value.code == -9
    b.reset()
    b.build_and_execute(rs, 0, 0); b.sync()
    assert not b.image_form()["tiles"]
    assert np.array_equal(b.download_hap(3), oracle_hap(c, coracle, 10))
    b.close(); rs.close()


@pytest.mark.parametrize("seed,shape,fasta", [(1, "snv", False), (3, "snv", True), (5, "mix", False), (6, "mix", True), (10, "long", False), (12, "long", True),
                                              (31, "snv", False), (32, "mix", True)])
def test_random_streams_as_tile_images(built, gpu_ctx, seed, shape, fasta):
    """Irregular streams (empty haplotypes, transcripts without Tasks, zero-length Tasks, gaps, tails, payloads of 1 .. 4 000 bytes, fused
    substitutions at every distance from a 16-byte piece boundary), with and without FASTA emit.  A stream whose tiles cannot hold a
    transcript ('long': transcripts of several windows) is refused by number (kernel 9: V2P_ERR_UNSUPPORTED) and built dense by the rule."""
    from vcf2prot_amd._native import V2PError
    rng = np.random.default_rng(seed)
    if fasta:
        proteome, headers, stream, want = random_stream(rng, n_haps=400, n_ref_tx=25, shape=shape, window=4096, fasta=True)
        gpu_ctx.upload_reference(proteome, headers)
    else:
        proteome, stream, want = random_stream(rng, n_haps=400, n_ref_tx=25, shape=shape, window=4096)
        gpu_ctx.upload_proteome(proteome)
    rs = gpu_ctx.upload_stream(stream)
    b = gpu_ctx.batch()
    try:
        b.build_and_execute(rs, 9, 0)
        tiles = True
    except V2PError as e:
        assert e.code == -9 and shape == "long", (e, shape)
        b.reset()
        b.build_and_execute(rs, 0, 0)
        tiles = False
    b.sync()
    assert b.image_form()["tiles"] == tiles
    for rep in range(2):
        for h, w in enumerate(want):
            got = b.download_hap(h)
            assert got.size == w.size and np.array_equal(got, w), (seed, shape, fasta, rep, h)
        b.scribble(); b.execute(); b.sync()
    b.close(); rs.close()
    if fasta:
        gpu_ctx.upload_proteome(proteome)


def test_what_the_reference_panics_on_is_reported_not_executed(built, gpu_ctx):
    """update_task / Task::execute (haplotype_instruction.rs:140-158, task.rs:43,47): a bad Task is reported with its row and the arena
    is not written -- the executor reads the build's status word."""
    from stream_util import Stream
    from vcf2prot_amd._native import V2PError
    rng = np.random.default_rng(8)
    proteome, stream, want = random_stream(rng, n_haps=60, n_ref_tx=12, shape="snv", window=4096)
    gpu_ctx.upload_proteome(proteome)
    k = stream.keep
    n_tasks = int(stream.struct.n_tasks)
    for row, field, value, code in ((n_tasks // 2, 6, 5, -3), (n_tasks // 3, 8, 1 << 30, None), (7, 7, 1 << 29, -5)):
        arrays = [x.copy() for x in k[:6]] + [x[:-64].copy() for x in k[6:11]]
        arrays[field][row] = value
        bad = Stream(*arrays)
        rs = gpu_ctx.upload_stream(bad)
        b = gpu_ctx.batch()
        with pytest.raises(V2PError) as e:
            b.build_and_execute(rs, 9, 0)
            b.sync()
        assert e.value.code in ((code,) if code else (-4, -5)) and e.value.index == row, (e.value.code, e.value.index, row)
        b.close(); rs.close()


def test_tiles_at_every_alignment(built, gpu_ctx):
    """A tile's result range starts anywhere: neighbouring tiles share a 16-byte block (and a cache line) and each writes its own bytes of it.
    Transcripts of every length from 1 to 70 residues, so that tile boundaries fall on every offset of a block."""
    from stream_util import Stream
    rng = np.random.default_rng(5)
    AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    proteome = AA[rng.integers(0, 20, size=5000)]
    hap_tx_begin, tx_off, tx_ref_len, tx_res_len, tb, ab, code, sp, ln, sr, alt, want = [0], [], [], [], [0], [0], [], [], [], [], [], []
    for h in range(300):
        res = []
        for _ in range(int(rng.integers(1, 40))):
            L = int(rng.integers(1, 71)); o = int(rng.integers(0, 5000 - L))
            p = int(rng.integers(0, L))
            byte = int(AA[rng.integers(0, 20)])
            out = proteome[o:o + L].copy(); out[p] = byte
            if p:
                code.append(0); sp.append(0); ln.append(p); sr.append(0)
            code.append(1); sp.append(0); ln.append(1); sr.append(p)
            if L - p - 1:
                code.append(0); sp.append(p + 1); ln.append(L - p - 1); sr.append(p + 1)
            alt.append(byte)
            tx_off.append(o); tx_ref_len.append(L); tx_res_len.append(L); tb.append(len(code)); ab.append(len(alt))
            res.append(out)
        hap_tx_begin.append(len(tx_off))
        want.append(np.concatenate(res))
    stream = Stream(hap_tx_begin, tx_off, tx_ref_len, tx_res_len, tb, ab, code, sp, ln, sr, alt)
    gpu_ctx.upload_proteome(proteome)
    rs = gpu_ctx.upload_stream(stream)
    b = gpu_ctx.batch()
    b.build_and_execute(rs, 0, 0); b.sync()
    assert b.image_form()["tiles"]
    for rep in range(2):
        for h, w in enumerate(want):
            assert np.array_equal(b.download_hap(h), w), (rep, h)
        b.scribble(0x11); b.execute(); b.sync()
    b.close(); rs.close()


def test_a_large_deep_stream(built, gpu_ctx):
    """Three million Tasks of delins runs (nothing fuses, payloads of 8 bytes in the alt stream), 170 MB of result: tiles of the size the
    statistics pick, every haplotype against numpy."""
    proteome, stream, want = regular_stream(n_haps=176, tx_per_hap=500, dense_every=1, seed=4, run=14, sub=8)
    gpu_ctx.upload_proteome(proteome)
    rs = gpu_ctx.upload_stream(stream)
    b = gpu_ctx.batch()
    b.build_and_execute(rs, 0, 0); b.sync()
    assert b.image_form()["tiles"] and b.oneshot_info()["kernel"] == 9
    for h in range(0, 176, 5):
        assert np.array_equal(b.download_hap(h), want(h)), h
    b.scribble(); b.execute(); b.sync()
    for h in range(1, 176, 7):
        assert np.array_equal(b.download_hap(h), want(h)), h
    b.close(); rs.close()


@pytest.mark.parametrize("seed,shape,fasta,slices", [(2, "snv", False, 3), (7, "mix", True, 5), (9, "mix", False, 32)])
def test_tile_image_built_and_executed_in_slices(built, dev_ctx, seed, shape, fasta, slices):
    """Development library (A/B, profiles/r06_tile_slices.txt): slice j's tiles executed on a second stream while slice j + 1 is parsed --
    the same arena as the product's one parse + one execute; the product library refuses n_slices > 1."""
    rng = np.random.default_rng(seed)
    if fasta:
        proteome, headers, stream, want = random_stream(rng, n_haps=300, n_ref_tx=20, shape=shape, window=4096, fasta=True)
        dev_ctx.upload_reference(proteome, headers)
    else:
        proteome, stream, want = random_stream(rng, n_haps=300, n_ref_tx=20, shape=shape, window=4096)
        dev_ctx.upload_proteome(proteome)
    rs = dev_ctx.upload_stream(stream)
    b = dev_ctx.batch()
    b.build_and_execute(rs, 9, slices)
    b.sync()
    assert b.image_form()["tiles"]
    for rep in range(2):
        for h, w in enumerate(want):
            got = b.download_hap(h)
            assert got.size == w.size and np.array_equal(got, w), (seed, shape, fasta, slices, rep, h)
        b.scribble(); b.execute(); b.sync()
    b.close(); rs.close()
    if fasta:
        dev_ctx.upload_proteome(proteome)
