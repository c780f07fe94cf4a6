"""The ONE-PASS device image builder (csrc/build_rows.hip; v2p_batch_build_on_device kernel 6 = wave image, 7 = dense): the image
in HBM must equal, byte for byte, the ROWS image the host builds from the same stream (csrc/rows_image.hpp: whole descriptors,
chunks cut afterwards on 1 KiB rows, head skip / row clip), and the executed arena must equal the oracle's tapes
(task.rs:38-50 per transcript, concatenated as haplotype_instruction.rs:94-133 does) -- stitchw_kernel's ROWS instance."""
import json
import os

import numpy as np
import pytest

from stream_util import random_stream

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_hap(c, coracle, h):
    hap = c.haplotype(h)
    t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
    return coracle.gir_execute_u8(t, c.ref_tape_u32(h).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8))


def _same_image(ctx, b, stream, proteome_len, mode):
    """device image == host rows image (chunk table dealt to the XCDs by the same host rule)"""
    from vcf2prot_amd.txstream import pack_rows
    want = pack_rows(stream, proteome_len, mode, 0)
    desc, chunks, hb = b.download_image()
    assert np.array_equal(hb, want.hap_out_begin)
    assert desc.size == want.desc.size, (desc.size, want.desc.size)
    assert np.array_equal(desc, want.desc), int(np.argmax(desc != want.desc))
    wc = np.ascontiguousarray(want.chunks)
    ctx._lib.v2p_order_chunks_for_xcds(wc.ctypes.data, wc.shape[0], want.desc.ctypes.data, want.desc.size, proteome_len)
    assert chunks.shape == wc.shape
    key = lambda t: t[np.lexsort((t[:, 0], t[:, 1] & np.uint64((1 << 48) - 1)))]
    assert np.array_equal(key(chunks), key(wc))                    # the same chunks ...
    assert np.array_equal(chunks, wc)                              # ... in the same launch order
    return want


@pytest.mark.parametrize("preset,h0,n,kernel", [
    ("C1", 0, 8, 6), ("C2", 5, 3, 6), ("C3", 100, 40, 6), ("C4", 7, 3, 6), ("C2", 0, 40, 6), ("C3", 0, 400, 6),
    ("C1", 0, 8, 7), ("C5", 50, 300, 7), ("C5", 11, 100, 7), ("C3", 100, 40, 7)])
def test_device_rows_image_equals_the_host_rows_image(built, gpu_ctx, coracle, preset, h0, n, kernel):
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset(preset)
    prot = c.proteome()
    gpu_ctx.upload_proteome(prot)
    stream = c.txstream(h0, h0 + n, n_threads=3)
    b = gpu_ctx.batch()
    ms = b.build_on_device(stream, 0, kernel)
    assert ms > 0
    _same_image(gpu_ctx, b, stream, prot.size, 1 if kernel == 6 else 2)
    b.execute()
    b.sync()
    for i in range(0, n, max(1, n // 40)):
        assert np.array_equal(b.download_hap(i), oracle_hap(c, coracle, h0 + i)), (preset, h0 + i)
    b.close()
    stream.close()


@pytest.mark.parametrize("seed,shape,kernel", [(1, "snv", 7), (2, "snv", 7), (3, "snv", 6), (5, "mix", 6), (6, "mix", 7), (7, "mix", 6),
                                               (10, "long", 6), (11, "long", 7), (12, "long", 6), (13, "mix", 7), (14, "long", 6)])
def test_random_streams_through_the_rows_builder(built, gpu_ctx, seed, shape, kernel):
    from vcf2prot_amd._native import V2PError
    from vcf2prot_amd.txstream import pack_rows, RowsError
    rng = np.random.default_rng(seed)
    proteome, stream, want = random_stream(rng, n_haps=40, n_ref_tx=25, shape=shape, window=4096)
    gpu_ctx.upload_proteome(proteome)
    b = gpu_ctx.batch()
    try:
        b.build_on_device(stream, 0, kernel)
    except V2PError as e:
        assert kernel == 6 and e.code == -9                         # more than 64 descriptors in a row: the host refuses it too
        with pytest.raises(RowsError):
            pack_rows(stream, proteome.size, 1, 0)
        b.build_on_device(stream, 0, 7)
        kernel = 7
    _same_image(gpu_ctx, b, stream, proteome.size, 1 if kernel == 6 else 2)
    b.execute()
    b.sync()
    for h, w in enumerate(want):
        got = b.download_hap(h)
        assert got.size == w.size and np.array_equal(got, w), (seed, shape, kernel, h, int(np.argmax(got != w)) if got.size == w.size else -1)
    b.close()


@pytest.mark.parametrize("kernel", [6, 7])
@pytest.mark.parametrize("fasta", [False, True])
def test_reference_task_dumps_through_the_rows_builder(gpu_ctx, golden, kernel, fasta):
    """The 36 transcript GIRs harvested from the reference binary (its own Vec<Task> dumps), several per haplotype; with FASTA emit
    the arena is the file text of personalized_genome.rs:90-113."""
    from test_gpu_device_build_fasta import _stream_of_cases
    from vcf2prot_amd._native import V2PError
    cases = golden["cases"]
    refs, off = {}, 0
    for c in cases:
        if c["ref"] not in refs:
            refs[c["ref"]] = off
            off += len(c["ref"])
    proteome = np.frombuffer("".join(refs).encode(), dtype=np.uint8)
    headers = "\n" + "".join(f">{c['name']}_1\n" for c in cases)
    hdr_off, o = [], 1
    for c in cases:
        hdr_off.append(o)
        o += len(c["name"]) + 4
    gpu_ctx.upload_reference(proteome, np.frombuffer(headers.encode(), dtype=np.uint8))
    per_hap = 7
    stream = _stream_of_cases(cases, refs, hdr_off, fasta, per_hap)
    b = gpu_ctx.batch()
    try:
        b.build_on_device(stream, 0, kernel)
    except V2PError as e:
        assert kernel == 6 and e.code == -9
        b.build_on_device(stream, 0, 7)
    b.execute()
    b.sync()
    for h in range(0, (len(cases) + per_hap - 1) // per_hap):
        mine = cases[h * per_hap:(h + 1) * per_hap]
        text = b.download_hap(h).tobytes().decode()
        want = "".join(f">{c['name']}_1\n{c['expected']}\n" for c in mine) if fasta else "".join(c["expected"] for c in mine)
        assert text == want, (kernel, fasta, h)
    b.close()


def test_rows_builder_reports_what_the_reference_would_panic_on(built, gpu_ctx):
    """update_task (haplotype_instruction.rs:154) and Task::execute's slices (task.rs:43,47): the first offending task, by index."""
    from stream_util import Stream
    from vcf2prot_amd._native import V2PError
    prot = np.frombuffer(b"MEDLGENTMVLSTLRSLNNFISQRVEGGSGLEELERGGAKLMNPQRSTVWYACDEFGHIK", dtype=np.uint8)
    gpu_ctx.upload_proteome(prot)

    def stream(code, sp, ln, sr, res_len=60, alt=b"AC"):
        return Stream([0, 1], [0], [60], [res_len], [0, len(code)], [0, len(alt)], code, sp, ln, sr, np.frombuffer(alt, dtype=np.uint8))
    for kernel in (6, 7):
        for (code, sp, ln, sr), want_code, row in [
                (([0, 2, 0], [0, 0, 11], [10, 1, 49], [0, 10, 11]), -3, 1),      # bad exe code
                (([0, 1, 0], [0, 0, 11], [10, 1, 50], [0, 10, 11]), -4, 2),      # result out of bounds
                (([0, 1, 0], [0, 0, 30], [10, 1, 40], [0, 10, 11]), -5, 2),      # source out of bounds
                (([0, 1, 0], [0, 0, 11], [10, 1, 49], [0, 9, 11]), -6, 1)]:      # overlapping result ranges
            b = gpu_ctx.batch()
            with pytest.raises(V2PError) as ei:
                b.build_on_device(stream(code, sp, ln, sr), 0, kernel)
            assert ei.value.code == want_code and ei.value.index == row, (kernel, ei.value.code, ei.value.index)
            b.build_on_device(stream([0, 1, 0], [0, 0, 11], [10, 1, 49], [0, 10, 11]), 0, kernel)     # the batch is reusable
            b.scribble(); b.execute(); b.sync()
            assert b.download_hap(0).tobytes() == bytes(prot[:10]) + b"A" + bytes(prot[11:])
            b.close()


def test_rows_builder_falls_back_for_tiles_that_do_not_fit_the_stage(built, gpu_ctx):
    """One transcript with thousands of tasks (more descriptors than a wave stages in LDS) and one with a result of several hundred
    KiB (more rows than it stages): the build is redone with the two-phase kernel, same image."""
    from stream_util import Stream
    rng = np.random.default_rng(5)
    AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    L = 300000
    prot = AA[rng.integers(0, 20, size=L)]
    gpu_ctx.upload_proteome(prot)
    # transcript 0: 3000 substitutions 60 residues apart; transcript 1: one copy of everything
    code, sp, ln, sr, alt = [], [], [], [], []
    pos = 0
    for k in range(3000):
        code += [0, 1]; sp += [pos, k]; ln += [59, 1]; sr += [pos, pos + 59]
        alt.append(int(AA[k % 20])); pos += 60
    code.append(0); sp.append(pos); ln.append(L - pos); sr.append(pos)
    n0 = len(code)
    code.append(0); sp.append(0); ln.append(L); sr.append(0)
    s = Stream([0, 2], [0, 0], [L, L], [L, L], [0, n0, n0 + 1], [0, len(alt), len(alt)], code, sp, ln, sr, np.array(alt, dtype=np.uint8))
    want = prot.copy()
    for k in range(3000):
        want[60 * k + 59] = AA[k % 20]
    for kernel in (6, 7):
        b = gpu_ctx.batch()
        b.build_on_device(s, 0, kernel)
        _same_image(gpu_ctx, b, s, prot.size, 1 if kernel == 6 else 2)
        b.scribble(); b.execute(); b.sync()
        got = b.download_hap(0)
        assert np.array_equal(got[:L], want) and np.array_equal(got[L:], prot)
        b.close()


def test_cutter_falls_back_when_a_segment_has_more_chunks_than_its_padded_slots(built, gpu_ctx, coracle):
    """100-residue transcripts with one substitution each: a wave chunk's 64 descriptors are 6 KiB, so a 640-row segment holds about
    107 chunks -- more than the 96 slots the cutter's single pass has per segment: the builder then runs the emitting pass after the
    scan instead of the copy.  Same image as the host's, same tapes as the oracle's."""
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset("C3", mean_len=100.0, len_model=0, fixed_len=100, n_transcripts=4000, alts_fixed=1, altered_per_hap=2000, n_samples=8,
                      mix=[1.0] + [0.0] * 5)
    prot = c.proteome()
    gpu_ctx.upload_proteome(prot)
    n = c.n_haplotypes
    stream = c.txstream(0, n, n_threads=3)
    b = gpu_ctx.batch()
    b.build_on_device(stream, 0, 6)
    want = _same_image(gpu_ctx, b, stream, prot.size, 1)
    per_seg = np.bincount((want.chunks[:, 1] & np.uint64((1 << 48) - 1)) // np.uint64(640 * 1024))
    assert per_seg.max() > 96, per_seg.max()                        # (the case this test is about)
    b.execute()
    b.sync()
    for i in range(n):
        assert np.array_equal(b.download_hap(i), oracle_hap(c, coracle, i)), i
    b.close()
    stream.close()
    c.close()


def test_a_haplotype_whose_arena_range_crosses_4_gib(built, gpu_ctx, coracle):
    """Arena offsets above 2^32 INSIDE one haplotype: [3 small transcripts] [4 300 transcripts of 1 MiB each = 4.5 GB] [3 small ones],
    one substitution per transcript.  The parse's positions are 32-bit offsets from a TILE's first byte, the cover map and the chunk
    records hold 48-bit arena offsets: the big haplotype's digest and bytes either side of the 4 GiB line must be the oracle's
    (task.rs:38-50 over the haplotype's own tapes, haplotype_instruction.rs:94-133).  A 1 MiB copy is 513 descriptors, so its tile
    overflows the one-pass stage and the build runs in its two-pass form (descriptor indices up to 2.2e6 in the cover words).
    (More than 2^25 TILES -- the one-pass cover word's tile field -- would take 2^25 transcripts of >= 360 tasks each, 157 GB of stream:
    not reachable here; from that many tiles on the builder takes the two-pass form, whose cover word has no tile field.)"""
    from stream_util import Stream
    rng = np.random.default_rng(17)
    AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    L, small, n_big = 1 << 20, 700, 4300
    prot = AA[rng.integers(0, 20, size=L + 64)]
    gpu_ctx.upload_proteome(prot)
    lens = [small] * 3 + [L] * n_big + [small] * 3
    n_tx = len(lens)
    pos = [int(rng.integers(1, ln - 1)) for ln in lens]
    sub = AA[rng.integers(0, 20, size=n_tx)]
    code = np.tile(np.array([0, 1, 0], dtype=np.uint8), n_tx)
    sp = np.array([v for t in range(n_tx) for v in (0, 1, pos[t] + 1)], dtype=np.uint32)
    ln = np.array([v for t in range(n_tx) for v in (pos[t], 1, lens[t] - pos[t] - 1)], dtype=np.uint32)
    sr = np.array([v for t in range(n_tx) for v in (0, pos[t], pos[t] + 1)], dtype=np.uint32)
    alt = np.repeat(sub, 2)                                                   # a missense payload is pushed twice (transcript_instructions.rs:659-660)
    s = Stream([0, 3, 3 + n_big, n_tx], [0] * n_tx, lens, lens, np.arange(0, 3 * n_tx + 1, 3), np.arange(0, 2 * n_tx + 1, 2), code, sp, ln, sr, alt)
    b = gpu_ctx.batch()
    b.build_on_device(s, 0, 6)
    _, chunks, hb = b.download_image()
    want_hb = np.concatenate([[0], np.cumsum([3 * small, n_big * L, 3 * small])]).astype(np.uint64)
    assert np.array_equal(hb, want_hb) and int(hb[1]) < (1 << 32) < int(hb[2])
    dst = chunks[:, 1] & np.uint64((1 << 48) - 1)
    assert int(dst.max()) > (1 << 32) and np.all(dst % np.uint64(1024) == 0)
    b.execute()
    b.sync()
    dig = b.digests()
    # the oracle on the big haplotype's own tapes: step 5 rebases every transcript's tasks onto the haplotype's ref / alt / result tapes
    t0, t1 = 3, 3 + n_big
    k = np.arange(t1 - t0, dtype=np.uint64)
    code_h = code[3 * t0:3 * t1]
    sp_h = sp[3 * t0:3 * t1].astype(np.uint64) + np.where(code_h == 0, np.repeat(k * np.uint64(L), 3), np.repeat(k * np.uint64(2), 3))
    sr_h = sr[3 * t0:3 * t1].astype(np.uint64) + np.repeat(k * np.uint64(L), 3)
    tasks = coracle.pack_tasks(code_h, sp_h, ln[3 * t0:3 * t1].astype(np.uint64), sr_h)
    want = coracle.gir_execute_u8(tasks, np.tile(prot[:L], n_big), alt[2 * t0:2 * t1], np.full(n_big * L, ord("."), dtype=np.uint8))
    assert int(dig[1]) == coracle.digest_u8(want)
    line = (1 << 32) - int(hb[1])                                             # the 4 GiB line inside the haplotype
    for a, n in ((0, 4096), (line - 70000, 140000), (n_big * L - 4096, 4096)):
        assert np.array_equal(b.download(int(hb[1]) + a, n), want[a:a + n]), a
    del want
    for h, (ta, tb) in ((0, (0, 3)), (2, (3 + n_big, n_tx))):
        w = np.concatenate([np.concatenate([prot[:pos[t]], sub[t:t + 1], prot[pos[t] + 1:lens[t]]]) for t in range(ta, tb)])
        assert np.array_equal(b.download_hap(h), w), h
        assert int(dig[h]) == coracle.digest_u8(w)
    b.close()


@pytest.mark.parametrize("kernel", [0, 6, 7])
def test_a_reference_beyond_4_gib(built, gpu_ctx, kernel):
    """Source offsets above 2^32: a resident proteome of 4 GiB + 2 MiB, transcripts just below, across and above the 4 GiB line, each with one
    substituted residue (task.rs:38-50).  The one-pass parse has a 32-bit instance for every cohort whose sources fit (SRC32); this one
    must take the 64-bit instance -- bases of 64 bits out of the per-transcript LDS record, sources of 40 bits in the descriptors -- and a
    tile image (31-bit sources) must not be picked by the rule, nor built by number."""
    from vcf2prot_amd._native import V2PError
    from stream_util import Stream
    rng = np.random.default_rng(23)
    AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    G4 = 1 << 32
    block = AA[rng.integers(0, 20, size=1 << 20)]
    prot = np.tile(block, (G4 >> 20) + 2)                                      # 4 GiB + 2 MiB
    # ... with distinct text where the transcripts sit, so that a wrapped or truncated source offset cannot read the same residues
    marks = [G4 - 5000, G4 - 300, G4 + 7, G4 + (1 << 20) + 12345]
    lens = [900, 777, 1200, 64]
    for m, ln in zip(marks, lens):
        prot[m:m + ln] = AA[rng.integers(0, 20, size=ln)]
    gpu_ctx.upload_proteome(prot)
    n_tx = len(marks)
    pos = [int(rng.integers(1, ln - 1)) for ln in lens]
    sub = AA[rng.integers(0, 20, size=n_tx)]
    code = np.tile(np.array([0, 1, 0], dtype=np.uint8), n_tx)
    sp = np.array([v for t in range(n_tx) for v in (0, 1, pos[t] + 1)], dtype=np.uint32)
    ln = np.array([v for t in range(n_tx) for v in (pos[t], 1, lens[t] - pos[t] - 1)], dtype=np.uint32)
    sr = np.array([v for t in range(n_tx) for v in (0, pos[t], pos[t] + 1)], dtype=np.uint32)
    alt = np.repeat(sub, 2)
    s = Stream([0, 2, n_tx], marks, lens, lens, np.arange(0, 3 * n_tx + 1, 3), np.arange(0, 2 * n_tx + 1, 2), code, sp, ln, sr, alt)
    want = [np.concatenate([np.concatenate([prot[marks[t]:marks[t] + pos[t]], sub[t:t + 1], prot[marks[t] + pos[t] + 1:marks[t] + lens[t]]]) for t in ts]) for ts in ((0, 1), (2, 3))]
    rs = gpu_ctx.upload_stream(s)
    b = gpu_ctx.batch()
    b.build_and_execute(rs, kernel, 0)
    b.sync()
    assert not b.image_form()["tiles"]
    for rep in range(2):
        for h in range(2):
            assert np.array_equal(b.download_hap(h), want[h]), (kernel, rep, h)
        b.scribble(); b.execute(); b.sync()
    b.reset()
    with pytest.raises(V2PError) as e:
        b.build_and_execute(rs, 9, 0)
    assert e.value.code == -9
    b.close(); rs.close()
    del prot
    gpu_ctx.upload_proteome(block[:1000])
