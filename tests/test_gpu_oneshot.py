"""Task vectors -> result bytes in ONE call (round 5): a transcript stream made resident on the device (v2p_stream_upload), then
v2p_batch_build_and_execute -- the rows image built slice by slice on a second HIP stream while the slice before it is stitched.
What the reference does once per haplotype (get_g_rep(..).execute(engine): haplotype_instruction.rs:75-137 -> gir.rs:197-241,
personalized_genome.rs:64-65).  The image must be the one-piece builder's -- descriptors byte for byte, the same chunk records, the same
haplotype offsets -- and the arena the oracle's tapes, for every slicing."""
import numpy as np
import pytest

from stream_util import random_stream

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["product", "development"])
def one(request, gpu_ctx, dev_ctx):
    """(context, development?) -- every test of this module runs on the PRODUCT library (one slice, no switches) and on the development
    library (libv2p_bench.so: the same engine with slices, the A/B switches and PATCH images compiled in; csrc/bench/v2p_bench.h)."""
    return (gpu_ctx, False) if request.param == "product" else (dev_ctx, True)


def oracle_hap(c, coracle, h):
    hap = c.haplotype(h)
    t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
    return coracle.gir_execute_u8(t, c.ref_tape_u32(h).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8))


def _sorted_chunks(chunks):
    return chunks[np.lexsort((chunks[:, 0], chunks[:, 1] & np.uint64((1 << 48) - 1)))]


def _one_piece(ctx, rs, kernel):
    m = ctx.batch()
    ms = m.build_from_stream(rs, kernel)
    assert ms > 0
    img = None if m.image_form()["tiles"] else m.download_image()        # (deep Task vectors under the routing rule: a tile image -- pieces, no descriptors)
    m.execute()
    m.sync()
    dig = m.digests()
    m.close()
    return img, dig


@pytest.mark.parametrize("preset,h0,n,kernel,slices", [
    ("C1", 0, 8, 0, 0), ("C2", 3, 60, 6, 3), ("C3", 100, 400, 0, 5), ("C3", 0, 900, 6, 2), ("C4", 7, 12, 0, 4),
    ("C5", 50, 1500, 0, 3), ("C5", 11, 900, 7, 7), ("C3", 100, 300, 7, 2), ("C2", 0, 40, 0, 1)])
def test_sliced_build_and_execute_equals_the_one_piece_builder_and_the_oracle(built, one, coracle, preset, h0, n, kernel, slices):
    gpu_ctx, dev = one
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset(preset)
    gpu_ctx.upload_proteome(c.proteome())
    stream = c.txstream(h0, h0 + n, n_threads=4)
    rs = gpu_ctx.upload_stream(stream)
    stream.close()                                          # (resident: the host copy may go)
    sizes = c.result_sizes(h0, h0 + n)
    assert rs.counts()["out_bytes"] == int(sizes.sum()) and rs.counts()["n_haps"] == n
    img1, dig1 = _one_piece(gpu_ctx, rs, kernel)
    b = gpu_ctx.batch()
    b.build_and_execute(rs, kernel, slices if dev else min(slices, 1))
    b.sync()
    info = b.oneshot_info()
    assert info["kernel"] in (6, 7, 9) and info["total_ms"] > 0
    tiles = b.image_form()["tiles"]
    assert tiles == (info["kernel"] == 9) == (img1 is None) and (not tiles or kernel == 0)
    if slices > 1 and not tiles and dev:
        assert 1 <= info["n_slices"] <= slices
    _, _, hb = b.download_image() if not tiles else (None, None, np.array([b.hap_range(i)[0] for i in range(n)] + [int(sizes.sum())], dtype=np.uint64))
    assert np.array_equal(np.diff(hb.astype(np.int64)), sizes.astype(np.int64))
    if not tiles:
        desc1, chunks1, hb1 = img1
        desc, chunks, hb = b.download_image()
        assert np.array_equal(hb, hb1)
        assert desc.size == desc1.size and np.array_equal(desc, desc1)
        assert chunks.shape == chunks1.shape and np.array_equal(_sorted_chunks(chunks), _sorted_chunks(chunks1))
    assert np.array_equal(b.digests(), dig1)
    for i in range(0, n, max(1, n // 30)):
        assert np.array_equal(b.download_hap(i), oracle_hap(c, coracle, h0 + i)), (preset, h0 + i)
    # the batch is an ordinary finalized batch: executing it again leaves the same arena (scribbled first: the re-execute must WRITE it)
    b.scribble()
    b.execute()
    b.sync()
    assert np.array_equal(b.digests(), dig1)
    # ... and a reset batch recycles its buffers for the next call, sliced differently
    b.reset()
    b.build_and_execute(rs, kernel, (max(1, slices - 1) if slices else 2) if dev else 1)
    b.sync()
    assert np.array_equal(b.digests(), dig1)
    if not tiles:
        assert np.array_equal(b.download_image()[0], desc1)
    b.close()
    rs.close()


@pytest.mark.parametrize("seed,shape,kernel", [(1, "snv", 7), (3, "snv", 0), (5, "mix", 6), (6, "mix", 7), (10, "long", 0), (12, "long", 6)])
def test_random_streams_in_one_call(built, one, seed, shape, kernel):
    """Irregular streams (empty haplotypes, transcripts without Tasks, gaps, long payloads): whatever path the call takes -- sliced,
    or the one-piece builder's two-pass form / a dense image where the sliced builder declines -- the arena is the expected text."""
    gpu_ctx, dev = one
    from vcf2prot_amd._native import V2PError
    rng = np.random.default_rng(seed)
    proteome, stream, want = random_stream(rng, n_haps=400, n_ref_tx=25, shape=shape, window=4096)
    gpu_ctx.upload_proteome(proteome)
    rs = gpu_ctx.upload_stream(stream)
    b = gpu_ctx.batch()
    try:
        b.build_and_execute(rs, kernel, 3 if dev else 0)
    except V2PError as e:
        assert kernel == 6 and e.code == -9                 # a row with more than 64 descriptors: kernel 6 was asked for by number
        b.reset()
        b.build_and_execute(rs, 7, 3 if dev else 0)
    b.sync()
    for h, w in enumerate(want):
        got = b.download_hap(h)
        assert got.size == w.size and np.array_equal(got, w), (seed, shape, kernel, h)
    b.close()
    rs.close()


@pytest.mark.parametrize("seed,shape,kernel", [(2, "snv", 0), (7, "mix", 6), (11, "long", 7), (13, "mix", 0)])
def test_random_fasta_streams_in_one_call(built, one, seed, shape, kernel):
    """The same irregular streams with FASTA emit (personalized_genome.rs:90-113): every transcript's `>name_1` header and line feed around
    its residues, a few transcripts without a header; first execute and the image's re-execution form (tools/fuzz_one_call.py runs thousands)."""
    gpu_ctx, dev = one
    from vcf2prot_amd._native import V2PError
    rng = np.random.default_rng(seed)
    proteome, headers, stream, want = random_stream(rng, n_haps=300, n_ref_tx=20, shape=shape, window=4096, fasta=True)
    gpu_ctx.upload_reference(proteome, headers)
    rs = gpu_ctx.upload_stream(stream)
    b = gpu_ctx.batch()
    try:
        b.build_and_execute(rs, kernel, 0)
    except V2PError as e:
        assert kernel == 6 and e.code == -9
        b.reset()
        b.build_and_execute(rs, 7, 0)
    b.sync()
    for rep in range(2):
        for h, w in enumerate(want):
            got = b.download_hap(h)
            assert got.size == w.size and np.array_equal(got, w), (seed, shape, kernel, rep, h)
        b.scribble()
        b.execute()
        b.sync()
    b.close()
    rs.close()
    gpu_ctx.upload_proteome(proteome)                       # (leave the shared context without a header table)


def test_a_refused_cut_in_recycled_memory_falls_back(built, one):
    """Found by tools/fuzz_one_call.py (seeds 152 -> 153): the one call launches the chunk-table pass before the host has looked at the
    status word; a cutter that refuses a row (more than 64 descriptors: the call then builds a dense image) used to leave its segment's
    count unwritten -- zero in fresh memory, anything in memory another batch had used: a table pass that followed it wrote out of bounds."""
    gpu_ctx, dev = one
    for seed in (152, 153):
        rng = np.random.default_rng(seed)
        shape = ("snv", "mix", "long")[seed % 3]
        n_haps = int(rng.integers(1, 700)); n_ref = int(rng.integers(1, 40)); window = int(rng.choice([1024, 4096, 8192]))
        proteome, stream, want = random_stream(rng, n_haps=n_haps, n_ref_tx=n_ref, shape=shape, window=window)
        gpu_ctx.upload_proteome(proteome)
        rs = gpu_ctx.upload_stream(stream)
        for kernel in ((0, 8, 0) if dev else (0, 9, 0)):
            b = gpu_ctx.batch()
            try:
                b.build_and_execute(rs, kernel, 0)
            except Exception as e:                          # (a tile image asked for by number may refuse the stream)
                assert kernel == 9 and getattr(e, "code", 0) == -9, e
                b.close()
                continue
            b.sync()
            for rep in range(2):
                for h, w in enumerate(want):
                    assert np.array_equal(b.download_hap(h), w), (seed, kernel, rep, h)
                b.scribble()
                b.execute()
                b.sync()
            b.close()
        rs.close()


def test_no_room_for_the_one_pass_scratch_falls_back_inside_the_call(built, one, coracle):
    """hipErrorOutOfMemory for the one call's scratch (its padded descriptor array is up to 2 KiB per tile): nothing has been launched yet,
    the scratch is released and the call builds in one piece -- whose builder has a two-pass form without that array -- and executes.
    The development library's variant 29 plays the full device."""
    gpu_ctx, dev = one
    if not dev:
        pytest.skip("the A/B switches exist in the development library only")
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset("C3")
    gpu_ctx.upload_proteome(c.proteome())
    stream = c.txstream(40, 100, n_threads=8)
    rs = gpu_ctx.upload_stream(stream)
    stream.close()
    b = gpu_ctx.batch()
    b.build_and_execute(rs, 0, 0); b.sync()
    assert b.oneshot_info()["n_slices"] == 1
    dig = b.digests()
    b.reset()
    gpu_ctx.set_launch_opts(variant=29)
    try:
        b.build_and_execute(rs, 0, 0); b.sync()
    finally:
        gpu_ctx.set_launch_opts()
    assert b.oneshot_info()["n_slices"] == 0                # the one-piece builder ran
    assert np.array_equal(b.digests(), dig)
    assert np.array_equal(b.download_hap(7), oracle_hap(c, coracle, 47))
    b.scribble(); b.execute(); b.sync()
    assert np.array_equal(b.digests(), dig)
    b.close()
    rs.close()


def test_a_tile_that_overflows_its_slots_falls_back_inside_the_call(built, one):
    """One transcript with thousands of Tasks: its tile's descriptors do not fit the one-pass stage, the sliced builder declines and the
    call builds in one piece (two-pass form) and executes -- same bytes, n_slices reported as 0."""
    gpu_ctx, dev = one
    from stream_util import Stream
    rng = np.random.default_rng(5)
    AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    L = 300000
    prot = AA[rng.integers(0, 20, size=L)]
    gpu_ctx.upload_proteome(prot)
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
    rs = gpu_ctx.upload_stream(s)
    for kernel in (6, 7, 0):
        b = gpu_ctx.batch()
        b.build_and_execute(rs, kernel, 2 if dev else 0)
        b.sync()
        assert b.oneshot_info()["n_slices"] == 0
        got = b.download_hap(0)
        assert np.array_equal(got[:L], want) and np.array_equal(got[L:], prot)
        b.close()
    rs.close()


def test_one_call_reports_what_the_reference_would_panic_on(built, one):
    """update_task (haplotype_instruction.rs:154) / Task::execute's slices (task.rs:43,47) in a LATER slice: the first offending task by
    index, and the batch is reusable after a reset."""
    gpu_ctx, dev = one
    from stream_util import Stream
    from vcf2prot_amd._native import V2PError
    prot = np.frombuffer(b"MEDLGENTMVLSTLRSLNNFISQRVEGGSGLEELERGGAKLMNPQRSTVWYACDEFGHIK", dtype=np.uint8)
    gpu_ctx.upload_proteome(prot)
    n_tx = 64 * 64 * 4                                      # enough tiles for several slices

    def stream(bad_tx, bad):
        code = np.tile(np.array([0, 1, 0], dtype=np.uint8), n_tx)
        sp = np.tile(np.array([0, 0, 11], dtype=np.uint32), n_tx)
        ln = np.tile(np.array([10, 1, 49], dtype=np.uint32), n_tx)
        sr = np.tile(np.array([0, 10, 11], dtype=np.uint32), n_tx)
        if bad == "code":
            code[3 * bad_tx + 1] = 2
        elif bad == "res":
            ln[3 * bad_tx + 2] = 50
        elif bad == "src":
            sp[3 * bad_tx + 2] = 30; ln[3 * bad_tx + 2] = 40; sr[3 * bad_tx + 2] = 11
        elif bad == "order":
            sr[3 * bad_tx + 1] = 9
        return Stream([0, n_tx // 2, n_tx], [0] * n_tx, [60] * n_tx, [60] * n_tx, np.arange(0, 3 * n_tx + 1, 3), np.arange(0, 2 * n_tx + 1, 2),
                      code, sp, ln, sr, np.tile(np.frombuffer(b"AC", dtype=np.uint8), n_tx))
    good = gpu_ctx.upload_stream(stream(0, None))
    for bad, want_code, row in (("code", -3, 1), ("res", -4, 2), ("src", -5, 2), ("order", -6, 1)):
        bad_tx = n_tx - 700                                 # in the last slice
        rs = gpu_ctx.upload_stream(stream(bad_tx, bad))
        b = gpu_ctx.batch()
        with pytest.raises(V2PError) as ei:
            b.build_and_execute(rs, 6, 4 if dev else 0)
            b.sync()
        assert ei.value.code == want_code and ei.value.index == 3 * bad_tx + row, (bad, ei.value.code, ei.value.index)
        b.reset()
        b.build_and_execute(good, 6, 4 if dev else 0)
        b.sync()
        one = bytes(prot[:10]) + b"A" + bytes(prot[11:])
        assert b.download_hap(1).tobytes() == one * (n_tx // 2)
        b.close()
        rs.close()
    good.close()


@pytest.mark.parametrize("kernel", [0, 6, 7])
@pytest.mark.parametrize("fasta", [False, True])
def test_reference_task_dumps_in_one_call(one, golden, kernel, fasta):
    """The 36 transcript GIRs harvested from the reference binary (its own Vec<Task> dumps), repeated over 600 haplotypes so that the
    sliced builder has slices to cut; with FASTA emit the arena is the file text of personalized_genome.rs:90-113."""
    gpu_ctx, dev = one
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
    per_hap, reps = 7, 100
    stream = _stream_of_cases(cases * reps, refs, hdr_off * reps, fasta, per_hap)
    rs = gpu_ctx.upload_stream(stream)
    b = gpu_ctx.batch()
    try:
        b.build_and_execute(rs, kernel, 4 if dev else 0)
    except V2PError as e:
        assert kernel == 6 and e.code == -9
        b.reset()
        b.build_and_execute(rs, 7, 4 if dev else 0)
    b.sync()
    many = cases * reps
    for h in range(0, (len(many) + per_hap - 1) // per_hap, 17):
        mine = many[h * per_hap:(h + 1) * per_hap]
        text = b.download_hap(h).tobytes().decode()
        want = "".join(f">{c['name']}_1\n{c['expected']}\n" for c in mine) if fasta else "".join(c["expected"] for c in mine)
        assert text == want, (kernel, fasta, h)
    b.close()
    rs.close()


@pytest.mark.parametrize("kernel", [0, 6, 7, 8, 9])
def test_degenerate_streams_in_one_call(built, one, kernel):
    """No haplotypes; haplotypes without transcripts; transcripts without Tasks (start-lost: an empty GIR, transcript_instructions.rs:338-343)
    and with nothing but cells no Task covers ('.', haplotype_instruction.rs:78); one Task."""
    gpu_ctx, dev = one
    if kernel == 8 and not dev:
        pytest.skip("PATCH images exist in the development library only")
    from stream_util import Stream
    prot = np.frombuffer(b"MEDLGENTMVLSTLRSLNNFISQRVEGGSGLEELERGGAKLMNPQRSTVWYACDEFGHIK", dtype=np.uint8)
    gpu_ctx.upload_proteome(prot)
    z8, z32 = np.zeros(0, np.uint8), np.zeros(0, np.uint32)
    cases = [
        (Stream([0], [], [], [], [0], [0], z8, z32, z32, z32, z8), []),                                               # no haplotypes
        (Stream([0, 0, 0], [], [], [], [0], [0], z8, z32, z32, z32, z8), [b"", b""]),                                 # two haplotypes, no transcripts
        (Stream([0, 2, 3], [0, 0, 0], [60, 60, 60], [0, 7, 60], [0, 0, 0, 1], [0, 0, 0, 0], [0], [0], [60], [0], z8),  # empty GIR, seven '.', one copy
         [b"" + b"." * 7, bytes(prot)]),
    ]
    for s, want in cases:
        rs = gpu_ctx.upload_stream(s)
        b = gpu_ctx.batch()
        try:
            b.build_and_execute(rs, kernel, 0)
        except Exception as e:                              # (a tile image asked for by number refuses a stream without transcripts; the rule builds a rows image)
            assert kernel == 9 and getattr(e, "code", 0) == -9 and s.struct.n_tx == 0, e
            b.reset()
            b.build_and_execute(rs, 0, 0)
        b.sync()
        assert b.counts()["n_haps"] == len(want)
        for h, w in enumerate(want):
            assert b.download_hap(h).tobytes() == w, (kernel, h)
        if want:
            b.scribble(); b.execute(); b.sync()
            assert b.download_hap(len(want) - 1).tobytes() == want[-1]
        b.close()
        rs.close()


@pytest.mark.parametrize("preset,h0,n,slices", [("C3", 100, 400, 1), ("C2", 3, 60, 3), ("C4", 7, 12, 1), ("C1", 0, 8, 1)])
def test_padded_wave_image_is_the_dense_one(built, one, preset, h0, n, slices):
    """A padded wave image (v2p_set_launch_opts variant 24 forces it, 22 forces the compaction; the rule: rich streams): the one call leaves
    the descriptors in their tiles' slots, the chunk records address slots, the launcher stages them phase by phase (variant 23: read in
    place); the arena, the digests and the image a download hands out (dense form) are the compacted build's, chunk order included."""
    gpu_ctx, dev = one
    if not dev:
        pytest.skip("the A/B switches exist in the development library only")
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset(preset)
    gpu_ctx.upload_proteome(c.proteome())
    stream = c.txstream(h0, h0 + n, n_threads=4)
    rs = gpu_ctx.upload_stream(stream)
    stream.close()
    try:
        imgs = {}
        small = dict(phase_min_chunks=16, phase_bytes=1 << 16)   # (phases, hence staging, also on these small images)
        for var in (22, 24):
            gpu_ctx.set_launch_opts(variant=var, **small)
            b = gpu_ctx.batch()
            b.build_and_execute(rs, 6, slices if dev else min(slices, 1))
            b.sync()
            d1 = b.digests()
            if var == 24:
                for keep in (26, 23):                       # executed again as it is: staged, then read in place
                    gpu_ctx.set_launch_opts(variant=keep, **small)
                    b.scribble(); b.execute(); b.sync()
                    assert np.array_equal(b.digests(), d1), keep
            gpu_ctx.set_launch_opts(variant=0, **small)     # ... and as the product does: made dense at the first re-execute
            b.scribble(); b.execute(); b.sync()
            assert np.array_equal(b.digests(), d1)
            b.scribble(); b.execute(); b.sync()
            assert np.array_equal(b.digests(), d1)
            imgs[var] = (b.download_image(), d1, b.counts())
            b.reset()                                       # the batch recycles its buffers for a padded build again
            gpu_ctx.set_launch_opts(variant=24, **small)
            b.build_and_execute(rs, 6, 1); b.sync()
            assert np.array_equal(b.digests(), d1)
            assert np.array_equal(b.download_image()[0], imgs[var][0][0])
            b.close()
    finally:
        gpu_ctx.set_launch_opts()
    (desc0, ch0, hb0), dig0, cn0 = imgs[22]
    (desc1, ch1, hb1), dig1, cn1 = imgs[24]
    assert np.array_equal(dig0, dig1) and cn0 == cn1
    assert np.array_equal(desc0, desc1) and np.array_equal(ch0, ch1) and np.array_equal(hb0, hb1)
    rs.close()


def test_large_uploads_and_whole_arena_downloads(built, gpu_ctx, coracle):
    """A stream of > 100 MB uploaded with its tables checked beside the copy, its arena (~ 470 MB) downloaded in one call and in ragged ranges
    (sizes and offsets off every alignment): the bytes in host memory are the oracle's."""
    import time
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset("C3")
    n = 260                                                              # ~ 125 MB of Task vectors, ~ 470 MB of result
    gpu_ctx.upload_proteome(c.proteome())
    stream = c.txstream(0, n, n_threads=8)
    rs = gpu_ctx.upload_stream(stream)
    stream.close()
    b = gpu_ctx.batch()
    b.build_and_execute(rs, 0, 0); b.sync()
    total = b.counts()["out_bytes"]
    assert total > 300 << 20
    t0 = time.perf_counter()
    whole = b.download(0, total)
    dt = time.perf_counter() - t0
    for h in (0, 1, 77, n - 1):
        lo, ln = b.hap_range(h)
        hap = c.haplotype(h)
        t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
        want = coracle.gir_execute_u8(t, c.ref_tape_u32(h).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8))
        assert np.array_equal(whole[lo:lo + ln], want), h
    dig = b.digests()
    for h in range(0, n, 13):
        lo, ln = b.hap_range(h)
        assert coracle.digest_u8(np.ascontiguousarray(whole[lo:lo + ln])) == int(dig[h]), h
    for begin, length in ((1, (64 << 20) + 3), (12345, total - 12345 - 7), (total - (64 << 20) - 1, (64 << 20) + 1)):
        part = b.download(begin, length)
        assert np.array_equal(part, whole[begin:begin + length]), (begin, length)
    print(f"download of {total / 1e6:.0f} MB: {total / dt / 1e9:.1f} GB/s")
    b.close(); rs.close()


@pytest.mark.parametrize("n_haps", [0, 1, 5])
def test_a_stream_without_a_single_transcript(built, gpu_ctx, n_haps):
    """Haplotypes that carry no transcript at all (an empty slice of a cohort: tools/fuzz_pipeline.py cuts them) -- one tile of no items:
    every builder takes it, the arena is empty, every haplotype's range is [0, 0)."""
    from vcf2prot_amd._native import V2PError
    from stream_util import Stream
    z64, z32, z8 = np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.uint8)
    gpu_ctx.upload_proteome(np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY" * 10, dtype=np.uint8))
    st = Stream(np.zeros(n_haps + 1, dtype=np.uint64), z64, z32, z32, np.zeros(1, dtype=np.uint64), np.zeros(1, dtype=np.uint64), z8, z32, z32, z32, z8)
    rs = gpu_ctx.upload_stream(st)
    for kernel in (0, 6, 7, 9):
        b = gpu_ctx.batch()
        try:
            b.build_and_execute(rs, kernel, 0)
        except V2PError as e:
            assert kernel == 9 and e.code == -9, (kernel, e)               # (no transcripts: no tile image by number)
            b.close()
            continue
        b.sync()
        cn = b.counts()
        assert cn["n_haps"] == n_haps and cn["out_bytes"] == 0, cn
        for h in range(n_haps):
            assert b.hap_range(h) == (0, 0)
        b.execute(); b.sync()
        b.close()
    rs.close()


def test_a_tile_of_more_than_2_gib_is_refused(built, gpu_ctx):
    """Positions inside a tile of transcripts are 32-bit offsets from its first byte: 64 consecutive transcripts with more than 2 GiB of result
    between them (here: 64 x 40 MB of '.', no Task at all) are V2P_ERR_UNSUPPORTED from every builder -- reported, never wrapped -- and the
    batch takes an ordinary stream afterwards."""
    from vcf2prot_amd._native import V2PError
    from stream_util import Stream, regular_stream
    n_tx = 64
    z32, z8 = np.zeros(0, dtype=np.uint32), np.zeros(0, dtype=np.uint8)
    gpu_ctx.upload_proteome(np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY" * 10, dtype=np.uint8))
    st = Stream(np.array([0, n_tx], dtype=np.uint64), np.zeros(n_tx, dtype=np.uint64), np.full(n_tx, 10, dtype=np.uint32), np.full(n_tx, 40_000_000, dtype=np.uint32),
                np.zeros(n_tx + 1, dtype=np.uint64), np.zeros(n_tx + 1, dtype=np.uint64), z8, z32, z32, z32, z8)
    rs = gpu_ctx.upload_stream(st)
    b = gpu_ctx.batch()
    for kernel in (0, 6, 7, 9):
        with pytest.raises(V2PError) as e:
            b.build_and_execute(rs, kernel, 0)
        assert e.value.code == -9, (kernel, e.value)
        b.reset()
    rs.close()
    proteome, stream, want = regular_stream(n_haps=4, tx_per_hap=6, seed=3)
    gpu_ctx.upload_proteome(proteome)
    rs = gpu_ctx.upload_stream(stream)
    b.build_and_execute(rs, 0, 0); b.sync()
    for h in range(4):
        assert np.array_equal(b.download_hap(h), want(h))
    b.close(); rs.close()
