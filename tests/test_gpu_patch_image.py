"""PATCH images (round 5; kernel 8: vcf2prot_amd/csrc/patch_image.h / .hip): deep Task vectors as SEGMENTS (runs of one source, the
reference going on beneath substituted residues) + PATCHES (the substituted residues) on a fixed grid of 8 KiB chunks, built by one
kernel and executed by stitch_patch_kernel.  The arena must be the oracle's tapes (task.rs:38-50 per Task, '.' where no Task writes:
haplotype_instruction.rs:78), the panics the reference's (haplotype_instruction.rs:140-158, task.rs:43,47), and the image well formed:
every byte of a chunk covered by exactly one segment, every patch inside a reference segment."""
import numpy as np
import pytest

from stream_util import random_stream

pytestmark = pytest.mark.gpu

G, SEG_CAP, PATCH_CAP = 8192, 1024, 1024


def oracle_hap(c, coracle, h):
    hap = c.haplotype(h)
    t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
    return coracle.gir_execute_u8(t, c.ref_tape_u32(h).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8))


def check_image(b):
    """every chunk: segments tile [0, span) exactly, patches sit on distinct cells inside reference segments"""
    seg, patch, chunks, n_seg, n_patch = b.download_patch_image()
    out_bytes = b.counts()["out_bytes"]
    tot_s = tot_p = 0
    for tb, dn in chunks:
        dst = int(dn) & ((1 << 48) - 1)
        ns, npatch = (int(dn) >> 48) & 0x7FF, (int(tb) >> 42) & 0xFFF
        assert (int(dn) >> 60) & 3 == 3 and dst % G == 0 and int(tb) & ((1 << 42) - 1) == (dst // G) * SEG_CAP
        k = dst // G
        span = min(G, out_bytes - dst)
        w = seg[k, :ns]
        start = ((w >> np.uint64(34)) & np.uint64(0x3FFF)).astype(np.int64)
        ln = ((w >> np.uint64(48)) & np.uint64(0x3FFF)).astype(np.int64)
        space = (w >> np.uint64(62)).astype(np.int64)
        o = np.argsort(start, kind="stable")
        assert ns >= 1 and start[o][0] == 0 and np.all(ln > 0)
        assert np.array_equal(start[o][1:], (start[o] + ln[o])[:-1]) and start[o][-1] + ln[o][-1] == span, (k, "segments do not tile the chunk")
        pos = (patch[k, :npatch] & np.uint32(0x3FFF)).astype(np.int64)
        assert np.unique(pos).size == pos.size and (pos.size == 0 or pos.max() < span)
        if pos.size:
            owner = np.searchsorted(start[o], pos, side="right") - 1
            assert np.all(space[o][owner] == 0), (k, "a patch outside a reference segment")
        tot_s += ns
        tot_p += npatch
    assert tot_s == n_seg and tot_p == n_patch
    return n_seg, n_patch


@pytest.mark.parametrize("preset,h0,n", [("C5", 50, 300), ("C5", 0, 64), ("C1", 0, 8), ("C2", 5, 12), ("C3", 100, 60), ("C4", 7, 3)])
def test_patch_image_of_the_preset_cohorts(built, dev_ctx, coracle, preset, h0, n):
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset(preset)
    dev_ctx.upload_proteome(c.proteome())
    stream = c.txstream(h0, h0 + n, n_threads=3)
    b = dev_ctx.batch()
    ms = b.build_on_device(stream, 0, 8)
    assert ms > 0
    sizes = c.result_sizes(h0, h0 + n)
    at = 0
    for i in range(n):
        assert b.hap_range(i) == (at, int(sizes[i]))
        at += int(sizes[i])
    n_seg, n_patch = check_image(b)
    if preset == "C5":
        assert n_patch > n_seg                                            # (what the format is for: most alterations are patches)
    b.execute()
    b.sync()
    for i in range(0, n, max(1, n // 40)):
        assert np.array_equal(b.download_hap(i), oracle_hap(c, coracle, h0 + i)), (preset, h0 + i)
    # the same through a resident stream and the one call
    rs = dev_ctx.upload_stream(stream)
    stream.close()
    b2 = dev_ctx.batch()
    b2.build_and_execute(rs, 8, 0)
    b2.sync()
    assert b2.oneshot_info()["kernel"] == 8
    assert np.array_equal(b2.digests(), b.digests())
    b.close(); b2.close(); rs.close()


@pytest.mark.parametrize("seed,shape", [(1, "snv"), (2, "snv"), (3, "snv"), (5, "mix"), (6, "mix"), (7, "mix"), (10, "long"), (11, "long"), (13, "mix"), (14, "long")])
def test_random_streams_as_patch_images(built, dev_ctx, seed, shape):
    """Irregular streams: empty haplotypes, transcripts without Tasks, zero-length Tasks, cells no Task covers, alt payloads from one byte
    to longer than a chunk, substitution triples at every distance from a chunk boundary."""
    rng = np.random.default_rng(seed)
    proteome, stream, want = random_stream(rng, n_haps=60, n_ref_tx=25, shape=shape, window=4096)
    dev_ctx.upload_proteome(proteome)
    b = dev_ctx.batch()
    b.build_on_device(stream, 0, 8)
    check_image(b)
    b.execute()
    b.sync()
    for h, w in enumerate(want):
        got = b.download_hap(h)
        assert got.size == w.size and np.array_equal(got, w), (seed, shape, h, int(np.argmax(got != w)) if got.size == w.size else -1)
    b.close()


@pytest.mark.parametrize("fasta", [False, True])
def test_reference_task_dumps_as_a_patch_image(dev_ctx, golden, fasta):
    """The 36 transcript GIRs harvested from the reference binary (its own Vec<Task> dumps), several per haplotype, repeated so that the
    arena has many chunks; with FASTA emit the arena is the file text of personalized_genome.rs:90-113."""
    from test_gpu_device_build_fasta import _stream_of_cases
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
    dev_ctx.upload_reference(proteome, np.frombuffer(headers.encode(), dtype=np.uint8))
    per_hap, reps = 7, 40
    many = cases * reps
    stream = _stream_of_cases(many, refs, hdr_off * reps, fasta, per_hap)
    b = dev_ctx.batch()
    b.build_on_device(stream, 0, 8)
    check_image(b)
    b.execute()
    b.sync()
    for h in range(0, (len(many) + per_hap - 1) // per_hap, 3):
        mine = many[h * per_hap:(h + 1) * per_hap]
        text = b.download_hap(h).tobytes().decode()
        want = "".join(f">{c['name']}_1\n{c['expected']}\n" for c in mine) if fasta else "".join(c["expected"] for c in mine)
        assert text == want, (fasta, h)
    b.close()


def test_patch_builder_reports_what_the_reference_would_panic_on(built, dev_ctx):
    """update_task (haplotype_instruction.rs:154) and Task::execute's slices (task.rs:43,47): the first offending Task, by index."""
    from stream_util import Stream
    from vcf2prot_amd._native import V2PError
    prot = np.frombuffer(b"MEDLGENTMVLSTLRSLNNFISQRVEGGSGLEELERGGAKLMNPQRSTVWYACDEFGHIK", dtype=np.uint8)
    dev_ctx.upload_proteome(prot)
    n_tx = 900                                             # several chunks

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
    for bad, want_code, row in (("code", -3, 1), ("res", -4, 2), ("src", -5, 2), ("order", -6, 1)):
        for bad_tx in (0, 455, n_tx - 1):
            b = dev_ctx.batch()
            with pytest.raises(V2PError) as ei:
                b.build_on_device(stream(bad_tx, bad), 0, 8)
            assert ei.value.code == want_code and ei.value.index == 3 * bad_tx + row, (bad, bad_tx, ei.value.code, ei.value.index)
            b.build_on_device(stream(0, None), 0, 8)                       # the batch is reusable
            b.scribble(); b.execute(); b.sync()
            one = bytes(prot[:10]) + b"A" + bytes(prot[11:])
            assert b.download_hap(1).tobytes() == one * (n_tx - n_tx // 2)
            b.close()


def test_a_window_with_too_many_segments_is_declined_not_mangled(built, dev_ctx):
    """Every second residue substituted: 4 096 patches in an 8 KiB window, more than its 1 024 slots -- V2P_ERR_UNSUPPORTED, the batch left
    empty and reusable; the dense rows image (kernel 7) takes the stream."""
    from stream_util import Stream
    from vcf2prot_amd._native import V2PError
    rng = np.random.default_rng(3)
    AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    L = 30000
    prot = AA[rng.integers(0, 20, size=L)]
    dev_ctx.upload_proteome(prot)
    n = L // 2
    code = np.tile(np.array([0, 1], dtype=np.uint8), n)
    sp = np.empty(2 * n, dtype=np.uint32); sp[0::2] = np.arange(0, L, 2); sp[1::2] = np.arange(n)
    ln = np.ones(2 * n, dtype=np.uint32)
    sr = np.arange(2 * n, dtype=np.uint32)
    alt = AA[rng.integers(0, 20, size=n)]
    s = Stream([0, 1], [0], [L], [L], [0, 2 * n], [0, n], code, sp, ln, sr, alt)
    want = np.empty(L, dtype=np.uint8); want[0::2] = prot[0::2]; want[1::2] = alt
    b = dev_ctx.batch()
    with pytest.raises(V2PError) as ei:
        b.build_on_device(s, 0, 8)
    assert ei.value.code == -9
    b.build_on_device(s, 0, 7)
    b.scribble(); b.execute(); b.sync()
    assert np.array_equal(b.download_hap(0), want)
    b.close()
