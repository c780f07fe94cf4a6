"""ROWS images on the host (csrc/rows_image.hpp; no GPU): the image v2p_batch_build_on_device(kernel 6 / 7) builds in one pass.

* the sequential restatement of the parse (the packer's fusion state machine, nothing cut) interpreted in numpy gives every
  haplotype the oracle's / the generator's bytes;
* the EMULATION of the device kernel -- tiles of K transcripts, 64-item windows with context and look-ahead lanes, the fusion
  solved on ballot masks (rows_parse), the per-row cover map -- produces the same descriptors byte for byte, for every tile size;
* the cutter: chunks tile the arena on 1 KiB rows, <= 64 descriptors and <= 10 rows (wave) / <= 1024 and <= 12 (dense), a
  descriptor across a cut is shared (head skip / row clip)."""
import numpy as np
import pytest

from gen_util import interpret_image, _desc_len
from stream_util import random_stream


def _check_geometry(img, mode, out_bytes):
    ch = img.chunks
    if out_bytes == 0:
        assert ch.shape[0] == 0
        return
    dn = ch[:, 1]
    assert ((dn >> np.uint64(59)) & np.uint64(1)).all()
    dst = (dn & np.uint64((1 << 48) - 1)).astype(np.int64)
    nd = ((dn >> np.uint64(48)) & np.uint64(0x7FF)).astype(np.int64)
    flag = (dn >> np.uint64(60)).astype(np.int64)
    assert (flag == (1 if mode == 1 else 2)).all()
    assert dst[0] == 0 and (np.diff(dst) > 0).all() and (dst % 1024 == 0).all()
    ends = np.concatenate([dst[1:], [out_bytes]])
    # head skip / tail clip: what the chunk's descriptors produce minus both is exactly its rows
    lens = np.array([_desc_len(int(d)) for d in img.desc], dtype=np.int64)
    cum = np.concatenate([[0], np.cumsum(lens)])
    tb = (ch[:, 0] & np.uint64((1 << 42) - 1)).astype(np.int64)
    skip = ((ch[:, 0] >> np.uint64(42)) & np.uint64(0x7FF)).astype(np.int64)
    clip = (ch[:, 0] >> np.uint64(53)).astype(np.int64)
    assert ((cum[tb + nd] - cum[tb]) - skip - clip == ends - dst).all()
    assert lens.max() <= 2047
    assert nd.max() <= (64 if mode == 1 else 1024) and (ends - dst).max() <= (10240 if mode == 1 else 12288)
    assert ((dst // (640 * 1024)) == ((ends - 1) // (640 * 1024))).all()          # no chunk crosses a segment of the cutter


@pytest.mark.parametrize("preset,h0,n", [("C1", 0, 8), ("C2", 1, 2), ("C3", 40, 4), ("C4", 3, 2), ("C5", 7, 12)])
@pytest.mark.parametrize("mode", [1, 2])
def test_rows_image_of_the_preset_cohorts(built, coracle, preset, h0, n, mode):
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.txstream import pack_rows, RowsError
    c = Cohort.preset(preset)
    prot = c.proteome()
    s = c.txstream(h0, h0 + n, n_threads=2)
    try:
        ref = pack_rows(s, prot.size, mode, 0)
    except RowsError as e:
        assert mode == 1 and preset == "C5" and e.reason == 5       # 7 bytes per task: more than 64 descriptors in a 1 KiB row
        return
    _check_geometry(ref, mode, ref.out_bytes)
    got = interpret_image(ref.desc, ref.chunks, prot, ref.payload, ref.out_bytes)
    base = c.pack(h0, h0 + n, n_threads=1, kernel=2)
    assert np.array_equal(ref.hap_out_begin, base.hap_out_begin)
    assert np.array_equal(got, interpret_image(base.desc, base.chunks, prot, base.payload, base.out_bytes))
    for k in (1, 2, 8, 41, 64):
        emu = pack_rows(s, prot.size, mode, k)
        assert np.array_equal(emu.desc, ref.desc), (preset, mode, k, int(np.argmax(emu.desc[:min(emu.desc.size, ref.desc.size)] != ref.desc[:min(emu.desc.size, ref.desc.size)])))
        assert np.array_equal(emu.chunks, ref.chunks) and np.array_equal(emu.hap_out_begin, ref.hap_out_begin)
    if mode == 1 and preset in ("C2", "C3"):
        # the wave rule IS the packer's: the host wave image holds the same fused substitutions
        w = c.pack(h0, h0 + n, n_threads=1, kernel=4)
        assert abs(int(((w.desc >> np.uint64(61)) == 7).sum()) - int(((ref.desc >> np.uint64(61)) == 7).sum())) <= w.chunks.shape[0]


@pytest.mark.parametrize("seed,shape", [(1, "snv"), (2, "snv"), (3, "mix"), (4, "mix"), (5, "long"), (6, "long"), (7, "snv"), (8, "mix")])
@pytest.mark.parametrize("mode", [1, 2])
def test_rows_image_of_random_streams(built, seed, shape, mode):
    from vcf2prot_amd.txstream import pack_rows, RowsError
    rng = np.random.default_rng(seed)
    proteome, stream, want = random_stream(rng, n_haps=40, n_ref_tx=25, shape=shape, window=4096)
    try:
        ref = pack_rows(stream, proteome.size, mode, 0)
    except RowsError as e:
        assert mode == 1 and e.reason == 5
        ref = None
    if ref is not None:
        _check_geometry(ref, mode, ref.out_bytes)
        got = interpret_image(ref.desc, ref.chunks, proteome, ref.payload, ref.out_bytes)
        assert np.array_equal(np.diff(ref.hap_out_begin.astype(np.int64)), [w.size for w in want])
        assert np.array_equal(got, np.concatenate(want) if want else np.zeros(0, np.uint8))
    for k in (1, 3, 16, 61, 64):
        try:
            emu = pack_rows(stream, proteome.size, mode, k)
        except RowsError as e:
            assert ref is None and e.reason == 5
            continue
        assert ref is not None
        assert np.array_equal(emu.desc, ref.desc), (seed, shape, mode, k)
        assert np.array_equal(emu.chunks, ref.chunks) and np.array_equal(emu.hap_out_begin, ref.hap_out_begin)


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("fasta", [False, True])
def test_rows_image_of_the_reference_task_dumps(built, golden, mode, fasta):
    """The 36 transcript GIRs harvested from the reference binary, several per haplotype, plain and with FASTA emit (header, line
    feed: personalized_genome.rs:90-113): the sequential restatement, the lane-by-lane emulation of the device kernel for several
    tile sizes, and the text the binary wrote."""
    from test_gpu_device_build_fasta import _stream_of_cases
    from vcf2prot_amd.txstream import pack_rows, RowsError
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
    resident = np.concatenate([proteome, np.frombuffer(headers.encode(), dtype=np.uint8)])
    per_hap = 7
    stream = _stream_of_cases(cases, refs, hdr_off, fasta, per_hap)
    try:
        ref = pack_rows(stream, proteome.size, mode, 0)
    except RowsError as e:
        assert mode == 1 and e.reason == 5
        return
    _check_geometry(ref, mode, ref.out_bytes)
    text = interpret_image(ref.desc, ref.chunks, resident, ref.payload, ref.out_bytes).tobytes().decode()
    want = "".join(f">{c['name']}_1\n{c['expected']}\n" for c in cases) if fasta else "".join(c["expected"] for c in cases)
    assert text == want
    for k in (1, 2, 5, 7, 36, 64):
        emu = pack_rows(stream, proteome.size, mode, k)
        assert np.array_equal(emu.desc, ref.desc), (mode, fasta, k)
        assert np.array_equal(emu.chunks, ref.chunks) and np.array_equal(emu.hap_out_begin, ref.hap_out_begin)
