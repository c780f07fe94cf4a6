"""CPU suite: PATCH images (round 5; csrc/patch_format.hpp, patch_image_host.hpp) -- the sequential restatement of the device builder's rules and
the interpreter that executes such an image cell by cell.  The executed image must be the oracle's tapes (task.rs:38-50 per Task, '.' where
no Task writes: haplotype_instruction.rs:78) for the preset cohorts, irregular random streams and the reference binary's own Task dumps
(plain and as FASTA text, personalized_genome.rs:90-113); the image must be well formed (the interpreter refuses a cell written twice or
never, a patch outside a reference segment); the reference's panics come back by Task index."""
import numpy as np
import pytest

from stream_util import Stream, random_stream


def _alt_of(stream):
    s = stream.struct
    n = int(s.n_alt)
    return np.ctypeslib.as_array(s.alt, shape=(n + 1,))[:n].copy() if n else np.zeros(0, np.uint8)


def _run(stream, resident):
    from vcf2prot_amd.txstream import interpret_patch, pack_patch
    seg, patch, chunks, hb, out_bytes, n_seg, n_patch = pack_patch(stream, resident.size if not isinstance(resident, tuple) else resident[1])
    src0 = resident if not isinstance(resident, tuple) else resident[0]
    out = interpret_patch(seg, patch, chunks, src0, _alt_of(stream), out_bytes)
    assert out is not None, "the interpreter refused the image"
    return out, hb, n_seg, n_patch, chunks


@pytest.mark.parametrize("preset,h0,n", [("C5", 7, 30), ("C1", 0, 8), ("C2", 3, 6), ("C3", 100, 40), ("C4", 2, 2)])
def test_host_patch_image_of_the_presets_executes_to_the_oracle(built, coracle, preset, h0, n):
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset(preset)
    prot = c.proteome()
    stream = c.txstream(h0, h0 + n, n_threads=2)
    out, hb, n_seg, n_patch, chunks = _run(stream, prot)
    sizes = c.result_sizes(h0, h0 + n)
    assert np.array_equal(np.diff(hb.astype(np.int64)), sizes.astype(np.int64))
    if preset == "C5":
        assert n_patch > n_seg
    for i in range(n):
        hap = c.haplotype(h0 + i)
        t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
        want = coracle.gir_execute_u8(t, c.ref_tape_u32(h0 + i).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8))
        assert np.array_equal(out[int(hb[i]):int(hb[i + 1])], want), (preset, h0 + i)
    stream.close()


@pytest.mark.parametrize("seed,shape", [(1, "snv"), (2, "snv"), (5, "mix"), (6, "mix"), (10, "long"), (11, "long"), (21, "snv"), (22, "mix")])
def test_host_patch_image_of_random_streams(built, seed, shape):
    rng = np.random.default_rng(seed)
    proteome, stream, want = random_stream(rng, n_haps=50, n_ref_tx=25, shape=shape, window=4096)
    out, hb, _, _, _ = _run(stream, proteome)
    for h, w in enumerate(want):
        got = out[int(hb[h]):int(hb[h + 1])]
        assert got.size == w.size and np.array_equal(got, w), (seed, shape, h)


@pytest.mark.parametrize("fasta", [False, True])
def test_host_patch_image_of_the_reference_task_dumps(built, golden, fasta):
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
    resident = np.concatenate([proteome, np.frombuffer(headers.encode(), dtype=np.uint8)])      # [proteome | record headers], as v2p_upload_reference lays it out
    stream = _stream_of_cases(cases * 9, refs, hdr_off * 9, fasta, 7)
    out, hb, _, _, _ = _run(stream, (resident, proteome.size))
    many = cases * 9
    for h in range((len(many) + 6) // 7):
        mine = many[h * 7:(h + 1) * 7]
        text = out[int(hb[h]):int(hb[h + 1])].tobytes().decode()
        want = "".join(f">{c['name']}_1\n{c['expected']}\n" for c in mine) if fasta else "".join(c["expected"] for c in mine)
        assert text == want, (fasta, h)


def test_host_patch_builder_reports_the_reference_panics_and_declines_what_does_not_fit(built):
    from vcf2prot_amd.txstream import RowsError, pack_patch
    prot = np.frombuffer(b"MEDLGENTMVLSTLRSLNNFISQRVEGGSGLEELERGGAKLMNPQRSTVWYACDEFGHIK", dtype=np.uint8)

    def stream(code, sp, ln, sr):
        return Stream([0, 1], [0], [60], [60], [0, len(code)], [0, 2], code, sp, ln, sr, np.frombuffer(b"AC", dtype=np.uint8))
    for (code, sp, ln, sr), reason, row in [
            (([0, 2, 0], [0, 0, 11], [10, 1, 49], [0, 10, 11]), 1, 1),      # bad exe code (haplotype_instruction.rs:154)
            (([0, 1, 0], [0, 0, 11], [10, 1, 50], [0, 10, 11]), 2, 2),      # result out of bounds (task.rs:43,47)
            (([0, 1, 0], [0, 0, 30], [10, 1, 40], [0, 10, 11]), 3, 2),      # source out of bounds
            (([0, 1, 0], [0, 0, 11], [10, 1, 49], [0, 9, 11]), 4, 1)]:      # overlapping result ranges
        with pytest.raises(RowsError) as ei:
            pack_patch(stream(code, sp, ln, sr), prot.size)
        assert (ei.value.reason, ei.value.index) == (reason, row)
    # every second residue substituted: 4 096 patches in an 8 KiB window, more than its 1 024 slots -- declined (reason 9), not mangled
    rng = np.random.default_rng(3)
    AA = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWY", dtype=np.uint8)
    L = 30000
    big = AA[rng.integers(0, 20, size=L)]
    n = L // 2
    code = np.tile(np.array([0, 1], dtype=np.uint8), n)
    sp = np.empty(2 * n, dtype=np.uint32); sp[0::2] = np.arange(0, L, 2); sp[1::2] = np.arange(n)
    s = Stream([0, 1], [0], [L], [L], [0, 2 * n], [0, n], code, sp, np.ones(2 * n, dtype=np.uint32), np.arange(2 * n, dtype=np.uint32), AA[rng.integers(0, 20, size=n)])
    with pytest.raises(RowsError) as ei:
        pack_patch(s, big.size)
    assert ei.value.reason == 9
