"""SURVEY 8f ranks 1 and 2 composed: the device-side image builder (v2p_batch_build_on_device: step 5 of
haplotype_instruction.rs:94-133 as prefix scans, image packing as kernels) also emits the FASTA record text of
personalized_genome.rs:90-113, and takes the Task vectors the reference binary itself printed."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stream_of_cases(cases, refs, hdr_off, fasta, per_hap):
    from vcf2prot_amd.txstream import TxStreamBuilder
    sb = TxStreamBuilder(fasta=fasta)
    for i, c in enumerate(cases):
        t = np.array(c["tasks"], dtype=np.uint64).reshape(-1, 4)
        sb.add_transcript(t[:, 0].astype(np.uint8), t[:, 1], t[:, 2], t[:, 3], refs[c["ref"]], len(c["ref"]),
                          np.frombuffer(c["alt"].encode(), dtype=np.uint8), c["res_len"],
                          hdr_off[i] if fasta else 0, len(c["name"]) + 4 if fasta else 0)
        if (i + 1) % per_hap == 0:
            sb.end_haplotype()
    if len(cases) % per_hap:
        sb.end_haplotype()
    return sb.finish()


def _build(b, stream, window, kernel):
    """A wave image (kernel 4) holds at most 64 descriptors per window: for these short transcripts the builder may refuse it
    (V2P_ERR_UNSUPPORTED, the batch left reusable) -- then the same batch is built per-block, as txstream.build_on_device_auto does."""
    from vcf2prot_amd._native import V2PError
    try:
        b.build_on_device(stream, window, kernel)
    except V2PError as e:
        assert kernel == 4 and e.code == -9, e
        b.build_on_device(stream, 4096, 2)


@pytest.mark.parametrize("kernel,window", [(1, 4096), (2, 4096), (3, 4096), (4, 1024), (4, 10240), (2, 32768)])
@pytest.mark.parametrize("fasta", [False, True])
def test_reference_task_dumps_through_the_device_builder(dev_ctx, golden, kernel, window, fasta):
    """The 36 transcript GIRs harvested from the reference binary (its own Vec<Task> dumps, transcript_instructions.rs:372-382,
    un-rebased), several per haplotype: device-built image, executed, every record the sequence the binary wrote -- incl. the empty
    start-lost record and the '.' cell of test_correct_translation_20; with FASTA emit the arena is the file text."""
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
    per_hap = 7
    stream = _stream_of_cases(cases, refs, hdr_off, fasta, per_hap)
    b = dev_ctx.batch()
    _build(b, stream, window, kernel)
    b.execute()
    b.sync()
    for h in range(0, (len(cases) + per_hap - 1) // per_hap):
        mine = cases[h * per_hap:(h + 1) * per_hap]
        text = b.download_hap(h).tobytes().decode()
        want = "".join(f">{c['name']}_1\n{c['expected']}\n" for c in mine) if fasta else "".join(c["expected"] for c in mine)
        assert text == want, (kernel, window, fasta, h)
    b.close()


@pytest.mark.parametrize("kernel,window", [(2, 4096), (3, 8192), (4, 1024), (1, 4096)])
def test_random_reference_task_vectors_through_the_device_builder(dev_ctx, kernel, window):
    """The 545 random single-transcript cases the reference binary answered with a record (tests/golden/kat_random.json: its printed
    Vec<Task> and FASTA record): all of them as one batch through the device builder, FASTA emit on -- every record as the binary
    wrote it."""
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "kat_random.json")))
    cases = [c for c in kat["cases"] if not c["panics"]]
    assert len(cases) >= 500
    refs, off = {}, 0
    for c in cases:
        if c["ref"] not in refs:
            refs[c["ref"]] = off
            off += len(c["ref"])
    proteome = np.frombuffer("".join(refs).encode(), dtype=np.uint8)
    names = [f"{c['transcript']}_1" for c in cases]                      # "<transcript>_<h>" as the binary names its records
    headers = "\n" + "".join(f">{nm}\n" for nm in names)
    from vcf2prot_amd.txstream import TxStreamBuilder
    sb = TxStreamBuilder(fasta=True)
    o = 1
    for i, c in enumerate(cases):
        t = np.array(c["tasks"], dtype=np.uint64).reshape(-1, 4)
        res_len = len(c["record"])
        alt = np.frombuffer(c["alt"].encode(), dtype=np.uint8) if c.get("alt") else np.zeros(0, np.uint8)
        sb.add_transcript(t[:, 0].astype(np.uint8), t[:, 1], t[:, 2], t[:, 3], refs[c["ref"]], len(c["ref"]), alt, res_len, o, len(names[i]) + 2)
        o += len(names[i]) + 2
        if (i + 1) % 40 == 0:
            sb.end_haplotype()
    sb.end_haplotype()
    stream = sb.finish()
    dev_ctx.upload_reference(proteome, np.frombuffer(headers.encode(), dtype=np.uint8))
    b = dev_ctx.batch()
    _build(b, stream, window, kernel)
    b.execute()
    b.sync()
    k = 0
    for h in range(int(stream.struct.n_haps)):
        text = b.download_hap(h).tobytes().decode()
        n_here = min(40, len(cases) - k)
        want = "".join(f">{names[i]}\n{cases[i]['record']}\n" for i in range(k, k + n_here))
        assert text == want, (kernel, window, h)
        k += n_here
    assert k == len(cases)
    b.close()


@pytest.mark.parametrize("preset,h0,n,kernel,window", [("C1", 0, 8, 2, 4096), ("C1", 0, 8, 3, 4096), ("C3", 30, 6, 2, 16384), ("C3", 30, 6, 4, 1024),
                                                       ("C2", 2, 2, 4, 4096), ("C5", 5, 12, 3, 4096)])
def test_device_built_fasta_equals_host_built_fasta(built, dev_ctx, preset, h0, n, kernel, window):
    """File-ready arena (header, residues, line feed per record) from the device builder == the one the host packer
    (ImageBuilder::add_literal between the tasks) produces for the same haplotypes, byte for byte."""
    import ctypes
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset(preset)
    dev_ctx.upload_reference(c.proteome(), c.fasta_headers())
    host = c.pack(h0, h0 + n, n_threads=2, fasta=True)
    b = dev_ctx.batch()
    b.set_packed(host.desc, host.chunks, host.payload, host.hap_out_begin)
    b.finalize()
    b.execute()
    b.sync()
    want = [b.download_hap(i) for i in range(n)]
    b.close()
    stream = c.txstream(h0, h0 + n, n_threads=2)
    hdr_off, hdr_len = [], []
    for h in range(h0, h0 + n):
        hap = c.haplotype(h)
        for tx in hap.tx_id:
            hdr_off.append(1 + (2 * int(tx) + (h & 1)) * Cohort.HEADER_BYTES)
            hdr_len.append(Cohort.HEADER_BYTES)
    assert len(hdr_off) == stream.n_tx
    ho, hl = np.array(hdr_off, dtype=np.uint64), np.array(hdr_len, dtype=np.uint32)
    stream.struct.tx_header_off = ho.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
    stream.struct.tx_header_len = hl.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    d = dev_ctx.batch()
    d.build_on_device(stream, window, kernel)
    d.execute()
    d.sync()
    for i in range(n):
        got = d.download_hap(i)
        assert got.size == want[i].size and np.array_equal(got, want[i]), (preset, kernel, i)
    d.close()
    stream.close()
