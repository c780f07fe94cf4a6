"""FASTA emit fused into the scatter.

The device image interleaves record headers and line feeds (resident header table behind the
proteome) with the tasks, so a haplotype's arena range is the file text the reference writes
(personalized_genome.rs:90-113: ">{name}_{1|2}\\n{seq}\\n" per annotated transcript).  Parity is
on the SET of (header, sequence) records -- the reference iterates a HashMap (random order);
the device order is the annotation order."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_fasta(text: bytes):
    lines = text.decode().split("\n")
    assert lines[-1] == ""
    recs = []
    for i in range(0, len(lines) - 1, 2):
        assert lines[i].startswith(">")
        recs.append((lines[i][1:], lines[i + 1]))
    return recs


def interpret(img, resident):
    """CPU reading of the packed image (host logic check)."""
    from gen_util import interpret_image
    return interpret_image(img.desc, img.chunks, resident, img.payload, img.out_bytes)


@pytest.fixture(scope="module")
def c1(built):
    from vcf2prot_amd.cohort import Cohort
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "c1_example.json")))
    return Cohort.preset(gold["preset"]), gold


def test_fasta_image_reproduces_reference_files(c1):
    cohort, gold = c1
    img = cohort.pack(0, cohort.n_haplotypes, n_threads=2, fasta=True)
    resident = np.concatenate([cohort.proteome(), cohort.fasta_headers()])
    out = interpret(img, resident)
    for s, sample in enumerate(gold["samples"]):      # a sample's file = haplotype 1 records then haplotype 2 records
        a, b = int(img.hap_out_begin[2 * s]), int(img.hap_out_begin[2 * s + 2])
        assert sorted(parse_fasta(out[a:b].tobytes())) == sorted(tuple(r) for r in gold["fasta"][sample]), sample
    plain = cohort.pack(0, cohort.n_haplotypes, n_threads=2)
    n_records = sum(len(v) for v in gold["fasta"].values())
    assert img.out_bytes == plain.out_bytes + n_records * 20          # 19-byte header + line feed per record
    assert img.n_tasks == plain.n_tasks and img.n_copy_bytes == plain.n_copy_bytes


@pytest.mark.gpu
def test_fasta_emit_on_gpu_matches_reference_files(c1, gpu_ctx):
    cohort, gold = c1
    gpu_ctx.upload_reference(cohort.proteome(), cohort.fasta_headers())
    b = gpu_ctx.batch()
    for h in range(cohort.n_haplotypes):
        hap = cohort.haplotype(h)
        hdr_off = np.uint64(1) + (2 * hap.tx_id.astype(np.uint64) + np.uint64(h & 1)) * np.uint64(cohort.HEADER_BYTES)
        b.add_haplotype_fasta(hap.code, hap.start_pos, hap.length, hap.start_pos_res, hap.seg_ref_begin, hap.seg_proteome_off,
                              hap.alt, hap.n_res, hap.tx_res_end, hdr_off, np.full(hap.tx_id.size, cohort.HEADER_BYTES, np.uint32))
    b.finalize()
    b.execute()
    b.sync()
    for s, sample in enumerate(gold["samples"]):
        text = b.download_hap(2 * s).tobytes() + b.download_hap(2 * s + 1).tobytes()
        assert sorted(parse_fasta(text)) == sorted(tuple(r) for r in gold["fasta"][sample]), sample
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("preset,h0,n", [("C3", 40, 24), ("C5", 0, 64)])
def test_packed_fasta_image_on_gpu_matches_oracle_records(built, gpu_ctx, coracle, preset, h0, n):
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset(preset)
    gpu_ctx.upload_reference(c.proteome(), c.fasta_headers())
    img = c.pack(h0, h0 + n, n_threads=4, fasta=True)
    b = gpu_ctx.batch()
    b.set_packed(img.desc, img.chunks, img.payload, img.hap_out_begin)
    b.finalize()
    b.execute()
    b.sync()
    for i in range(0, n, 5):
        hap = c.haplotype(h0 + i)
        t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
        res = coracle.gir_execute_u8(t, c.ref_tape_u32(h0 + i).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8))
        s = res.tobytes().decode()
        want = [(f"{c.tx_name(int(tx))}_{(h0 + i) % 2 + 1}", s[int(a):int(e)]) for tx, a, e in zip(hap.tx_id, hap.tx_res_begin, hap.tx_res_end)]
        assert parse_fasta(b.download_hap(i).tobytes()) == want, (preset, i)
    b.close()
