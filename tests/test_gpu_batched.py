"""Cohorts that do not fit one device image: vcf2prot_amd.driver.run_batched cuts the haplotypes into HBM-sized images and streams
them through v2p_pipeline_*; and BASELINE config 2 (C3: 10 000 samples) whole on ONE MI355X in a single launch."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cut_by_bytes_is_contiguous_and_bounded():
    from vcf2prot_amd.driver import cut_by_bytes
    sizes = [5, 1, 9, 3, 3, 3, 20, 1]
    cuts = cut_by_bytes(sizes, 10)
    assert cuts[0][0] == 0 and cuts[-1][1] == len(sizes) and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    assert all(sum(sizes[a:b]) <= 10 or b - a == 1 for a, b in cuts)


@pytest.mark.parametrize("preset,n,budget", [("C4", 48, 64 << 20), ("C3", 300, 96 << 20), ("C5", 900, 16 << 20)])
def test_batched_driver_every_haplotype_equals_the_oracle(built, gpu_ctx, coracle, preset, n, budget):
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.driver import run_batched
    c = Cohort.preset(preset)
    gpu_ctx.upload_proteome(c.proteome())
    sizes = c.result_sizes(0, n)
    seen, batches = 0, 0
    for br in run_batched(gpu_ctx, lambda a, b: c.pack(a, b, n_threads=8), sizes.tolist(), budget):
        batches += 1
        for h in range(br.h_begin, br.h_end):
            hap = c.haplotype(h)
            t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
            want = coracle.gir_execute_u8(t, c.ref_tape_u32(h).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8))
            assert np.array_equal(br.haplotype(h), want), (preset, h)
            seen += 1
    assert seen == n and batches >= 3


def test_c3_whole_cohort_on_one_gpu_every_haplotype_by_digest(built, gpu_ctx, coracle):
    """BASELINE configs[2]: 10 000 samples = 20 000 haplotypes, about 38 GB of result, one launch on one MI355X (288 GB HBM);
    the digest of EVERY haplotype equals the digest of the oracle's result -- for the image the PRODUCT builds (on the device, from
    the per-transcript Task vectors: v2p_batch_build_on_device, kernel 6 -- haplotype_instruction.rs:94-133 as kernels) and for the
    host-packed image of the same haplotypes."""
    from concurrent.futures import ThreadPoolExecutor
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.txstream import build_on_device_auto
    c = Cohort.preset("C3")
    n = c.n_haplotypes
    assert n == 20000
    threads = min(64, os.cpu_count() or 1)
    gpu_ctx.upload_proteome(c.proteome())
    # the shipped path: stream -> device-built rows image -> execute
    stream = c.txstream(0, n, n_threads=threads)
    sizes = c.result_sizes(0, n, n_threads=threads)
    db = gpu_ctx.batch()
    info = build_on_device_auto(db, stream, int(sizes.sum()))
    stream.close()
    assert info["kernel"] == 6, info
    cn = db.counts()
    assert cn["out_bytes"] == int(sizes.sum()) > 35 * 10 ** 9
    _, _, hb = db.download_image()
    assert np.array_equal(np.diff(hb.astype(np.int64)), sizes.astype(np.int64))
    db.execute()
    db.sync()
    dig_dev = db.digests()
    db.scribble()
    db.execute()                                             # executed again: descriptors staged by the read-ahead, 44 MB phases (round 5)
    db.sync()
    assert np.array_equal(db.digests(), dig_dev) and db.image_form()["staging_buffers"]
    db.close()
    # ... and the ONE call of round 5 (v2p_stream_upload + v2p_batch_build_and_execute): the image stays PADDED, its first execute reads
    # staged descriptors; the first re-execute makes it dense -- every form must leave the arena the two-call builder's image leaves
    stream = c.txstream(0, n, n_threads=threads)
    rs = gpu_ctx.upload_stream(stream)
    stream.close()
    ob = gpu_ctx.batch()
    ob.build_and_execute(rs, 0, 0)
    ob.sync()
    form = ob.image_form()
    assert form["padded"] and form["staging_buffers"], form
    assert np.array_equal(ob.digests(), dig_dev), "the one call's padded, staged image"
    ob.scribble()
    ob.execute()
    ob.sync()
    assert not ob.image_form()["padded"] and np.array_equal(ob.digests(), dig_dev), "the one call's image, made dense and executed again"
    _, _, hb1 = ob.download_image()
    assert np.array_equal(hb1, hb) and ob.counts() == cn
    ob.close()
    rs.close()
    # the host packer's image
    img = c.pack(0, n, n_threads=threads)
    assert img.out_bytes > 35 * 10 ** 9
    b = gpu_ctx.batch()
    b.set_packed(img.desc, img.chunks, img.payload, img.hap_out_begin)
    del img
    b.finalize()
    b.execute()
    b.sync()
    dig = b.digests()
    b.close()
    workers = min(64, os.cpu_count() or 1)

    def oracle_digests(w):
        cc = Cohort.preset("C3")
        out = {}
        for h in range(w, n, workers):
            hap = cc.haplotype(h)
            t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
            want = coracle.gir_execute_u8(t, cc.ref_tape_u32(h).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8))
            out[h] = coracle.digest_u8(want)
        return out
    want = {}
    with ThreadPoolExecutor(workers) as pool:
        for part in pool.map(oracle_digests, range(workers)):
            want.update(part)
    bad = [h for h in range(n) if int(dig_dev[h]) != want[h]]
    assert not bad, ("device-built image", bad[:10])
    bad = [h for h in range(n) if int(dig[h]) != want[h]]
    assert not bad, ("host-packed image", bad[:10])


def test_c2_whole_cohort_through_the_one_call_and_the_stream_pipeline(built, gpu_ctx, coracle):
    """BASELINE configs[1]: 1 000 samples x 20 000 transcripts, SNV-only Tasks -- 2 000 haplotypes, 16 GB of result -- WHOLE, through the
    product's one call (v2p_stream_upload + v2p_batch_build_and_execute) and, slice by slice with the results returning to the host,
    through the stream-fed pipeline (v2p_pipeline_submit_stream): the digest of EVERY haplotype is the oracle's, and what comes back over
    the link digests the same on the host."""
    from concurrent.futures import ThreadPoolExecutor
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.driver import run_streamed
    c = Cohort.preset("C2")
    n = c.n_haplotypes
    assert n == 2000
    threads = min(64, os.cpu_count() or 1)
    gpu_ctx.upload_proteome(c.proteome())
    sizes = c.result_sizes(0, n, n_threads=threads)
    assert int(sizes.sum()) > 15 * 10 ** 9
    stream = c.txstream(0, n, n_threads=threads)
    rs = gpu_ctx.upload_stream(stream)
    stream.close()
    b = gpu_ctx.batch()
    b.build_and_execute(rs, 0, 0)
    b.sync()
    assert b.oneshot_info()["kernel"] == 6
    dig = b.digests()
    b.scribble(); b.execute(); b.sync()
    assert np.array_equal(b.digests(), dig), "the re-executed image left another arena"
    b.close(); rs.close()
    workers = threads

    def oracle_digests(w):
        cc = Cohort.preset("C2")
        out = {}
        for h in range(w, n, workers):
            hap = cc.haplotype(h)
            t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
            want = coracle.gir_execute_u8(t, cc.ref_tape_u32(h).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8))
            out[h] = coracle.digest_u8(want)
        return out
    want = {}
    with ThreadPoolExecutor(workers) as pool:
        for part in pool.map(oracle_digests, range(workers)):
            want.update(part)
    bad = [h for h in range(n) if int(dig[h]) != want[h]]
    assert not bad, ("the one call", bad[:10])
    seen = 0
    for r in run_streamed(gpu_ctx, lambda a, e: c.txstream(a, e, n_threads=threads), sizes, 1 << 30, slots=4, digests=True, copy_threads=8):
        for h in range(r.h_begin, r.h_end):
            assert int(r.digests[h - r.h_begin]) == want[h], ("stream pipeline", h)
        for h in (r.h_begin, r.h_end - 1):                  # the bytes in host memory
            assert coracle.digest_u8(np.ascontiguousarray(r.haplotype(h))) == want[h], ("stream pipeline, host bytes", h)
        seen += r.h_end - r.h_begin
    assert seen == n
