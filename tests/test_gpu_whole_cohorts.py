"""(round 5: every whole cohort also through the DEVICE builder -- the path the product ships.)
The 8-GPU configurations of BASELINE.json WHOLE on one MI355X (both fit its 288 GB): C4 (2 504 samples = 5 008 haplotypes over the
100 000-transcript proteome, ~30 GB of result) and C5 (100 000 deep haplotypes, ~8 GB, 1.3e9 Tasks) in ONE launch, the digest of
every haplotype against the oracle; then the same cohort the way eight ranks would run it -- shard_by_bytes ranges executed one
after the other -- must reproduce the single launch: per-haplotype digests, and the byte offsets the size all-gather derives
(shard.layout_from_sizes) must be the single arena's own haplotype offsets."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_digests(preset, coracle, n, workers):
    from vcf2prot_amd.cohort import Cohort

    def work(w):
        cc = Cohort.preset(preset)                     # own generator state per thread
        out = {}
        for h in range(w, n, workers):
            hap = cc.haplotype(h)
            t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
            want = coracle.gir_execute_u8(t, cc.ref_tape_u32(h).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8))
            out[h] = coracle.digest_u8(want)
        return out
    res = {}
    with ThreadPoolExecutor(workers) as pool:
        for part in pool.map(work, range(workers)):
            res.update(part)
    return np.array([res[h] for h in range(n)], dtype=np.uint64)


def _run_device_built(gpu_ctx, c, n, threads, kernel):
    """The image the product ships: built ON the device from the transcript stream (v2p_batch_build_on_device, rows images)."""
    stream = c.txstream(0, n, n_threads=threads)
    b = gpu_ctx.batch()
    ms = b.build_on_device(stream, 0, kernel)
    stream.close()
    assert ms > 0
    _, _, hb = b.download_image()
    b.execute()
    b.sync()
    dig = np.array(b.digests(), dtype=np.uint64)
    cn = b.counts()
    # ... and executed AGAIN, in the image's re-execution form (round 5): a dense rows image from its pieces (csrc/dense_pieces.h), a rich
    # wave image with its descriptors staged by the read-ahead -- the same arena, written again (scribbled in between)
    b.scribble()
    b.execute()
    b.sync()
    assert np.array_equal(np.array(b.digests(), dtype=np.uint64), dig), "the re-executed image left another arena"
    assert b.image_form()["pieces"] == (kernel == 7)
    b.close()
    return dig, hb.astype(np.int64), cn


def _run(gpu_ctx, img):
    b = gpu_ctx.batch()
    b.set_packed(img.desc, img.chunks, img.payload, img.hap_out_begin)
    b.finalize()
    b.execute()
    b.sync()
    dig = np.array(b.digests(), dtype=np.uint64)
    b.close()
    return dig


@pytest.mark.parametrize("preset,n_expected,min_bytes", [("C4", 5008, 25 * 10 ** 9), ("C5", 100000, 7 * 10 ** 9)])
def test_whole_cohort_one_launch_and_as_eight_shards(built, gpu_ctx, coracle, preset, n_expected, min_bytes):
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.shard import layout_from_sizes, shard_by_bytes
    threads = min(64, os.cpu_count() or 1)
    c = Cohort.preset(preset)
    n = c.n_haplotypes
    assert n == n_expected
    gpu_ctx.upload_proteome(c.proteome())
    img = c.pack(0, n, n_threads=threads)
    assert img.out_bytes > min_bytes
    whole_begin = img.hap_out_begin.astype(np.int64)
    whole = _run(gpu_ctx, img)
    del img
    want = _oracle_digests(preset, coracle, n, min(32, os.cpu_count() or 1))
    bad = np.nonzero(whole != want)[0]
    assert bad.size == 0, (preset, bad[:10])
    # ... and the device-built image of the whole cohort (C4: kernel 6, wave rows image over the 56 MB proteome, 30 GB of arena; C5:
    # kernel 7, dense rows image, 1.29e9 Tasks): haplotype offsets, every digest against the oracle's and the host-packed image's
    dev, dev_begin, cn = _run_device_built(gpu_ctx, c, n, threads, 6 if preset == "C4" else 7)
    assert np.array_equal(dev_begin, whole_begin)
    assert cn["out_bytes"] == int(whole_begin[-1])
    bad = np.nonzero(dev != want)[0]
    assert bad.size == 0, (preset, "device-built", bad[:10])
    # the same cohort as world = 8 would run it: contiguous ranges of equal result bytes, every rank its own image and arena
    sizes = c.result_sizes(0, n, n_threads=threads)
    assert np.array_equal(np.diff(whole_begin), sizes.astype(np.int64))
    ranges = shard_by_bytes(sizes.tolist(), 8)
    assert ranges[0][0] == 0 and ranges[-1][1] == n and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    n_haps, out_bytes = [], []
    for r, (h0, h1) in enumerate(ranges):
        part = c.pack(h0, h1, n_threads=threads)
        assert int(part.hap_out_begin[0]) == 0                       # a rank's arena is its own
        dig = _run(gpu_ctx, part)
        assert np.array_equal(dig, whole[h0:h1]), (preset, r)
        n_haps.append(h1 - h0)
        out_bytes.append(part.out_bytes)
        del part
    share = np.array(out_bytes, dtype=np.float64) / sum(out_bytes)
    assert share.max() - share.min() < 0.01                          # balanced by bytes
    for r, (h0, h1) in enumerate(ranges):
        lay = layout_from_sizes(r, n_haps, out_bytes)                 # what the RCCL all-gather of 16 bytes per rank delivers
        assert lay.hap_offset == h0 and lay.byte_offset == int(whole_begin[h0]) and lay.total_bytes == int(whole_begin[-1])
