"""The stream-fed pipeline (round 6): Task vectors in, host bytes out, nothing packed on the host -- v2p_pipeline_submit_stream.
What the reference's driver does per sample (parts/exec.rs:23-42: get_g_rep(..).execute(engine), personalized_genome.rs:61-69, the bytes
to the writer :90-113) for a slice of the cohort at a time, H2D / build + execute / D2H of successive slices overlapping.  Every slice's
host bytes must be the oracle's tapes, its digests the oracle's digests."""
import threading

import numpy as np
import pytest

from stream_util import random_stream
from test_gpu_oneshot import oracle_hap

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("preset,h0,n,budget,kernel", [("C3", 40, 600, 8 << 20, 0), ("C5", 100, 3000, 2 << 20, 0), ("C2", 0, 48, 24 << 20, 0),
                                                        ("C4", 3, 16, 16 << 20, 0), ("C1", 0, 8, 1 << 10, 0), ("C3", 0, 300, 6 << 20, 7)])
def test_cohort_slices_through_the_stream_pipeline(built, gpu_ctx, coracle, preset, h0, n, budget, kernel):
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.driver import run_streamed
    c = Cohort.preset(preset)
    gpu_ctx.upload_proteome(c.proteome())
    sizes = c.result_sizes(h0, h0 + n)
    seen, n_slices = 0, 0
    for r in run_streamed(gpu_ctx, lambda a, b: c.txstream(a, b, n_threads=4), sizes, budget, h0=h0, slots=3, kernel=kernel, digests=True, copy_threads=4):
        assert r.h_begin == h0 + seen
        assert np.array_equal(np.diff(r.hap_out_begin.astype(np.int64)), sizes[seen:seen + (r.h_end - r.h_begin)].astype(np.int64))
        assert r.out.size == int(r.hap_out_begin[-1])
        step = max(1, (r.h_end - r.h_begin) // 6)
        for h in range(r.h_begin, r.h_end, step):
            want = oracle_hap(c, coracle, h)
            assert np.array_equal(r.haplotype(h), want), (preset, h)
            assert int(r.digests[h - r.h_begin]) == coracle.digest_u8(want), (preset, h)
        # every haplotype: the device digest against a digest of the HOST bytes (what crossed the link is what the kernel wrote)
        for h in range(r.h_begin, r.h_end):
            assert int(r.digests[h - r.h_begin]) == coracle.digest_u8(np.ascontiguousarray(r.haplotype(h))), (preset, h, "host bytes")
        seen += r.h_end - r.h_begin
        n_slices += 1
    assert seen == n and n_slices >= (1 if preset == "C1" else 3)


@pytest.mark.parametrize("seed,shape,fasta", [(3, "mix", False), (8, "snv", True), (21, "long", True), (22, "mix", True)])
def test_random_streams_through_the_stream_pipeline(built, gpu_ctx, seed, shape, fasta):
    """Irregular slices (empty haplotypes, transcripts without Tasks, gaps, payloads longer than a chunk), with and without FASTA emit;
    slots are reused round-robin, slices differ in size: the slots' buffers grow and are recycled."""
    from vcf2prot_amd.engine import Pipeline
    from vcf2prot_amd._native import V2PError
    rng = np.random.default_rng(seed)
    slices = []
    protein = None
    if fasta:
        proteome, headers, stream, want = random_stream(rng, n_haps=500, n_ref_tx=30, shape=shape, window=4096, fasta=True)
        gpu_ctx.upload_reference(proteome, headers)
    else:
        proteome, stream, want = random_stream(rng, n_haps=500, n_ref_tx=30, shape=shape, window=4096)
        gpu_ctx.upload_proteome(proteome)
    # cut the one stream into slices of haplotypes (views of its arrays)
    from stream_util import Stream
    k = stream.keep
    hb = k[0].astype(np.int64)
    cuts = [0, 7, 8, 130, 131, 300, 500]
    for a, b in zip(cuts[:-1], cuts[1:]):
        t0, t1 = int(hb[a]), int(hb[b])
        tb, ab = k[4].astype(np.int64), k[5].astype(np.int64)
        k0, k1, a0, a1 = int(tb[t0]), int(tb[t1]), int(ab[t0]), int(ab[t1])
        slices.append((a, b, Stream(hb[a:b + 1] - t0, k[1][t0:t1], k[2][t0:t1], k[3][t0:t1], tb[t0:t1 + 1] - k0, ab[t0:t1 + 1] - a0,
                                    k[6][k0:k1], k[7][k0:k1], k[8][k0:k1], k[9][k0:k1], k[10][a0:a1],
                                    k[11][t0:t1] if fasta else None, k[12][t0:t1] if fasta else None)))
    pipe = Pipeline(gpu_ctx, 2)
    try:
        inflight = []
        for a, b, st in slices:
            if len(inflight) == 2:
                _check(pipe, inflight.pop(0), want)
            inflight.append((pipe.submit_stream(st, 0, True), a, b))
        while inflight:
            _check(pipe, inflight.pop(0), want)
        # a slice the reference would panic on: reported by wait, the slot lives on
        a, b, st = slices[3]
        bad = Stream(*([x.copy() for x in st.keep[:6]] + [x[:-64].copy() for x in st.keep[6:11]])) if not fasta else None      # (Stream pads the Task arrays by 64)
        if bad is not None and bad.struct.n_tasks:
            bad.keep[6][0] = 7                              # exe_code 7: haplotype_instruction.rs:154
            t = pipe.submit_stream(bad, 0, False)
            with pytest.raises(V2PError) as e:
                pipe.wait(t)
            assert e.value.code == -3 and e.value.index == 0
            pipe.release(t)
            _check(pipe, (pipe.submit_stream(st, 0, True), a, b), want)
    finally:
        pipe.close()
    if fasta:
        gpu_ctx.upload_proteome(proteome)


def _check(pipe, job, want):
    t, a, b = job
    out = pipe.wait(t)
    info = pipe.result_info(t)
    hob = info["hap_out_begin"]
    assert hob.size == b - a + 1
    for h in range(a, b):
        got = out[int(hob[h - a]):int(hob[h - a + 1])]
        assert got.size == want[h].size and np.array_equal(got, want[h]), h
    pipe.release(t)


def test_submissions_from_several_threads(built, gpu_ctx, coracle):
    """Each worker prepares and submits its own slices (the shape of a Rayon pool: parts/exec.rs:36-39); staging runs on the callers'
    threads, concurrently; tickets identify the slots."""
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.engine import Pipeline
    c = Cohort.preset("C3")
    gpu_ctx.upload_proteome(c.proteome())
    pipe = Pipeline(gpu_ctx, 4)
    pipe.reserve(8 << 20, 16 << 20, 2)
    errors, lock = [], threading.Lock()
    sem = threading.Semaphore(4)                            # at most as many claimed slots as the pipeline has

    def worker(w):
        try:
            cc = Cohort.preset("C3")
            for j in range(3):
                h0 = 50 * (3 * w + j)
                st = cc.txstream(h0, h0 + 50, n_threads=1)
                with sem:
                    t = pipe.submit_stream(st, 0, True)
                    st.close()
                    out = pipe.wait(t)
                    info = pipe.result_info(t)
                    hob = info["hap_out_begin"]
                    for i in (0, 17, 49):
                        want = oracle_hap(cc, coracle, h0 + i)
                        assert np.array_equal(out[int(hob[i]):int(hob[i + 1])], want), (w, j, i)
                        assert int(info["digests"][i]) == coracle.digest_u8(want)
                    pipe.release(t)
        except Exception as e:                              # noqa: BLE001
            with lock:
                errors.append((w, repr(e)))
    ts = [threading.Thread(target=worker, args=(w,)) for w in range(6)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    pipe.close()
    assert not errors, errors


def test_slots_tables_and_busy(built, gpu_ctx):
    """Every slot in use: V2P_BUSY, nothing staged (Python: -1).  A slice with a broken offset table: refused at submit with the offending index
    (as v2p_stream_upload refuses it), its slot free again.  Results released out of order: the next submission takes the free slot."""
    from stream_util import Stream
    from vcf2prot_amd.engine import Pipeline
    from vcf2prot_amd._native import V2PError
    rng = np.random.default_rng(17)
    proteome, stream, want = random_stream(rng, n_haps=40, n_ref_tx=10, shape="mix", window=4096)
    gpu_ctx.upload_proteome(proteome)
    pipe = Pipeline(gpu_ctx, 2)
    try:
        t0 = pipe.submit_stream(stream, 0, False)
        t1 = pipe.submit_stream(stream, 0, False)
        assert {t0, t1} == {0, 1}
        assert pipe.submit_stream(stream, 0, False) == -1              # V2P_BUSY
        k = stream.keep
        arrays = [x.copy() for x in k[:6]] + [x[:-64].copy() for x in k[6:11]]
        arrays[4][3] = arrays[4][2] - 1 if arrays[4][2] else 10 ** 9    # tx_task_begin not ascending at transcript 2
        bad = Stream(*arrays)
        out1 = pipe.wait(t1)                                            # out of order: the second first
        hob = pipe.result_info(t1)["hap_out_begin"]
        assert np.array_equal(out1[int(hob[5]):int(hob[6])], want[5])
        pipe.release(t1)
        with pytest.raises(V2PError) as e:
            pipe.submit_stream(bad, 0, False)
        assert e.value.code == -1 and e.value.index == 2, (e.value.code, e.value.index)
        t2 = pipe.submit_stream(stream, 0, True)                        # the refused submission left its slot free
        assert t2 == t1
        for t in (t0, t2):
            out = pipe.wait(t)
            hob = pipe.result_info(t)["hap_out_begin"]
            for h in (0, 17, 39):
                assert np.array_equal(out[int(hob[h]):int(hob[h + 1])], want[h])
            pipe.release(t)
    finally:
        pipe.close()
