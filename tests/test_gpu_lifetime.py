"""Lifetime and bounds hygiene at the ABI (round 6).

* A batch built from a resident v2p_stream reads the stream's alt bytes: v2p_stream_destroy orphans it, and executing an orphan is
  V2P_ERR_STATE -- never a read of freed memory that returns V2P_OK.  (The reference cannot dangle: GIR::execute(self) owns its tapes by
  move, gir.rs:197,230-234.)
* The cutter's padded chunk table: a stream between 24 and ~100 result bytes per Task cuts a 640-row segment into up to 640 chunks.  Until
  round 5 the table had 96 slots per segment whatever the stream, and the table pass behind an overflowing cut wrote past the scratch the
  one call had sized for it.  The table is now sized from the stream, and where the estimate is too low -- a stream
  whose dense stretch hides in a thin average -- nothing is written and the call builds in one piece."""
import numpy as np
import pytest

from stream_util import random_stream, regular_stream

pytestmark = pytest.mark.gpu


def test_a_batch_outliving_its_stream_refuses_to_execute(built, gpu_ctx):
    from vcf2prot_amd._native import V2P_ERR_STATE, V2PError
    rng = np.random.default_rng(4)
    proteome, stream, want = random_stream(rng, n_haps=200, n_ref_tx=20, shape="mix", window=4096)
    gpu_ctx.upload_proteome(proteome)
    for how in ("one_call", "two_calls", "dense"):
        rs = gpu_ctx.upload_stream(stream)
        b = gpu_ctx.batch()
        if how == "one_call":
            b.build_and_execute(rs, 0, 0)
        else:
            b.build_from_stream(rs, 7 if how == "dense" else 0)
            b.execute()
        b.sync()
        dig = b.digests()
        rs.close()                                          # the stream goes first: the batch is an orphan
        for _ in range(2):
            with pytest.raises(V2PError) as e:
                b.execute()
            assert e.value.code == V2P_ERR_STATE
        # the arena is the batch's own memory: still there, still right
        b.sync()
        assert np.array_equal(b.digests(), dig)
        for h in range(0, len(want), 17):
            assert np.array_equal(b.download_hap(h), want[h])
        # reset: an ordinary empty batch again, buildable from another stream
        b.reset()
        rs2 = gpu_ctx.upload_stream(stream)
        b.build_and_execute(rs2, 0, 0)
        b.sync()
        assert np.array_equal(b.digests(), dig)
        b.scribble(); b.execute(); b.sync()
        assert np.array_equal(b.digests(), dig)
        b.close()                                           # the batch goes first this time
        rs2.close()


def test_destroying_a_stream_leaves_other_batches_alone(built, gpu_ctx):
    rng = np.random.default_rng(9)
    proteome, stream, want = random_stream(rng, n_haps=150, n_ref_tx=20, shape="snv", window=4096)
    gpu_ctx.upload_proteome(proteome)
    rs_a, rs_b = gpu_ctx.upload_stream(stream), gpu_ctx.upload_stream(stream)
    a, b = gpu_ctx.batch(), gpu_ctx.batch()
    a.build_and_execute(rs_a, 0, 0); b.build_and_execute(rs_b, 0, 0)
    a.sync(); b.sync()
    dig = a.digests()
    rs_a.close()
    b.scribble(); b.execute(); b.sync()                    # b's stream is alive
    assert np.array_equal(b.digests(), dig)
    a.close(); b.close(); rs_b.close()


def test_oneshot_info_after_reset_and_another_builder(built, gpu_ctx):
    """reset + a two-call build left the previous one call's timings readable."""
    from vcf2prot_amd._native import V2P_ERR_STATE, V2PError
    rng = np.random.default_rng(2)
    proteome, stream, _ = random_stream(rng, n_haps=60, n_ref_tx=10, shape="mix", window=4096)
    gpu_ctx.upload_proteome(proteome)
    rs = gpu_ctx.upload_stream(stream)
    b = gpu_ctx.batch()
    b.build_and_execute(rs, 0, 0); b.sync()
    assert b.oneshot_info()["kernel"] in (6, 7, 9)
    b.reset()
    b.build_from_stream(rs, 0)
    with pytest.raises(V2PError) as e:
        b.oneshot_info()
    assert e.value.code == V2P_ERR_STATE
    b.close(); rs.close()


@pytest.mark.parametrize("dense_every,expect_one_pass", [(1, True), (8, False)])
def test_segments_of_one_row_chunks(built, gpu_ctx, dense_every, expect_one_pass):
    """More than 130 cutter segments whose rows hold ~43 descriptors each: one-row chunks, 640 per segment (kernel 6).  dense_every = 1: the
    whole stream is like that, the table is sized for it and the one call stays on its one-pass path.  dense_every = 8: seven haplotypes of
    eight are single reference copies -- the stream's average (190 bytes per Task) sizes the table for ten-row chunks, the dense
    haplotypes' segments overflow it, the call must notice (totals[3]) and build in one piece; nothing may be written out of bounds
    meanwhile: a neighbouring batch's image and arena are checked to be intact."""
    proteome, stream, want = regular_stream(n_haps=176 if dense_every == 1 else 1408, tx_per_hap=500, dense_every=dense_every, seed=3)
    gpu_ctx.upload_proteome(proteome)
    # the canaries: batches (over the same resident proteome) whose allocations sit around the call's scratch in the device heap
    rs = gpu_ctx.upload_stream(stream)
    n_haps = rs.counts()["n_haps"]
    out_bytes = rs.counts()["out_bytes"]
    assert out_bytes >= 130 * 640 * 1024
    canary = []
    for _ in range(3):
        crs = gpu_ctx.upload_stream(regular_stream(n_haps=6, tx_per_hap=50, seed=11)[1])
        cb = gpu_ctx.batch()
        cb.build_and_execute(crs, 0, 0); cb.sync()
        canary.append((cb, crs, cb.digests(), cb.download_image()))
    b = gpu_ctx.batch()
    for rep in range(2):                                    # (the second round recycles the first one's buffers)
        b.build_and_execute(rs, 6, 0)
        b.sync()
        info = b.oneshot_info()
        assert info["kernel"] == 6
        assert (info["n_slices"] >= 1) == expect_one_pass, info     # n_slices == 0: the call fell back to the one-piece builder
        for h in list(range(0, n_haps, 37)) + [n_haps - 1]:
            assert np.array_equal(b.download_hap(h), want(h)), (dense_every, rep, h)
        b.scribble(); b.execute(); b.sync()
        for h in list(range(5, n_haps, 41)):
            assert np.array_equal(b.download_hap(h), want(h)), (dense_every, rep, h, "re-executed")
        b.reset()
    for cb, crs, dig, img in canary:
        assert np.array_equal(cb.digests(), dig)
        again = cb.download_image()
        assert all(np.array_equal(x, y) for x, y in zip(again, img))
        cb.scribble(); cb.execute(); cb.sync()
        assert np.array_equal(cb.digests(), dig)
        cb.close(); crs.close()
    b.close(); rs.close()
