"""PIECE images (csrc/dense_pieces.h): a dense rows image that is executed AGAIN is re-written, once, as pieces of <= 16 result bytes with
their positions and substituted residues, and stitch_pieces_kernel executes those.  Bit-exact against the dense kernel's arena (the first
execute), the oracle and the reference's own Task dumps; an image the form does not take stays on the dense kernel."""
import numpy as np
import pytest

from stream_util import random_stream                          # noqa: E402
from test_gpu_oneshot import oracle_hap                        # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("preset,h0,n", [("C5", 50, 1500), ("C5", 11, 300), ("C3", 100, 300), ("C1", 0, 8), ("C4", 7, 12), ("C2", 0, 40)])
def test_a_dense_image_executed_again_runs_from_pieces(built, gpu_ctx, coracle, preset, h0, n):
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset(preset)
    gpu_ctx.upload_proteome(c.proteome())
    stream = c.txstream(h0, h0 + n, n_threads=4)
    rs = gpu_ctx.upload_stream(stream)
    stream.close()
    b = gpu_ctx.batch()
    b.build_and_execute(rs, 7, 0)                           # a dense rows image whatever the rule says; first execute: stitch_dense_kernel
    b.sync()
    assert not b.image_form()["pieces"]
    d1 = b.digests()
    for i in range(0, n, max(1, n // 20)):
        assert np.array_equal(b.download_hap(i), oracle_hap(c, coracle, h0 + i)), (preset, h0 + i)
    b.scribble(); b.execute(); b.sync()                                   # executed again: pieces
    assert b.image_form()["pieces"]
    assert np.array_equal(b.digests(), d1)
    for i in range(0, n, max(1, n // 20)):
        assert np.array_equal(b.download_hap(i), oracle_hap(c, coracle, h0 + i)), (preset, h0 + i)
    b.scribble(); b.execute(); b.sync()
    assert np.array_equal(b.digests(), d1)
    # the two-call form: the first v2p_batch_execute is the image's first execute (dense kernel), the second runs from pieces
    b2 = gpu_ctx.batch()
    b2.build_from_stream(rs, 7)
    b2.scribble(); b2.execute(); b2.sync()
    assert not b2.image_form()["pieces"] and np.array_equal(b2.digests(), d1)
    b2.scribble(); b2.execute(); b2.sync()
    assert b2.image_form()["pieces"] and np.array_equal(b2.digests(), d1)
    b2.close(); b.close(); rs.close()


@pytest.mark.parametrize("seed,shape", [(1, "snv"), (3, "snv"), (5, "mix"), (6, "mix"), (10, "long"), (12, "long")])
def test_random_streams_from_pieces(built, gpu_ctx, seed, shape):
    """Irregular streams (empty haplotypes, transcripts without Tasks, gaps, long payloads, immediates, fused pairs)."""
    rng = np.random.default_rng(seed)
    proteome, stream, want = random_stream(rng, n_haps=400, n_ref_tx=25, shape=shape, window=4096)
    gpu_ctx.upload_proteome(proteome)
    rs = gpu_ctx.upload_stream(stream)
    b = gpu_ctx.batch()
    b.build_and_execute(rs, 7, 0)
    b.sync()
    b.scribble(); b.execute(); b.sync()
    assert b.image_form()["pieces"]
    for h, w in enumerate(want):
        got = b.download_hap(h)
        assert got.size == w.size and np.array_equal(got, w), (seed, shape, h)
    b.close()
    rs.close()
