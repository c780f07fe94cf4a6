"""stitchw_kernel (one wave per chunk of <= 64 descriptors / <= 10 KiB, a fused substitution = one record): the same bytes as the
oracle (task.rs:38-50 restated) for the same Task vectors, through the C ABI, whatever the chunking."""
import numpy as np
import pytest

from test_gpu_kernels import oracle_haps, run_image

pytestmark = pytest.mark.gpu


def _is_wave(img):
    return (((img.chunks[:, 1] >> np.uint64(60)) & np.uint64(1)) == 1).all() and (img.launch_bits & 4) and (img.launch_bits & 48) == 48


@pytest.mark.parametrize("preset,h0,n", [("C1", 0, 8), ("C2", 3, 3), ("C3", 11, 6), ("C4", 2, 2), ("C5", 9, 30)])
def test_wave_kernel_equals_the_oracle(built, gpu_ctx, coracle, preset, h0, n):
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset(preset)
    gpu_ctx.upload_proteome(c.proteome())
    want = oracle_haps(c, coracle, h0, n)
    for threads in (1, 3):                   # (parts packed side by side: cuts aligned in the arena, not in the part)
        img = c.pack(h0, h0 + n, n_threads=threads, kernel=4)
        assert _is_wave(img)
        nd = (img.chunks[:, 1] >> np.uint64(48)) & np.uint64(0x7FF)
        assert int(nd.max()) <= 64
        if preset != "C5":
            assert ((img.desc >> np.uint64(61)) == 7).any()                 # fused substitutions are in use
        got = run_image(gpu_ctx, img)
        for i in range(n):
            assert np.array_equal(got[i], want[i]), (preset, threads, h0 + i)


@pytest.mark.parametrize("preset,h0,n", [("C1", 0, 8), ("C3", 40, 5), ("C2", 1, 2), ("C5", 7, 20)])
@pytest.mark.parametrize("chunk_tasks,chunk_bytes,cut_align,soft_window", [
    (3, 64, 16, 0), (17, 1000, 16, 2), (64, 4096, 64, 8), (64, 8192, 1024, 8), (64, 300, 16, 0), (5, 8192, 4096, 1), (64, 8192, 16, 0), (33, 2049, 16, 3)])
def test_wave_kernel_any_chunking_gives_the_same_bytes(built, gpu_ctx, coracle, preset, h0, n, chunk_tasks, chunk_bytes, cut_align, soft_window):
    """Ragged cuts (descriptor limit at unaligned offsets), cuts that split a fused substitution, tiny chunks, chunks of every row count."""
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset(preset)
    gpu_ctx.upload_proteome(c.proteome())
    want = oracle_haps(c, coracle, h0, n)
    img = c.pack(h0, h0 + n, n_threads=2, kernel=4, chunk_tasks=chunk_tasks, chunk_bytes=chunk_bytes, cut_align=cut_align, soft_window=soft_window)
    assert _is_wave(img)
    got = run_image(gpu_ctx, img)
    for i in range(n):
        assert np.array_equal(got[i], want[i]), (preset, i)


def test_wave_kernel_with_fasta_records(built, gpu_ctx, coracle):
    """FASTA emit (headers and line feeds as descriptors between the tasks) through the wave kernel: the arena equals the other kernels'."""
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset("C3")
    gpu_ctx.upload_reference(c.proteome(), c.fasta_headers())
    a = c.pack(5, 9, n_threads=2, kernel=2, fasta=True)
    w = c.pack(5, 9, n_threads=2, kernel=4, fasta=True)
    assert _is_wave(w) and np.array_equal(a.hap_out_begin, w.hap_out_begin)
    ga, gw = run_image(gpu_ctx, a), run_image(gpu_ctx, w)
    for i in range(4):
        assert ga[i].size and np.array_equal(ga[i], gw[i]), i
        assert bytes(gw[i][:5]) == b">ENST"


def test_wave_kernel_refuses_bad_images(built, gpu_ctx):
    """A descriptor that would read outside its source, a chunk with more than 64 descriptors or more than 10 KiB of result, a chunk
    table that points outside the descriptor array: reported (the reference would panic, task.rs:43,47), never executed."""
    from vcf2prot_amd.cohort import Cohort
    from vcf2prot_amd.engine import V2PError
    c = Cohort.preset("C2", n_transcripts=200)
    prot = c.proteome()
    gpu_ctx.upload_proteome(prot)
    img = c.pack(0, 2, n_threads=1, kernel=4)

    def run(desc, chunks):
        b = gpu_ctx.batch()
        try:
            b.set_packed(desc, chunks, img.payload, img.hap_out_begin)
            b.finalize()
            b.execute()
            b.sync()
        finally:
            b.close()

    run(img.desc, img.chunks)                                              # the image itself is fine
    bad = img.desc.copy()
    k = int(np.nonzero((bad >> np.uint64(62)) == 0)[0][3])                 # a plain proteome copy: push its source past the end
    bad[k] = (bad[k] & ~np.uint64((1 << 40) - 1)) | np.uint64(prot.size - 1)
    with pytest.raises(V2PError):
        run(bad, img.chunks)
    ch = img.chunks.copy()                                                 # 65 descriptors in one wave chunk
    ch[0, 1] = (ch[0, 1] & ~(np.uint64(0x7FF) << np.uint64(48))) | (np.uint64(65) << np.uint64(48))
    with pytest.raises(V2PError):
        run(img.desc, ch)
    ch = img.chunks.copy()                                                 # a long-run image's 32 KiB chunk flagged for the wave kernel
    big = c.pack(0, 2, n_threads=1, kernel=1)
    chb = big.chunks.copy()
    chb[:, 1] = (chb[:, 1] & ~(np.uint64(0xF) << np.uint64(60))) | (np.uint64(1) << np.uint64(60))
    with pytest.raises(V2PError):
        run(big.desc, chb)


@pytest.mark.parametrize("preset,h0,n,kernel,phase_bytes", [("C3", 20, 12, 4, 40_000), ("C3", 20, 12, 4, 7_000), ("C2", 1, 3, 4, 20_000), ("C4", 3, 3, 4, 100_000),
                                                            ("C2", 1, 3, 1, 30_000), ("C1", 0, 8, 4, 600)])
def test_phased_launches_on_small_images(built, gpu_ctx, coracle, preset, h0, n, kernel, phase_bytes):
    """launch_stitch cuts wave and long-run images into phases (a read-ahead of the phase's chunk records, descriptors and payload
    lines, riding on the previous phase's trailing workgroups for a pure wave image); only images of >= 16 384 chunks are phased, so
    here the threshold is lowered and the phases made tiny: dozens of phases, a ragged last one, read-ahead sets smaller than a wave
    -- every haplotype still the oracle's, with and without the read-ahead riding."""
    from test_gpu_kernels import oracle_haps, run_image
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset(preset)
    gpu_ctx.upload_proteome(c.proteome())
    want = oracle_haps(c, coracle, h0, n)
    img = c.pack(h0, h0 + n, n_threads=2, kernel=kernel)
    gpu_ctx.set_launch_opts(phase_bytes=phase_bytes, phase_min_chunks=1)
    try:
        per_chunk = 16 + 8 * img.desc.size / max(img.chunks.shape[0], 1)
        assert img.chunks.shape[0] > 2 * max(8, phase_bytes / per_chunk) or preset == "C1"      # really several phases
        got = run_image(gpu_ctx, img)
        for i in range(n):
            assert np.array_equal(got[i], want[i]), (preset, kernel, phase_bytes, i)
        gpu_ctx.set_launch_opts(phase_min_chunks=1)
        got = run_image(gpu_ctx, img)                                     # default phase size, threshold still lowered
        for i in range(n):
            assert np.array_equal(got[i], want[i]), (preset, kernel, "default", i)
    finally:
        gpu_ctx.set_launch_opts()
