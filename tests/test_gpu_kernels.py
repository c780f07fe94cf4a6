"""The two stitch kernels behind one C ABI (Task::execute, task.rs:38-50): stitch4_kernel (long-run chunks, fused
substitution descriptors, every load before the first store) and the per-block stitch_kernel must write the same bytes as the
oracle for the same Task vectors, alone and mixed in one image."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run_image(gpu_ctx, img):
    b = gpu_ctx.batch()
    b.set_packed(img.desc, img.chunks, img.payload, img.hap_out_begin)
    b.finalize()
    b.execute()
    b.sync()
    out = [b.download_hap(i) for i in range(img.hap_out_begin.size - 1)]
    b.close()
    return out


def oracle_haps(c, coracle, h0, n):
    res = []
    for h in range(h0, h0 + n):
        hap = c.haplotype(h)
        t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
        res.append(coracle.gir_execute_u8(t, c.ref_tape_u32(h).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8)))
    return res


@pytest.mark.parametrize("preset,h0,n", [("C1", 0, 8), ("C2", 3, 2), ("C3", 11, 5), ("C4", 2, 2), ("C5", 9, 30)])
def test_both_kernels_equal_the_oracle(built, gpu_ctx, coracle, preset, h0, n):
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset(preset)
    gpu_ctx.upload_proteome(c.proteome())
    want = oracle_haps(c, coracle, h0, n)
    for kernel in (1, 2, 0):                 # long-run kernel for every chunk, per-block kernel for every chunk, the builder's choice
        img = c.pack(h0, h0 + n, n_threads=2, kernel=kernel)
        long_flags = (img.chunks[:, 1] >> np.uint64(63)).astype(bool)
        if kernel == 2:
            assert not long_flags.any() and not ((img.desc >> np.uint64(61)) == 7).any()
        got = run_image(gpu_ctx, img)
        for i in range(n):
            assert np.array_equal(got[i], want[i]), (preset, kernel, h0 + i)


def test_fused_substitutions_are_used_and_exact(built, gpu_ctx, coracle):
    """C2 (one missense per transcript): reference copy + literal + reference copy -> one descriptor, also for a substitution at
    the first or last residue (an empty copy on one side)."""
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset("C2")
    gpu_ctx.upload_proteome(c.proteome())
    img = c.pack(0, 3, n_threads=1, kernel=1)
    fused = (img.desc >> np.uint64(61)) == 7
    assert fused.sum() > 0.9 * 3 * c.n_transcripts                         # nearly every transcript (not those cut by a chunk boundary)
    len1 = (img.desc[fused] >> np.uint64(29)) & np.uint64(0xFFF)
    len2 = (img.desc[fused] >> np.uint64(41)) & np.uint64(0xFFF)
    assert (len1 == 0).any() and (len2 == 0).any()                          # substitutions at the first / last residue of a transcript
    want = oracle_haps(c, coracle, 0, 3)
    got = run_image(gpu_ctx, img)
    for i in range(3):
        assert np.array_equal(got[i], want[i]), i


def test_mixed_image_runs_both_kernels(built, gpu_ctx, coracle):
    """One image holding long-run chunks (with fused descriptors) and per-block chunks: each kernel takes its own."""
    from vcf2prot_amd.cohort import Cohort, Packed
    c = Cohort.preset("C3")
    gpu_ctx.upload_proteome(c.proteome())
    a = c.pack(0, 4, n_threads=1, kernel=1)
    b = c.pack(4, 8, n_threads=1, kernel=2)
    desc_b = b.desc.copy()
    pay = (desc_b >> np.uint64(62)) == 1                                    # payload-space descriptors move behind a's payload
    desc_b[pay] += np.uint64(a.payload.size)
    chunks_b = b.chunks.copy()
    chunks_b[:, 0] += np.uint64(a.desc.size)
    chunks_b[:, 1] += np.uint64(a.out_bytes)                               # result offsets sit in the low 48 bits
    img = Packed(np.concatenate([a.desc, desc_b]), np.concatenate([a.chunks, chunks_b]), np.concatenate([a.payload, b.payload]),
                 np.concatenate([a.hap_out_begin, b.hap_out_begin[1:] + np.uint64(a.out_bytes)]), a.n_tasks + b.n_tasks,
                 a.n_copy_bytes + b.n_copy_bytes)
    flags = (img.chunks[:, 1] >> np.uint64(63)).astype(bool)
    assert flags.any() and not flags.all()
    assert img.launch_bits & 0x30 == 0                                      # both kernels have work
    want = oracle_haps(c, coracle, 0, 8)
    got = run_image(gpu_ctx, img)
    for i in range(8):
        assert np.array_equal(got[i], want[i]), i


@pytest.mark.parametrize("chunk_bytes", [48, 1000, 32768])
def test_long_run_kernel_handles_ragged_and_split_cuts(built, gpu_ctx, coracle, chunk_bytes):
    """Chunk cuts that are not 16-byte aligned (hard task limit with tiny chunks) and cuts that split a fused triple."""
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset("C2", n_transcripts=300)
    gpu_ctx.upload_proteome(c.proteome())
    want = oracle_haps(c, coracle, 0, 4)
    for chunk_tasks in (4, 7, 256):
        img = c.pack(0, 4, n_threads=1, kernel=1, chunk_tasks=chunk_tasks, chunk_bytes=chunk_bytes, cut_align=16)
        got = run_image(gpu_ctx, img)
        for i in range(4):
            assert np.array_equal(got[i], want[i]), (chunk_tasks, i)


@pytest.mark.parametrize("preset,h0,n", [("C5", 3, 24), ("C3", 7, 3), ("C1", 0, 8)])
def test_dense_kernel_on_any_per_block_image(built, coracle, preset, h0, n):
    """stitch_dense_kernel (lane = task, LDS image, piece list) is picked for images of short tasks; forced (variant 8, and 9 =
    dword-aligned gathers) it must also be exact on images it is not picked for: long tasks go through the piece list, chunks
    larger than the LDS image through several windows, literals may be cut by a window."""
    from hip_util import DevBuf
    from vcf2prot_amd import _native as N
    from vcf2prot_amd.cohort import Cohort
    lib, blib = N.hip_lib(), N.bench_lib()             # (both switches live in libv2p_bench.so: 9 is a kernel variant, 8 a routing switch of its launcher)
    c = Cohort.preset(preset)
    prot = c.proteome()
    want = np.concatenate(oracle_haps(c, coracle, h0, n))
    d_prot = DevBuf.of(prot)
    for pack in (dict(kernel=2), dict(kernel=2, chunk_tasks=1000, chunk_bytes=65520), dict(kernel=2, chunk_tasks=700, chunk_bytes=20000, cut_align=16)):
        img = c.pack(h0, h0 + n, n_threads=2, **pack)
        chunks = np.ascontiguousarray(img.chunks)
        d_desc, d_chunks, d_pay = DevBuf.of(img.desc), DevBuf.of(chunks), DevBuf.of(img.payload)
        bits = int(lib.v2p_stitch_launch_bits(chunks.ctypes.data, chunks.shape[0]))
        for var in (8, 9):
            d_out = DevBuf(img.out_bytes + 32, fill=0x2E)
            d_status = DevBuf.of(np.full(1, -1, dtype=np.int64))
            if var == 8:
                rc = N.stitch_launch(blib, None, d_desc.ptr, img.desc.size, d_chunks.ptr, chunks.shape[0], d_prot.ptr, prot.size, d_pay.ptr, img.payload.size,
                                     d_out.ptr, img.out_bytes, d_status.ptr, N.LaunchOpts(routing=bits, reserved=8))
                # (the product's launcher takes no routing switch: reserved must be 0)
                assert N.stitch_launch(lib, None, d_desc.ptr, img.desc.size, d_chunks.ptr, chunks.shape[0], d_prot.ptr, prot.size, d_pay.ptr, img.payload.size,
                                       d_out.ptr, img.out_bytes, d_status.ptr, N.LaunchOpts(routing=bits, reserved=8)) == -1
            else:
                rc = blib.v2p_stitch_launch(None, d_desc.ptr, img.desc.size, d_chunks.ptr, chunks.shape[0], d_prot.ptr, prot.size, d_pay.ptr, img.payload.size,
                                            d_out.ptr, img.out_bytes, d_status.ptr, 1 | bits | (var << 12), 0)
            assert rc == 0
            assert d_status.download().view(np.int64)[0] == -1
            got = d_out.download()
            hb = img.hap_out_begin.astype(np.int64)
            res = np.concatenate([got[hb[i]:hb[i] + c.haplotype(h0 + i).n_res] for i in range(n)])
            assert np.array_equal(res, want), (preset, pack, var)
            d_out.free(); d_status.free()
        for b in (d_desc, d_chunks, d_pay):
            b.free()
    d_prot.free()


def test_oversized_dense_chunk_is_refused_not_executed(built):
    """A chunk flagged dense must fit the kernel's 12 KiB LDS image; the builders never make a larger one, and one that arrives
    through v2p_stitch_launch_opts is reported in the status word and nothing of it is written."""
    from hip_util import DevBuf
    from vcf2prot_amd import _native as N
    from vcf2prot_amd.cohort import Cohort
    lib = N.hip_lib()
    c = Cohort.preset("C3")
    prot = c.proteome()
    img = c.pack(0, 2, n_threads=1, kernel=2, chunk_tasks=1000, chunk_bytes=65520)
    chunks = np.ascontiguousarray(img.chunks)
    assert (((chunks[:, 1] >> np.uint64(48)) & np.uint64(0x7FF)).max() > 0)
    chunks[:, 1] |= np.uint64(1 << 61)                                   # CHUNK_DENSE on 64 KiB chunks
    d_prot, d_desc, d_chunks, d_pay = DevBuf.of(prot), DevBuf.of(img.desc), DevBuf.of(chunks), DevBuf.of(img.payload)
    d_out = DevBuf(img.out_bytes + 32, fill=0x2E)
    d_status = DevBuf.of(np.full(1, -1, dtype=np.int64))
    bits = int(lib.v2p_stitch_launch_bits(chunks.ctypes.data, chunks.shape[0]))
    assert bits & 2
    rc = N.stitch_launch(lib, None, d_desc.ptr, img.desc.size, d_chunks.ptr, chunks.shape[0], d_prot.ptr, prot.size, d_pay.ptr, img.payload.size,
                         d_out.ptr, img.out_bytes, d_status.ptr, N.LaunchOpts(routing=bits))
    assert rc == 0
    assert d_status.download().view(np.int64)[0] != -1                   # reported
    got = d_out.download()[:img.out_bytes]
    big = ((chunks[:, 1] & np.uint64((1 << 48) - 1)).astype(np.int64))
    assert np.all(got[big[0]:big[0] + 13000] == 0x2E)                    # the first (oversized) chunk's range is untouched
    for b in (d_prot, d_desc, d_chunks, d_pay, d_out, d_status):
        b.free()


@pytest.mark.parametrize("preset,h0,n", [("C2", 1, 2), ("C3", 5, 4), ("C1", 0, 8)])
def test_lds_staged_reference_variant_is_exact(built, coracle, preset, h0, n):
    """stitch4_kernel with the chunk's reference span staged in LDS by global_load_lds_dwordx4 (variants 7 / 11: 36 / 20 KiB windows;
    the design brief's "LDS-staged reference tile", measured 2-4x slower than the L2 gathers and kept only as that measurement):
    same bytes, also where a chunk's span does not fit the window and the gathers take over."""
    from hip_util import DevBuf
    from vcf2prot_amd import _native as N
    from vcf2prot_amd.cohort import Cohort
    lib = N.bench_lib()                                  # (kernel variants 7 / 11 exist only in the bench build of the engine)
    c = Cohort.preset(preset)
    prot = c.proteome()
    want = np.concatenate(oracle_haps(c, coracle, h0, n))
    d_prot = DevBuf.of(prot)
    img = c.pack(h0, h0 + n, n_threads=2, kernel=1)
    chunks = np.ascontiguousarray(img.chunks)
    assert (chunks[:, 1] >> np.uint64(63)).all()
    d_desc, d_chunks, d_pay = DevBuf.of(img.desc), DevBuf.of(chunks), DevBuf.of(img.payload)
    bits = int(lib.v2p_stitch_launch_bits(chunks.ctypes.data, chunks.shape[0]))
    for var in (7, 11):
        d_out = DevBuf(img.out_bytes + 32, fill=0x2E)
        d_status = DevBuf.of(np.full(1, -1, dtype=np.int64))
        rc = lib.v2p_stitch_launch(None, d_desc.ptr, img.desc.size, d_chunks.ptr, chunks.shape[0], d_prot.ptr, prot.size, d_pay.ptr, img.payload.size,
                                   d_out.ptr, img.out_bytes, d_status.ptr, 1 | bits | (var << 12), 0)
        assert rc == 0
        assert d_status.download().view(np.int64)[0] == -1
        got = d_out.download()
        hb = img.hap_out_begin.astype(np.int64)
        res = np.concatenate([got[hb[i]:hb[i] + c.haplotype(h0 + i).n_res] for i in range(n)])
        assert np.array_equal(res, want), (preset, var)
        d_out.free(); d_status.free()
    for b in (d_desc, d_chunks, d_pay, d_prot):
        b.free()
