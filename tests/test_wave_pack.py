"""Host logic of wave images (sir_pack.hpp, kernel choice 4: chunks of <= 64 descriptors and <= 10 KiB for stitchw_kernel), no GPU:
interpreted in numpy the image equals the per-block image of the same haplotypes; every chunk respects what one wave takes;
parts packed by several threads are cut in arena coordinates."""
import numpy as np
import pytest

from gen_util import interpret_image


def _geometry(img):
    nd = ((img.chunks[:, 1] >> np.uint64(48)) & np.uint64(0x7FF)).astype(np.int64)
    dst = (img.chunks[:, 1] & np.uint64((1 << 48) - 1)).astype(np.int64)
    order = np.argsort(dst, kind="stable")
    ends = np.concatenate([dst[order][1:], [img.out_bytes]])
    return nd, dst[order], ends - dst[order]


@pytest.mark.parametrize("preset,h0,n", [("C1", 0, 8), ("C2", 1, 2), ("C3", 40, 4), ("C4", 3, 2), ("C5", 7, 12)])
def test_wave_image_equals_the_per_block_image(built, preset, h0, n):
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset(preset)
    prot = c.proteome()
    base = c.pack(h0, h0 + n, n_threads=1, kernel=2)
    want = interpret_image(base.desc, base.chunks, prot, base.payload, base.out_bytes)
    for threads in (1, 3):
        w = c.pack(h0, h0 + n, n_threads=threads, kernel=4)
        assert np.array_equal(w.hap_out_begin, base.hap_out_begin)
        assert np.array_equal(interpret_image(w.desc, w.chunks, prot, w.payload, w.out_bytes), want), (preset, threads)
        nd, dst, nbytes = _geometry(w)
        assert nd.max() <= 64 and ((dst & 15) + nbytes).max() <= 10240
        assert (((w.chunks[:, 1] >> np.uint64(60)) & np.uint64(0xF)) == 1).all()              # CHUNK_WAVE and no other routing flag
        assert (w.launch_bits & 4) and (w.launch_bits & 48) == 48
        if preset == "C2":
            # cuts are aligned in the ARENA whatever part a thread packed: whole 1 KiB rows, nearly always ten of them
            assert (dst % 1024 == 0).mean() > 0.9 and (nbytes == 10240).mean() > 0.8, (preset, threads)


def test_long_run_cohorts_choose_the_wave_kernel(built):
    """The packer's own choice (kernel 0): >= 40 result bytes per task -> a wave image with fused substitutions (C2: 134, C3: 84);
    C5 (7) is a dense one."""
    from vcf2prot_amd.cohort import Cohort
    c2 = Cohort.preset("C2").pack(0, 2, n_threads=2)
    assert (c2.launch_bits & 4) and (c2.launch_bits & 48) == 48 and ((c2.desc >> np.uint64(61)) == 7).sum() > 0.9 * 2 * 20000
    c3 = Cohort.preset("C3").pack(0, 2, n_threads=2)
    assert (c3.launch_bits & 4) and (c3.launch_bits & 48) == 48 and ((c3.desc >> np.uint64(61)) == 7).sum() > 0
    c5 = Cohort.preset("C5").pack(0, 4, n_threads=2)
    assert (c5.launch_bits & 2) and not (c5.launch_bits & 4)


@pytest.mark.parametrize("chunk_tasks,chunk_bytes,cut_align,soft_window", [(3, 64, 16, 0), (17, 1000, 16, 2), (64, 4096, 64, 8), (64, 300, 16, 0), (5, 8192, 4096, 1), (33, 2049, 16, 3)])
def test_wave_image_any_chunking(built, chunk_tasks, chunk_bytes, cut_align, soft_window):
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset("C3")
    prot = c.proteome()
    base = c.pack(40, 43, n_threads=1, kernel=2)
    want = interpret_image(base.desc, base.chunks, prot, base.payload, base.out_bytes)
    w = c.pack(40, 43, n_threads=2, kernel=4, chunk_tasks=chunk_tasks, chunk_bytes=chunk_bytes, cut_align=cut_align, soft_window=soft_window)
    assert np.array_equal(interpret_image(w.desc, w.chunks, prot, w.payload, w.out_bytes), want)
    nd, dst, nbytes = _geometry(w)
    assert nd.max() <= min(chunk_tasks, 64) and nbytes.max() <= chunk_bytes and ((dst & 15) + nbytes).max() <= 10240


def test_wave_grid_image_equals_oracle_and_refuses_overfull_windows(built, coracle):
    """Grid cutting for the device builder: chunk k = result bytes [k*W, (k+1)*W), W <= 10 KiB; a window with more than 64 descriptors
    is refused (the device builder reports the same window)."""
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset("C2")
    img = c.pack_grid(0, 2, 10240, 4)
    dst = (img.chunks[:, 1] & np.uint64((1 << 48) - 1)).astype(np.int64)
    assert np.array_equal(dst, np.arange(dst.size) * 10240) and (img.launch_bits & 4)
    out = interpret_image(img.desc, img.chunks, c.proteome(), img.payload, img.out_bytes)
    for h in range(2):
        hap = c.haplotype(h)
        t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
        want = coracle.gir_execute_u8(t, c.ref_tape_u32(h).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8))
        a, b = int(img.hap_out_begin[h]), int(img.hap_out_begin[h + 1])
        assert np.array_equal(out[a:b], want), h
    with pytest.raises(RuntimeError):
        Cohort.preset("C5").pack_grid(0, 4, 8192, 4)               # ~1 000 descriptors per 8 KiB window
    with pytest.raises(RuntimeError):
        c.pack_grid(0, 2, 12288, 4)                                 # a wave chunk is at most ten rows


def test_chunk_order_inside_blocks_of_the_arena(built):
    """v2p_order_chunks_for_xcds (host logic, no GPU): for a descriptor-rich image the XCD / window order is applied inside blocks of
    the arena -- equal shares of the table's entries in result order, about eight times the proteome each -- so the table stays a
    permutation, every block holds exactly its own stretch of the arena, and inside a block entry 8j + x reads proteome slice x;
    a thin image (C2) keeps one order for the whole table."""
    from vcf2prot_amd import _native as N
    from vcf2prot_amd.cohort import Cohort
    lib = N.hip_lib()
    c = Cohort.preset("C3")
    n_prot = c.proteome().size
    img = c.pack(0, 160, n_threads=4)                                   # ~290 MB of result: several 64 MB blocks
    before = np.ascontiguousarray(img.chunks).copy()
    after = before.copy()
    assert lib.v2p_order_chunks_for_xcds(after.ctypes.data, after.shape[0], img.desc.ctypes.data, img.desc.size, n_prot) == 0
    key = lambda t: t[:, 0].astype(np.uint64) * np.uint64(1 << 20) + (t[:, 1] & np.uint64((1 << 20) - 1))
    assert np.array_equal(np.sort(key(before)), np.sort(key(after)))  # a permutation
    dst_b = (before[:, 1] & np.uint64((1 << 48) - 1)).astype(np.int64)
    dst_a = (after[:, 1] & np.uint64((1 << 48) - 1)).astype(np.int64)
    assert (np.diff(dst_b) >= 0).all()                                  # the packer emits arena order
    n = before.shape[0]
    span = int(dst_b[-1])
    nb = min((span + 8 * n_prot - 1) // (8 * n_prot), n // 64)
    assert nb >= 3
    first = [((n * k // nb) & ~7) if k < nb else n for k in range(nb + 1)]
    for k in range(nb):
        a, b = first[k], first[k + 1]
        assert np.array_equal(np.sort(dst_a[a:b]), dst_b[a:b]), k       # block k = its own stretch of the arena, reordered inside
        # entry 8j + x of the block: a chunk whose first reference read lies in proteome slice x (while all slices still have chunks)
        tb = after[a:a + 64, 0].astype(np.int64)
        d = img.desc[tb]
        snv = (d >> np.uint64(61)) == np.uint64(7)
        src = np.where(snv, d & np.uint64((1 << 29) - 1), d & np.uint64((1 << 40) - 1)).astype(np.int64)
        is_ref = snv | ((d >> np.uint64(62)) == 0)
        per = (n_prot + 7) // 8
        ok = ~is_ref | (np.minimum(src // per, 7) == (np.arange(64) % 8))
        assert ok.mean() > 0.9, (k, ok.mean())
    one = before.copy()
    os_env = __import__("os").environ
    blib = N.bench_lib()                                                # (the experiment switch is read by the bench build of the library only)
    os_env["V2P_ORDER_MAX_BLOCKS"] = "1"
    try:
        assert blib.v2p_order_chunks_for_xcds(one.ctypes.data, one.shape[0], img.desc.ctypes.data, img.desc.size, n_prot) == 0
    finally:
        os_env.pop("V2P_ORDER_MAX_BLOCKS", None)
    assert not np.array_equal(one, after)                               # one order over the whole table is a different table
    c2 = Cohort.preset("C2")
    img2 = c2.pack(0, 40, n_threads=4)                                  # 320 MB, descriptors 2 % of it: thin
    t1 = np.ascontiguousarray(img2.chunks).copy(); t2 = t1.copy()
    assert lib.v2p_order_chunks_for_xcds(t1.ctypes.data, t1.shape[0], img2.desc.ctypes.data, img2.desc.size, c2.proteome().size) == 0
    os_env["V2P_ORDER_MAX_BLOCKS"] = "1"
    try:
        assert blib.v2p_order_chunks_for_xcds(t2.ctypes.data, t2.shape[0], img2.desc.ctypes.data, img2.desc.size, c2.proteome().size) == 0
    finally:
        os_env.pop("V2P_ORDER_MAX_BLOCKS", None)
    # (a thin image keeps ONE order only from 2 GB on -- round 4's sweep -- which tests/test_routing_rules.py pins; this 320 MB one is dealt in blocks)
    assert not np.array_equal(t1, t2) and np.array_equal(np.sort(t1[:, 1]), np.sort(t2[:, 1]))
