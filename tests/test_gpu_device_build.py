"""step 5 (haplotype_instruction.rs:94-133) and the image packing ON THE DEVICE (csrc/build_kernels.hip).
The device-built image must equal, byte for byte, the image the host builder cuts on the same result grid, and the result tapes
must equal the oracle's."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def oracle_hap(c, coracle, h):
    hap = c.haplotype(h)
    t = coracle.pack_tasks(hap.code, hap.start_pos, hap.length, hap.start_pos_res)
    return coracle.gir_execute_u8(t, c.ref_tape_u32(h).astype(np.uint8), hap.alt, np.full(hap.n_res, ord("."), dtype=np.uint8))


@pytest.mark.parametrize("preset,h0,n,window,kernel", [
    ("C1", 0, 8, 4096, 2), ("C1", 0, 8, 4096, 1), ("C2", 5, 3, 28672, 1), ("C2", 5, 3, 32768, 2),
    ("C3", 100, 40, 16384, 2), ("C3", 100, 40, 16384, 1), ("C4", 7, 3, 32768, 2), ("C5", 50, 300, 4096, 2),
    ("C5", 50, 300, 4096, 3), ("C5", 11, 100, 8192, 3), ("C5", 200, 150, 12288, 3), ("C1", 0, 8, 4096, 3),     # 3: dense image (fused descriptors, stitch_dense_kernel)
    ("C2", 5, 3, 10240, 4), ("C2", 9, 2, 4096, 4), ("C3", 100, 40, 4096, 4), ("C4", 7, 3, 4096, 4)])     # 4: wave image (stitchw_kernel)
def test_device_built_image_equals_host_grid_image(built, dev_ctx, coracle, preset, h0, n, window, kernel):
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset(preset)
    dev_ctx.upload_proteome(c.proteome())
    want = c.pack_grid(h0, h0 + n, window, kernel)
    stream = c.txstream(h0, h0 + n, n_threads=3)
    b = dev_ctx.batch()
    ms = b.build_on_device(stream, window, kernel)
    assert ms > 0
    desc, chunks, hb = b.download_image()
    assert np.array_equal(hb, want.hap_out_begin)
    assert desc.size == want.desc.size and np.array_equal(desc, want.desc)
    # the host table is in result order until finalize() deals it to the XCDs; order both the same way
    wc = np.ascontiguousarray(want.chunks)
    dev_ctx._lib.v2p_order_chunks_for_xcds(wc.ctypes.data, wc.shape[0], want.desc.ctypes.data, want.desc.size, c.proteome().size)
    assert chunks.shape == wc.shape and np.array_equal(chunks, wc)
    b.execute()
    b.sync()
    for i in range(n):
        assert np.array_equal(b.download_hap(i), oracle_hap(c, coracle, h0 + i)), (preset, h0 + i)
    b.close()
    stream.close()


def test_device_build_reports_what_the_reference_would_panic_on(built, dev_ctx):
    """A task that reads beyond its transcript (task.rs:43/47) is found by the count pass; nothing is emitted."""
    from vcf2prot_amd._native import V2PError
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset("C3")
    dev_ctx.upload_proteome(c.proteome())
    stream = c.txstream(0, 4, n_threads=1)
    s = stream.struct
    victim = 1234 % int(s.n_tasks)
    while s.code[victim] != 0:
        victim += 1
    old = s.length[victim]
    s.length[victim] = 1 << 30
    b = dev_ctx.batch()
    with pytest.raises(V2PError) as e:
        b.build_on_device(stream, 16384, 2)
    assert e.value.index == victim
    s.length[victim] = old
    b.close()
    b2 = dev_ctx.batch()                                   # a window too dense for one chunk is refused, not mis-built
    dense = Cohort.preset("C5")
    dev_ctx.upload_proteome(dense.proteome())
    st2 = dense.txstream(0, 20, n_threads=1)
    with pytest.raises(V2PError):
        b2.build_on_device(st2, 32768, 2)
    b2.close()


def test_device_build_refuses_broken_offset_tables(built, dev_ctx):
    """The kernels index device memory through every entry of hap_tx_begin / tx_task_begin / tx_alt_begin: a table that is not
    ascending from 0 or leaves its array is refused on the host, with the offending index, and the batch stays usable."""
    from vcf2prot_amd._native import V2PError
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset("C3")
    dev_ctx.upload_proteome(c.proteome())
    stream = c.txstream(0, 3, n_threads=1)
    s = stream.struct
    n_tx = int(s.n_tx)
    b = dev_ctx.batch()
    for name, idx, value in [("tx_task_begin", 7, int(s.n_tasks) + 5), ("tx_task_begin", 9, 0), ("tx_alt_begin", 11, int(s.n_alt) + 1),
                             ("tx_alt_begin", 12, 0), ("hap_tx_begin", 1, n_tx + 3), ("hap_tx_begin", 2, 0), ("tx_proteome_off", 5, (1 << 64) - 1)]:
        arr = getattr(s, name)
        old = arr[idx]
        arr[idx] = value
        with pytest.raises(V2PError) as e:
            b.build_on_device(stream, 16384, 2)
        assert e.value.code == -1, name                                    # V2P_ERR_INVALID_ARG, nothing was launched
        arr[idx] = old
    ms = b.build_on_device(stream, 16384, 2)                              # the same batch builds once the stream is whole again
    assert ms > 0 and b.counts()["n_haps"] == 3
    b.close()
    stream.close()


def test_device_build_can_be_retried_with_a_smaller_window(built, dev_ctx):
    """A window with too many descriptors is refused (V2P_ERR_UNSUPPORTED); the header says "pick a smaller window", and the same
    batch then takes one."""
    from vcf2prot_amd._native import V2PError
    from vcf2prot_amd.cohort import Cohort
    dense = Cohort.preset("C5")
    dev_ctx.upload_proteome(dense.proteome())
    st = dense.txstream(0, 20, n_threads=1)
    b = dev_ctx.batch()
    with pytest.raises(V2PError):
        b.build_on_device(st, 32768, 2)
    assert b.counts()["n_haps"] == 0
    b.build_on_device(st, 4096, 3)
    assert b.counts()["n_haps"] == 20
    b.execute()
    b.sync()
    b.close()
    st.close()


@pytest.mark.parametrize("preset,h0,n,window,fasta", [
    ("C3", 100, 40, 8192, False), ("C3", 100, 40, 10240, False), ("C3", 7, 25, 6144, False), ("C3", 100, 12, 4096, True), ("C4", 7, 3, 10240, False),
    ("C4", 7, 3, 6144, True), ("C2", 5, 3, 10240, False), ("C2", 5, 2, 10240, True), ("C1", 0, 8, 2048, False)])
def test_wave_windows_that_split(built, dev_ctx, coracle, preset, h0, n, window, fasta):
    """kernel = 5: wave windows of up to ten 1 KiB rows where a window of 65 .. 127 descriptors becomes TWO chunks, cut on a row, the
    descriptor under the cut split in two (copy / fill / immediate / fused substitution on either side of its literal) -- so the grid
    can be as coarse as the AVERAGE window allows, not the worst one.  Every chunk respects what one wave takes, every haplotype is
    the oracle's (with FASTA emit: the host packer's file text)."""
    import ctypes
    from vcf2prot_amd.cohort import Cohort
    c = Cohort.preset(preset)
    dev_ctx.upload_reference(c.proteome(), c.fasta_headers())
    stream = c.txstream(h0, h0 + n, n_threads=3)
    keep = []
    if fasta:
        off, ln = [], []
        for h in range(h0, h0 + n):
            for tx in c.haplotype(h).tx_id:
                off.append(1 + (2 * int(tx) + (h & 1)) * Cohort.HEADER_BYTES)
                ln.append(Cohort.HEADER_BYTES)
        keep = [np.array(off, dtype=np.uint64), np.array(ln, dtype=np.uint32)]
        stream.struct.tx_header_off = keep[0].ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
        stream.struct.tx_header_len = keep[1].ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
    b = dev_ctx.batch()
    b.build_on_device(stream, window, 5)
    desc, chunks, hb = b.download_image()
    nd = ((chunks[:, 1] >> np.uint64(48)) & np.uint64(0x7FF)).astype(np.int64)
    dst = (chunks[:, 1] & np.uint64((1 << 48) - 1)).astype(np.int64)
    n_windows = (int(hb[-1]) + window - 1) // window
    assert nd.max() <= 64 and (dst % 1024 == 0).all() and chunks.shape[0] >= n_windows
    assert (((chunks[:, 1] >> np.uint64(60)) & np.uint64(0xF)) == 1).all()                       # CHUNK_WAVE only
    order = np.argsort(dst)
    ends = np.concatenate([dst[order][1:], [int(hb[-1])]])
    assert ((ends - dst[order]) <= window).all() and ((ends - dst[order]) > 0).all()
    if preset in ("C3", "C4") and (window >= 8192 or fasta):
        assert chunks.shape[0] > n_windows                                                        # some windows did split
    b.execute()
    b.sync()
    if fasta:
        host = c.pack(h0, h0 + n, n_threads=2, fasta=True)
        hbatch = dev_ctx.batch()
        hbatch.set_packed(host.desc, host.chunks, host.payload, host.hap_out_begin)
        hbatch.finalize()
        hbatch.execute()
        hbatch.sync()
        for i in range(n):
            assert np.array_equal(b.download_hap(i), hbatch.download_hap(i)), (preset, window, i)
        hbatch.close()
    else:
        for i in range(n):
            assert np.array_equal(b.download_hap(i), oracle_hap(c, coracle, h0 + i)), (preset, window, h0 + i)
    b.close()
    stream.close()
