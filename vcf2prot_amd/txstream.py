"""A v2p_txstream (include/vcf2prot_hip.h) assembled on the host: per-transcript GIRs exactly as step 4b returns them
(TranscriptInstruction::get_g_rep, transcript_instructions.rs:335-427: un-rebased Task SoA, the transcript's own alt tape),
concatenated over the transcripts of every haplotype in result order -- the input of v2p_batch_build_on_device, where step 5
(haplotype_instruction.rs:94-133) and the image packing run as kernels.  With record headers the device emits FASTA text
(personalized_genome.rs:90-113)."""
from __future__ import annotations

import ctypes
from typing import List, Optional

import numpy as np

from ._cohort_api import TxStreamBuf

_PAD = 64          # the builder's slab loads read a few entries past a transcript's last task / alt byte


class TxStreamBuilder:
    def __init__(self, fasta: bool = False):
        self.fasta = fasta
        self.hap_tx_begin: List[int] = [0]
        self._off: List[int] = []
        self._ref_len: List[int] = []
        self._res_len: List[int] = []
        self._task_begin: List[int] = [0]
        self._alt_begin: List[int] = [0]
        self._hdr_off: List[int] = []
        self._hdr_len: List[int] = []
        self._code: List[np.ndarray] = []
        self._sp: List[np.ndarray] = []
        self._ln: List[np.ndarray] = []
        self._sr: List[np.ndarray] = []
        self._alt: List[np.ndarray] = []
        self._n_tasks = 0
        self._n_alt = 0

    def add_transcript(self, code, start_pos, length, start_pos_res, tx_proteome_off: int, tx_ref_len: int, alt, res_len: int,
                       header_off: int = 0, header_len: int = 0):
        """Same arguments as v2p_batch_add_transcript (the host builder's entry point)."""
        code = np.ascontiguousarray(code, dtype=np.uint8)
        n = int(code.size)
        alt = np.ascontiguousarray(alt, dtype=np.uint8)
        for name, arr in (("start_pos", start_pos), ("length", length), ("start_pos_res", start_pos_res)):
            if n and int(np.max(arr)) > 0xFFFFFFFF:
                raise ValueError(f"{name} does not fit the stream's 32-bit fields")
        self._code.append(code)
        self._sp.append(np.asarray(start_pos, dtype=np.uint32))
        self._ln.append(np.asarray(length, dtype=np.uint32))
        self._sr.append(np.asarray(start_pos_res, dtype=np.uint32))
        self._alt.append(alt)
        self._n_tasks += n
        self._n_alt += int(alt.size)
        self._off.append(int(tx_proteome_off)); self._ref_len.append(int(tx_ref_len)); self._res_len.append(int(res_len))
        self._task_begin.append(self._n_tasks); self._alt_begin.append(self._n_alt)
        self._hdr_off.append(int(header_off)); self._hdr_len.append(int(header_len))

    def end_haplotype(self):
        self.hap_tx_begin.append(len(self._off))

    @property
    def n_tx(self) -> int:
        return len(self._off)

    @property
    def n_tasks(self) -> int:
        return self._n_tasks

    def result_bytes(self) -> int:
        return int(sum(self._res_len)) + (int(sum(h + 1 for h in self._hdr_len if h)) if self.fasta else 0)

    def finish(self) -> "HostTxStream":
        def cat(parts, dtype):
            body = np.concatenate(parts) if parts else np.zeros(0, dtype)
            return np.concatenate([body.astype(dtype, copy=False), np.zeros(_PAD, dtype)])
        keep = [np.asarray(self.hap_tx_begin, dtype=np.uint64), np.asarray(self._off, dtype=np.uint64), np.asarray(self._ref_len, dtype=np.uint32),
                np.asarray(self._res_len, dtype=np.uint32), np.asarray(self._task_begin, dtype=np.uint64), np.asarray(self._alt_begin, dtype=np.uint64),
                cat(self._code, np.uint8), cat(self._sp, np.uint32), cat(self._ln, np.uint32), cat(self._sr, np.uint32), cat(self._alt, np.uint8),
                np.asarray(self._hdr_off, dtype=np.uint64), np.asarray(self._hdr_len, dtype=np.uint32)]
        return HostTxStream(keep, len(self.hap_tx_begin) - 1, self.n_tx, self._n_tasks, self._n_alt, self.fasta, self.result_bytes())


class HostTxStream:
    """Owner of the arrays; `.struct` is what v2p_batch_build_on_device takes."""

    def __init__(self, keep, n_haps, n_tx, n_tasks, n_alt, fasta, result_bytes):
        self.keep = [np.ascontiguousarray(a) for a in keep]
        self.result_bytes = result_bytes
        k = self.keep
        s = TxStreamBuf()
        s.n_haps, s.n_tx, s.n_tasks, s.n_alt = n_haps, n_tx, n_tasks, n_alt
        P64, P32, P8 = ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint8)
        s.hap_tx_begin, s.tx_proteome_off = k[0].ctypes.data_as(P64), k[1].ctypes.data_as(P64)
        s.tx_ref_len, s.tx_res_len = k[2].ctypes.data_as(P32), k[3].ctypes.data_as(P32)
        s.tx_task_begin, s.tx_alt_begin = k[4].ctypes.data_as(P64), k[5].ctypes.data_as(P64)
        s.code, s.start_pos, s.length, s.start_pos_res = k[6].ctypes.data_as(P8), k[7].ctypes.data_as(P32), k[8].ctypes.data_as(P32), k[9].ctypes.data_as(P32)
        s.alt = k[10].ctypes.data_as(P8)
        if fasta:
            s.tx_header_off, s.tx_header_len = k[11].ctypes.data_as(P64), k[12].ctypes.data_as(P32)
        self.struct = s

    @property
    def n_tasks(self) -> int:
        return int(self.struct.n_tasks)

    @property
    def n_tx(self) -> int:
        return int(self.struct.n_tx)

    def close(self):
        self.struct = None
        self.keep = []


def pack_rows(stream, proteome_len: int, mode: int = 1, emulate_k: int = 0):
    """The ROWS image of a transcript stream built on the HOST (csrc/rows_image.hpp; include/v2p_cohort.h: v2p_txstream_pack_rows) --
    what v2p_batch_build_on_device(kernel 6 / 7) must reproduce: mode 1 = wave image, 2 = dense; emulate_k = 0: the sequential
    restatement of the packer's state machine, 1 .. 64: the device kernel's tiles / windows / ballot masks emulated
    lane by lane.  Returns a cohort.Packed (chunk table in arena order); raises RowsError(status word) for what the device reports."""
    from . import _native as N
    from ._cohort_api import PackedImage
    from .cohort import Packed, _arr
    lib = N.cohort_lib()
    img = PackedImage()
    status = ctypes.c_uint64(0)
    rc = lib.v2p_txstream_pack_rows(ctypes.byref(stream.struct), proteome_len, mode, emulate_k, ctypes.byref(img), ctypes.byref(status))
    if rc != 0:
        raise RowsError(rc, int(status.value))
    try:
        chunks = (np.ctypeslib.as_array(ctypes.cast(img.chunks, ctypes.POINTER(ctypes.c_uint64)), shape=(int(img.n_chunks) * 2,)).astype(np.uint64, copy=True).reshape(-1, 2)
                  if img.n_chunks else np.zeros((0, 2), dtype=np.uint64))
        return Packed(_arr(img.desc, img.n_desc, np.uint64), chunks, _arr(img.payload, img.n_payload, np.uint8), _arr(img.hap_out_begin, img.n_haps + 1, np.uint64),
                      int(img.n_tasks), int(img.n_copy_bytes), 0)
    finally:
        lib.v2p_packed_free(ctypes.byref(img))


def pack_patch(stream, proteome_len: int):
    """The PATCH image of a transcript stream built on the HOST (csrc/patch_image_host.hpp; include/v2p_cohort.h: v2p_txstream_pack_patch):
    (segments [n_chunks, 1024] u64, patches [n_chunks, 1024] u32, chunk table [n_chunks, 2] u64 in arena order, hap_out_begin, out_bytes,
    total segments, total patches); raises RowsError(status word) for what the device reports (reason 9: the format declines the stream)."""
    from . import _native as N
    from ._cohort_api import PatchImage
    lib = N.cohort_lib()
    img = PatchImage()
    status = ctypes.c_uint64(0)
    rc = lib.v2p_txstream_pack_patch(ctypes.byref(stream.struct), proteome_len, ctypes.byref(img), ctypes.byref(status))
    if rc != 0:
        raise RowsError(rc, int(status.value))
    try:
        n = int(img.n_chunks)
        seg = np.ctypeslib.as_array(img.seg, shape=(max(n, 1) * 1024,))[:n * 1024].astype(np.uint64, copy=True).reshape(n, 1024)
        patch = np.ctypeslib.as_array(img.patch, shape=(max(n, 1) * 1024,))[:n * 1024].astype(np.uint32, copy=True).reshape(n, 1024)
        chunks = (np.ctypeslib.as_array(ctypes.cast(img.chunks, ctypes.POINTER(ctypes.c_uint64)), shape=(n * 2,)).astype(np.uint64, copy=True).reshape(-1, 2)
                  if n else np.zeros((0, 2), dtype=np.uint64))
        hb = np.ctypeslib.as_array(img.hap_out_begin, shape=(int(img.n_haps) + 1,)).astype(np.uint64, copy=True)
        return seg, patch, chunks, hb, int(img.out_bytes), int(img.n_seg), int(img.n_patch)
    finally:
        lib.v2p_patch_image_free(ctypes.byref(img))


def interpret_patch(seg, patch, chunks, src0, src1, out_bytes: int):
    """A PATCH image executed on the host, cell by cell (v2p_patch_interpret): the arena, or None when a chunk is malformed (a cell written
    twice or never, a range out of bounds, a patch outside a reference segment)."""
    from . import _native as N
    lib = N.cohort_lib()
    seg = np.ascontiguousarray(seg, dtype=np.uint64); patch = np.ascontiguousarray(patch, dtype=np.uint32)
    chunks = np.ascontiguousarray(chunks, dtype=np.uint64).reshape(-1, 2)
    src0 = np.ascontiguousarray(src0, dtype=np.uint8); src1 = np.ascontiguousarray(src1, dtype=np.uint8)
    out = np.zeros(out_bytes, dtype=np.uint8)
    rc = lib.v2p_patch_interpret(seg.ctypes.data, patch.ctypes.data, chunks.ctypes.data, chunks.shape[0], src0.ctypes.data, src0.size,
                                 src1.ctypes.data, src1.size, out.ctypes.data, out_bytes)
    return out if rc == 0 else None


def _wave_bytes_per_task() -> int:
    """The library's own wave / dense threshold (v2p_routing_rules: sir_pack.hpp WAVE_BYTES_PER_TASK = 24; profiles/r04_routing_sweep.json:
    the wave kernel wins from ~23 result bytes per task up) -- read from the library so that the Python plan cannot drift from it."""
    from . import _native as N
    return N.routing_rules(0, 0, 0, 0)["wave_bytes_per_task"]


WAVE_BYTES_PER_TASK = 24      # (documentation; build_plan asks the library)


class RowsError(RuntimeError):
    def __init__(self, rc: int, status: int):
        self.rc, self.status = rc, status
        self.index, self.reason = (status >> 8, status & 0xFF) if status != 0xFFFFFFFFFFFFFFFF else (-1, 0)
        super().__init__(f"rows image refused: rc={rc} task/descriptor {self.index} reason {self.reason}")


def build_plan(bytes_per_task: float) -> list:
    """(kernel, window_bytes) pairs to try in order for v2p_batch_build_on_device, by result bytes per task -- the routing the library's one
    call applies.  ROWS images (one pass over the stream, chunks cut afterwards on 1 KiB rows; no window to choose): a wave image (kernel 6)
    from 24 result bytes per task, a dense one (7) below -- and whenever a 1 KiB row of the result holds more descriptors than a wave has
    lanes.  A stream even the dense rows image refuses (a row with more than 1 024 descriptors) goes to the HOST builder (host_build below).
    (Until round 5 the grid builders of rounds 2-3 stood behind them; they live in the development library now.)"""
    bpt = float(bytes_per_task)
    if bpt < _wave_bytes_per_task():
        return [(7, 0)]
    return [(6, 0), (7, 0)]


def build_on_device_auto(batch, stream, result_bytes: Optional[int] = None) -> dict:
    """v2p_batch_build_on_device with build_plan(): an image kind that refuses the stream (V2P_ERR_UNSUPPORTED) is followed by the next;
    the last resort is the host builder, which takes any stream."""
    from ._native import V2PError
    n_tasks = max(int(stream.struct.n_tasks), 1)
    rb = result_bytes if result_bytes is not None else getattr(stream, "result_bytes", 0)
    for kernel, window in build_plan(rb / n_tasks):
        try:
            ms = batch.build_on_device(stream, window, kernel)
            return {"kernel": kernel, "window": window, "build_ms": ms}
        except V2PError as e:
            if e.code != -9:                 # V2P_ERR_UNSUPPORTED: too many descriptors in some row -- anything else is final
                raise
    host_build(batch, stream)
    return {"kernel": -1, "window": 0, "build_ms": 0.0}


def host_build(batch, stream):
    """The last resort for a stream the device builders refuse: its transcripts through the HOST builder (v2p_batch_begin_haplotype /
    _add_transcript / _end_haplotype + v2p_batch_finalize: step 5 on the host, haplotype_instruction.rs:94-133) -- any stream, at host speed."""
    s = stream.struct
    n_h, n_tx, n_tk, n_alt = int(s.n_haps), int(s.n_tx), int(s.n_tasks), int(s.n_alt)

    def arr(p, n, dt):
        return np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].astype(dt, copy=False) if n else np.zeros(0, dt)
    hb, tb, ab = arr(s.hap_tx_begin, n_h + 1, np.uint64), arr(s.tx_task_begin, n_tx + 1, np.uint64), arr(s.tx_alt_begin, n_tx + 1, np.uint64)
    off, rl, res = arr(s.tx_proteome_off, n_tx, np.uint64), arr(s.tx_ref_len, n_tx, np.uint32), arr(s.tx_res_len, n_tx, np.uint32)
    code, sp, ln, sr = arr(s.code, n_tk, np.uint8), arr(s.start_pos, n_tk, np.uint32), arr(s.length, n_tk, np.uint32), arr(s.start_pos_res, n_tk, np.uint32)
    alt = arr(s.alt, n_alt, np.uint8)
    fasta = bool(s.tx_header_off) and bool(s.tx_header_len)
    ho = arr(s.tx_header_off, n_tx, np.uint64) if fasta else None
    hl = arr(s.tx_header_len, n_tx, np.uint32) if fasta else None
    for h in range(n_h):
        batch.begin_haplotype()
        for t in range(int(hb[h]), int(hb[h + 1])):
            a, b = int(tb[t]), int(tb[t + 1])
            batch.add_transcript(code[a:b], sp[a:b].astype(np.uint64), ln[a:b].astype(np.uint64), sr[a:b].astype(np.uint64), int(off[t]), int(rl[t]),
                                 alt[int(ab[t]):int(ab[t + 1])], int(res[t]), int(ho[t]) if fasta else 0, int(hl[t]) if fasta else 0)
        batch.end_haplotype()
    batch.finalize()
