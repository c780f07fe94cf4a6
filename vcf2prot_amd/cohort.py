"""Synthetic cohorts at the Task boundary (binding of include/v2p_cohort.h).

``Cohort.preset("C2")`` etc. are BASELINE.json's configs as concretised in
SURVEY.md section 8d.  A haplotype comes back as the reference's executor
receives it: SoA Task arrays (gir.rs:283-299), alt tape, ref-tape layout,
annotation (haplotype_instruction.rs:75-137).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from . import _native as N
from ._cohort_api import CohortParams, HapView, PackedImage, TxStreamBuf


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(int(n),)).astype(dtype, copy=True)


@dataclass
class Haplotype:
    index: int
    code: np.ndarray
    start_pos: np.ndarray
    length: np.ndarray
    start_pos_res: np.ndarray
    alt: np.ndarray              # uint8
    n_res: int
    n_ref: int
    seg_ref_begin: np.ndarray    # [n_seg + 1]
    seg_proteome_off: np.ndarray
    tx_id: np.ndarray
    tx_res_begin: np.ndarray
    tx_res_end: np.ndarray

    @property
    def n_tasks(self) -> int:
        return int(self.code.size)


@dataclass
class Packed:
    desc: np.ndarray
    chunks: np.ndarray           # (n_chunks, 2) uint64
    payload: np.ndarray
    hap_out_begin: np.ndarray
    n_tasks: int                 # N of the roofline formula
    n_copy_bytes: int            # A of the roofline formula
    max_chunk_tasks: int = 0     # largest chunk (selects the kernel's descriptors per lane)

    @property
    def tasks_per_lane(self) -> int:
        return (self.launch_bits >> 8) & 15

    @property
    def launch_bits(self) -> int:
        """Bits 4..11 of v2p_stitch_launch's flags: which kernels the chunk table needs and their tasks per lane."""
        ch = np.ascontiguousarray(self.chunks)
        # (the host-only twin in libv2p_cohort.so: looking at an image needs neither hipcc nor a HIP runtime)
        return int(N.cohort_lib().v2p_cohort_launch_bits(ch.ctypes.data if ch.size else None, ch.shape[0]))

    @property
    def out_bytes(self) -> int:
        return int(self.hap_out_begin[-1])


class Cohort:
    def __init__(self, params: CohortParams):
        self._lib = N.cohort_lib()
        self.params = params
        h = ctypes.c_void_p()
        if self._lib.v2p_cohort_create(ctypes.byref(params), ctypes.byref(h)) != 0:
            raise ValueError("invalid cohort parameters")
        self._h = h
        self._buf = ctypes.c_void_p(self._lib.v2p_hapbuf_create())

    @staticmethod
    def preset_params(name: str) -> CohortParams:
        p = CohortParams()
        if N.cohort_lib().v2p_cohort_preset(name.encode(), ctypes.byref(p)) != 0:
            raise ValueError(f"unknown cohort preset {name!r}")
        return p

    @classmethod
    def preset(cls, name: str, **overrides) -> "Cohort":
        p = cls.preset_params(name)
        for k, v in overrides.items():
            if k == "mix":
                for i, x in enumerate(v):
                    p.mix[i] = x
            else:
                setattr(p, k, v)
        return cls(p)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.v2p_hapbuf_destroy(self._buf)
            self._lib.v2p_cohort_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_haplotypes(self) -> int:
        return int(self._lib.v2p_cohort_n_haplotypes(self._h))

    @property
    def n_transcripts(self) -> int:
        return int(self._lib.v2p_cohort_n_transcripts(self._h))

    def proteome(self) -> np.ndarray:
        n = int(self._lib.v2p_cohort_proteome_len(self._h))
        return _arr(self._lib.v2p_cohort_proteome(self._h), n, np.uint8)

    def tx_offsets(self) -> np.ndarray:
        return _arr(self._lib.v2p_cohort_tx_offsets(self._h), self.n_transcripts + 1, np.uint64)

    @staticmethod
    def tx_name(t: int) -> str:
        return f"ENST{t:011d}"

    def haplotype(self, h: int) -> Haplotype:
        v = HapView()
        if self._lib.v2p_cohort_generate(self._h, h, self._buf, ctypes.byref(v)) != 0:
            raise RuntimeError("cohort generation failed")
        return Haplotype(h, _arr(v.code, v.n_tasks, np.uint8), _arr(v.start_pos, v.n_tasks, np.uint64),
                         _arr(v.length, v.n_tasks, np.uint64), _arr(v.start_pos_res, v.n_tasks, np.uint64),
                         _arr(v.alt, v.n_alt, np.uint8), int(v.n_res), int(v.n_ref),
                         _arr(v.seg_ref_begin, v.n_seg + 1, np.uint64), _arr(v.seg_proteome_off, v.n_seg, np.uint64),
                         _arr(v.tx_id, v.n_tx, np.uint32), _arr(v.tx_res_begin, v.n_tx, np.uint64),
                         _arr(v.tx_res_end, v.n_tx, np.uint64))

    def ref_tape_u32(self, h: int) -> np.ndarray:
        """The private ref tape step 5 builds for haplotype h (Rust chars)."""
        v = HapView()
        if self._lib.v2p_cohort_generate(self._h, h, self._buf, ctypes.byref(v)) != 0:
            raise RuntimeError("cohort generation failed")
        out = np.empty(int(v.n_ref), dtype=np.uint32)
        if self._lib.v2p_cohort_ref_tape_u32(self._h, ctypes.byref(v), out.ctypes.data if out.size else None) != 0:
            raise RuntimeError("ref tape materialisation failed")
        return out

    def describe(self, h: int) -> List[Tuple[int, str, str]]:
        """[(transcript index, csq type, aa change)] of haplotype h, in generation order."""
        need = int(self._lib.v2p_cohort_describe(self._h, h, None, 0))
        buf = ctypes.create_string_buffer(need + 1)
        self._lib.v2p_cohort_describe(self._h, h, buf, need + 1)
        out = []
        for line in buf.value.decode().split("\n"):
            if line:
                t, kind, aa = line.split("\t")
                out.append((int(t), kind, aa))
        return out

    def result_sizes(self, h0: int = 0, h1: int = -1, n_threads: int = 0) -> np.ndarray:
        """Result tape length of every haplotype in [h0, h1) (for byte-balanced sharding, shard.shard_by_bytes)."""
        import os
        h1 = self.n_haplotypes if h1 < 0 else h1
        out = np.zeros(max(h1 - h0, 0), dtype=np.uint64)
        if out.size and self._lib.v2p_cohort_result_sizes(self._h, h0, h1, n_threads or min(64, os.cpu_count() or 1), out.ctypes.data) != 0:
            raise RuntimeError("v2p_cohort_result_sizes failed")
        return out

    def txstream(self, h0: int, h1: int, n_threads: int = 0) -> "TxStream":
        """Haplotypes [h0, h1) one step before the image: per-transcript GIRs as step 4b returns them (un-rebased), concatenated --
        the input of Batch.build_on_device."""
        import os
        buf = TxStreamBuf()
        if self._lib.v2p_cohort_txstream(self._h, h0, h1, n_threads or min(64, os.cpu_count() or 1), ctypes.byref(buf)) != 0:
            raise RuntimeError("v2p_cohort_txstream failed")
        return TxStream(self._lib, buf)

    def pack_grid(self, h0: int, h1: int, window_bytes: int, kernel: int = 2) -> Packed:
        """The host image cut on a fixed result grid -- what the device-side builder must reproduce byte for byte."""
        img = PackedImage()
        rc = self._lib.v2p_cohort_pack_grid(self._h, h0, h1, window_bytes, kernel, ctypes.byref(img))
        if rc != 0:
            raise RuntimeError(f"v2p_cohort_pack_grid failed ({rc})")
        try:
            desc = _arr(img.desc, img.n_desc, np.uint64)
            chunks = (np.ctypeslib.as_array(ctypes.cast(img.chunks, ctypes.POINTER(ctypes.c_uint64)),
                                            shape=(int(img.n_chunks) * 2,)).astype(np.uint64, copy=True).reshape(-1, 2)
                      if img.n_chunks else np.zeros((0, 2), dtype=np.uint64))
            return Packed(desc, chunks, _arr(img.payload, img.n_payload, np.uint8), _arr(img.hap_out_begin, img.n_haps + 1, np.uint64),
                          int(img.n_tasks), int(img.n_copy_bytes), int(img.max_chunk_tasks))
        finally:
            self._lib.v2p_packed_free(ctypes.byref(img))

    HEADER_BYTES = 19

    def fasta_headers(self) -> np.ndarray:
        """Resident header table for FASTA emit: a leading line feed, then '>ENST%011d_h\\n' at 1 + (2*t + parity) * 19."""
        need = int(self._lib.v2p_cohort_fasta_headers(self._h, None, 0))
        out = np.empty(need, dtype=np.uint8)
        self._lib.v2p_cohort_fasta_headers(self._h, out.ctypes.data, need)
        return out

    def pack(self, h0: int, h1: int, n_threads: int = 0, chunk_tasks: int = 0, chunk_bytes: int = 0,
             fasta: bool = False, cut_align: int = 0, soft_window: int = 0, inline_payload: bool = True, fuse: bool = True, double: bool = True,
             kernel: int = 0, line_cut: bool = True) -> Packed:
        import os
        img = PackedImage()
        nt = n_threads or min(32, os.cpu_count() or 1)
        rc = self._lib.v2p_cohort_pack(self._h, h0, h1, nt, chunk_tasks, chunk_bytes, (1 if fasta else 0) | (0 if inline_payload else 2) | (0 if fuse else 4) | (0 if double else 8) | {2: 16, 1: 32, 3: 64, 4: 128}.get(kernel, 0) | (cut_align << 8) | (soft_window << 24) | (0 if line_cut else 0x800000), ctypes.byref(img))
        if rc != 0:
            raise RuntimeError(f"v2p_cohort_pack failed ({rc})")
        try:
            desc = _arr(img.desc, img.n_desc, np.uint64)
            chunks = (np.ctypeslib.as_array(ctypes.cast(img.chunks, ctypes.POINTER(ctypes.c_uint64)),
                                            shape=(int(img.n_chunks) * 2,)).astype(np.uint64, copy=True).reshape(-1, 2)
                      if img.n_chunks else np.zeros((0, 2), dtype=np.uint64))
            payload = _arr(img.payload, img.n_payload, np.uint8)
            hb = _arr(img.hap_out_begin, img.n_haps + 1, np.uint64)
            return Packed(desc, chunks, payload, hb, int(img.n_tasks), int(img.n_copy_bytes), int(img.max_chunk_tasks))
        finally:
            self._lib.v2p_packed_free(ctypes.byref(img))


class TxStream:
    """Owner of a v2p_txstream_buf; `.struct` is what v2p_batch_build_on_device takes."""

    def __init__(self, lib, buf: TxStreamBuf):
        self._lib, self.struct = lib, buf

    @property
    def n_tasks(self) -> int:
        return int(self.struct.n_tasks)

    @property
    def n_tx(self) -> int:
        return int(self.struct.n_tx)

    @property
    def nbytes(self) -> int:
        s = self.struct
        return int((s.n_haps + 1) * 8 + s.n_tx * 32 + 16 + s.n_tasks * 13 + s.n_alt)

    def close(self):
        if self.struct is not None:
            self._lib.v2p_txstream_free(ctypes.byref(self.struct))
            self.struct = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
