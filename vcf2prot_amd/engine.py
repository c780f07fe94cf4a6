"""Host-side mirror of the reference's step-6 interface on top of the C ABI.

Names and argument meaning follow the reference so parity tests read like its
own tests (paths under /root/reference/src/data_structures/InternalRep):

* ``Engine``  -- engines.rs:15-29 (``Engine.from_str("gpu")``)
* ``Task``    -- task.rs:2-18
* ``GIR``     -- gir.rs:15-46; ``GIR.execute(engine)`` is gir.rs:197-241 with the
  ``Engine::GPU`` arm implemented (it panics in the reference, gir.rs:236-239).
  ``Engine.ST``/``Engine.MT`` are NOT implemented here: this package is the gpu
  plugin only, the CPU engines stay in the Rust host.
* ``Context`` / ``Batch`` -- thin owners of ``v2p_ctx`` / ``v2p_batch``.

Error behaviour: the reference panics (process abort); here a ``V2PError`` is
raised with the same condition (bounds, stream code, contiguity under DEBUG_GPU).
"""
from __future__ import annotations

import ctypes
import enum
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _native as N
from ._native import V2PError


class Engine(enum.Enum):
    ST = 0
    MT = 1
    GPU = 2

    @staticmethod
    def from_str(name: str) -> "Engine":
        out = ctypes.c_int(-1)
        rc = N.hip_lib().v2p_engine_from_str(name.encode(), ctypes.byref(out))
        if rc != N.V2P_OK:
            raise ValueError(f"{name} is not a supported engine")  # engines.rs:27
        return Engine(out.value)


@dataclass(frozen=True)
class Task:
    exe_code: int
    start_pos: int
    length: int
    start_pos_res: int


def _soa(tasks: Sequence[Task]):
    n = len(tasks)
    code = np.fromiter((t.exe_code for t in tasks), dtype=np.uint8, count=n)
    sp = np.fromiter((t.start_pos for t in tasks), dtype=np.uint64, count=n)
    ln = np.fromiter((t.length for t in tasks), dtype=np.uint64, count=n)
    sr = np.fromiter((t.start_pos_res for t in tasks), dtype=np.uint64, count=n)
    return code, sp, ln, sr


def _p(a: Optional[np.ndarray]):
    return None if a is None or a.size == 0 else a.ctypes.data


class Context:
    """One ``v2p_ctx`` (one HIP stream on one GPU).  Not thread-safe by design: one per worker."""

    def __init__(self, device: int = 0, debug_gpu: Optional[bool] = None, temporal_stores: bool = False, result_order: bool = False, development: bool = False):
        # development: libv2p_bench.so -- the same engine compiled with V2P_BENCH_VARIANTS (the A/B switches of the builders and launchers,
        # the grid builders of rounds 2-3, PATCH images: csrc/bench/v2p_bench.h).  Tools and the tests of those paths; never the product.
        self._lib = N.bench_lib() if development else N.hip_lib()
        self.development = development
        if debug_gpu is None:  # README.md:156-157: DEBUG_GPU is an environment flag
            debug_gpu = "DEBUG_GPU" in os.environ
        flags = (N.V2P_FLAG_DEBUG_GPU if debug_gpu else 0) | (N.V2P_FLAG_TEMPORAL if temporal_stores else 0) | (4 if result_order else 0)   # 4: V2P_FLAG_RESULT_ORDER (chunk tables are launched as given)
        h = ctypes.c_void_p()
        rc = self._lib.v2p_init(device, flags, ctypes.byref(h))
        if rc != N.V2P_OK:
            raise V2PError(rc, (self._lib.v2p_last_error(None) or b"").decode())
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.v2p_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc != N.V2P_OK:
            raise V2PError(rc, (self._lib.v2p_last_error(self._h) or b"").decode(),
                           int(self._lib.v2p_last_error_index(self._h)))

    def set_stream(self, hip_stream: int):
        self._check(self._lib.v2p_set_stream(self._h, ctypes.c_void_p(hip_stream)))

    def set_launch_opts(self, phase_bytes: int = 0, phase_min_chunks: int = 0, store_sc1: int = -1, variant: int = 0):
        """Phase size / phase threshold / store policy of every batch this context executes from now on (A/B runs, tests); no arguments:
        the library's defaults (v2p_set_launch_opts).  variant (development contexts only: v2p_bench_set_variant, csrc/bench/v2p_bench.h):
        16 / 17 / 18 one launch for all phases / the read-ahead as kernels of its own / no read-ahead, 20 .. 29 the builders' switches."""
        o = N.LaunchOpts(1, 0, phase_bytes, phase_min_chunks, store_sc1, 0, 0)
        self._check(self._lib.v2p_set_launch_opts(self._h, ctypes.byref(o)))
        if variant or self.development:
            if not self.development:
                raise ValueError("launch variants (A/B switches) exist in the development library only: Context(development=True)")
            self._check(self._lib.v2p_bench_set_variant(self._h, variant))

    def upload_proteome(self, aa: np.ndarray):
        aa = np.ascontiguousarray(aa, dtype=np.uint8)
        self._check(self._lib.v2p_upload_proteome(self._h, _p(aa), aa.size))

    def upload_reference(self, aa: np.ndarray, record_headers: np.ndarray):
        """Proteome + resident FASTA record headers (for Batch.add_haplotype_fasta)."""
        aa = np.ascontiguousarray(aa, dtype=np.uint8)
        hd = np.ascontiguousarray(record_headers, dtype=np.uint8)
        self._check(self._lib.v2p_upload_reference(self._h, _p(aa), aa.size, _p(hd), hd.size))

    # -- GIR-faithful mode --------------------------------------------------
    def execute_gir(self, code, start_pos, length, start_pos_res, ref: np.ndarray, alt: np.ndarray,
                    res: np.ndarray) -> np.ndarray:
        code = np.ascontiguousarray(code, dtype=np.uint8)
        sp = np.ascontiguousarray(start_pos, dtype=np.uint64)
        ln = np.ascontiguousarray(length, dtype=np.uint64)
        sr = np.ascontiguousarray(start_pos_res, dtype=np.uint64)
        assert ref.dtype == np.uint32 and alt.dtype == np.uint32 and res.dtype == np.uint32
        assert res.flags.c_contiguous and res.flags.writeable
        ref = np.ascontiguousarray(ref)
        alt = np.ascontiguousarray(alt)
        self._check(self._lib.v2p_execute_gir(self._h, _p(code), _p(sp), _p(ln), _p(sr), code.size,
                                              _p(ref), ref.size, _p(alt), alt.size, _p(res), res.size))
        return res

    def execute_gir_shared(self, code, start_pos, length, start_pos_res, ref: np.ndarray, alt: np.ndarray,
                           res: np.ndarray) -> np.ndarray:
        """v2p_execute_gir_shared: callable from many threads on this one context (ctypes releases the GIL during the call);
        concurrent calls are coalesced into one upload / launch / download.  Exec codes travel as uint64 (gir.rs:283-299)."""
        code = np.ascontiguousarray(code, dtype=np.uint64)
        sp = np.ascontiguousarray(start_pos, dtype=np.uint64)
        ln = np.ascontiguousarray(length, dtype=np.uint64)
        sr = np.ascontiguousarray(start_pos_res, dtype=np.uint64)
        assert ref.dtype == np.uint32 and alt.dtype == np.uint32 and res.dtype == np.uint32
        assert res.flags.c_contiguous and res.flags.writeable
        ref = np.ascontiguousarray(ref)
        alt = np.ascontiguousarray(alt)
        row = ctypes.c_int64(-1)
        rc = self._lib.v2p_execute_gir_shared(self._h, _p(code), _p(sp), _p(ln), _p(sr), code.size,
                                              _p(ref), ref.size, _p(alt), alt.size, _p(res), res.size, ctypes.byref(row))
        if rc != N.V2P_OK:
            raise V2PError(rc, (self._lib.v2p_last_error(self._h) or b"").decode(), int(row.value))
        return res

    def gir_submit(self, code, start_pos, length, start_pos_res, ref: np.ndarray, alt: np.ndarray, res: np.ndarray):
        """v2p_gir_submit: the first half of execute_gir_shared -- checks, narrows and stages this call's share of a batch on the calling
        thread and returns a ticket; the arrays stay referenced by the ticket until gir_collect."""
        code = np.ascontiguousarray(code, dtype=np.uint64)
        sp = np.ascontiguousarray(start_pos, dtype=np.uint64)
        ln = np.ascontiguousarray(length, dtype=np.uint64)
        sr = np.ascontiguousarray(start_pos_res, dtype=np.uint64)
        assert ref.dtype == np.uint32 and alt.dtype == np.uint32 and res.dtype == np.uint32
        assert res.flags.c_contiguous and res.flags.writeable
        ref = np.ascontiguousarray(ref)
        alt = np.ascontiguousarray(alt)
        h = ctypes.c_void_p()
        rc = self._lib.v2p_gir_submit(self._h, _p(code), _p(sp), _p(ln), _p(sr), code.size, _p(ref), ref.size, _p(alt), alt.size, _p(res), res.size, ctypes.byref(h))
        if rc == 1:                                      # V2P_BUSY: every batch of the queue is in flight -- collect a ticket, then submit again
            return None
        if rc != N.V2P_OK:
            raise V2PError(rc, (self._lib.v2p_last_error(self._h) or b"").decode(), int(self._lib.v2p_last_error_index(self._h)))
        return (h, (code, sp, ln, sr, ref, alt, res))

    def gir_collect(self, ticket) -> np.ndarray:
        """v2p_gir_collect: waits for the ticket's batch, widens the result into the `res` given to gir_submit and returns it; a task
        the reference would panic on raises here."""
        h, keep = ticket
        row = ctypes.c_int64(-1)
        rc = self._lib.v2p_gir_collect(self._h, h, ctypes.byref(row))
        if rc != N.V2P_OK:
            raise V2PError(rc, (self._lib.v2p_last_error(self._h) or b"").decode(), int(row.value))
        return keep[6]

    def coalesce_stats(self) -> Tuple[int, int]:
        """(batches launched, calls served) by execute_gir_shared on this context."""
        nb, nc = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._check(self._lib.v2p_coalesce_stats(self._h, ctypes.byref(nb), ctypes.byref(nc)))
        return int(nb.value), int(nc.value)

    def validate_gir(self, code, start_pos, length, start_pos_res, n_ref: int, n_alt: int, n_res: int) -> Tuple[int, int]:
        """DEBUG_GPU inspection: (first bad row or -1, reason status)."""
        code = np.ascontiguousarray(code, dtype=np.uint8)
        sp = np.ascontiguousarray(start_pos, dtype=np.uint64)
        ln = np.ascontiguousarray(length, dtype=np.uint64)
        sr = np.ascontiguousarray(start_pos_res, dtype=np.uint64)
        bad, reason = ctypes.c_int64(-1), ctypes.c_int(0)
        self._check(self._lib.v2p_validate_gir(self._h, _p(code), _p(sp), _p(ln), _p(sr), code.size,
                                               n_ref, n_alt, n_res, ctypes.byref(bad), ctypes.byref(reason)))
        return int(bad.value), int(reason.value)

    def batch(self) -> "Batch":
        return Batch(self)

    def upload_stream(self, stream) -> "ResidentStream":
        """v2p_stream_upload: a transcript stream (cohort.TxStream, txstream.HostTxStream: anything with a `.struct` laid out like
        v2p_txstream) made resident on this context's device -- the input of Batch.build_and_execute / build_from_stream."""
        return ResidentStream(self, stream)


class ResidentStream:
    def __init__(self, ctx: "Context", stream):
        self.ctx = ctx
        self._lib = ctx._lib
        h = ctypes.c_void_p()
        ctx._check(self._lib.v2p_stream_upload(ctx._h, ctypes.byref(stream.struct), ctypes.byref(h)))
        self._h = h

    def counts(self) -> Dict[str, int]:
        v = [ctypes.c_uint64() for _ in range(4)]
        self.ctx._check(self._lib.v2p_stream_counts(self._h, *[ctypes.byref(x) for x in v]))
        return dict(zip(("n_haps", "n_tx", "n_tasks", "out_bytes"), (int(x.value) for x in v)))

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            self._lib.v2p_stream_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """Many haplotypes executed per launch from one concatenated device image."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self._lib = ctx._lib
        h = ctypes.c_void_p()
        ctx._check(self._lib.v2p_batch_create(ctx._h, ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            self._lib.v2p_batch_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_gir(self, code, start_pos, length, start_pos_res, ref: np.ndarray, alt: np.ndarray, n_res: int):
        code = np.ascontiguousarray(code, dtype=np.uint8)
        sp = np.ascontiguousarray(start_pos, dtype=np.uint64)
        ln = np.ascontiguousarray(length, dtype=np.uint64)
        sr = np.ascontiguousarray(start_pos_res, dtype=np.uint64)
        ref = np.ascontiguousarray(ref, dtype=np.uint32)
        alt = np.ascontiguousarray(alt, dtype=np.uint32)
        self.ctx._check(self._lib.v2p_batch_add_gir(self._h, _p(code), _p(sp), _p(ln), _p(sr), code.size,
                                                    _p(ref), ref.size, _p(alt), alt.size, n_res))

    def add_haplotype(self, code, start_pos, length, start_pos_res, seg_ref_begin, seg_proteome_off,
                      alt: np.ndarray, n_res: int):
        code = np.ascontiguousarray(code, dtype=np.uint8)
        sp = np.ascontiguousarray(start_pos, dtype=np.uint64)
        ln = np.ascontiguousarray(length, dtype=np.uint64)
        sr = np.ascontiguousarray(start_pos_res, dtype=np.uint64)
        sb = np.ascontiguousarray(seg_ref_begin, dtype=np.uint64)
        so = np.ascontiguousarray(seg_proteome_off, dtype=np.uint64)
        assert sb.size == so.size + 1 or (sb.size == 0 and so.size == 0)
        alt = np.ascontiguousarray(alt, dtype=np.uint8)
        self.ctx._check(self._lib.v2p_batch_add_haplotype(self._h, _p(code), _p(sp), _p(ln), _p(sr), code.size,
                                                          _p(sb), _p(so), so.size, _p(alt), alt.size, n_res))

    def add_haplotype_fasta(self, code, start_pos, length, start_pos_res, seg_ref_begin, seg_proteome_off,
                            alt: np.ndarray, n_res: int, rec_res_end, rec_header_off, rec_header_len):
        """FASTA emit: the haplotype's arena range becomes header / residues / line feed per record."""
        code = np.ascontiguousarray(code, dtype=np.uint8)
        sp = np.ascontiguousarray(start_pos, dtype=np.uint64)
        ln = np.ascontiguousarray(length, dtype=np.uint64)
        sr = np.ascontiguousarray(start_pos_res, dtype=np.uint64)
        sb = np.ascontiguousarray(seg_ref_begin, dtype=np.uint64)
        so = np.ascontiguousarray(seg_proteome_off, dtype=np.uint64)
        alt = np.ascontiguousarray(alt, dtype=np.uint8)
        re_ = np.ascontiguousarray(rec_res_end, dtype=np.uint64)
        ho = np.ascontiguousarray(rec_header_off, dtype=np.uint64)
        hl = np.ascontiguousarray(rec_header_len, dtype=np.uint32)
        self.ctx._check(self._lib.v2p_batch_add_haplotype_fasta(
            self._h, _p(code), _p(sp), _p(ln), _p(sr), code.size, _p(sb), _p(so), so.size, _p(alt), alt.size, n_res,
            _p(re_), _p(ho), _p(hl), re_.size))

    def begin_haplotype(self):
        self.ctx._check(self._lib.v2p_batch_begin_haplotype(self._h))

    def add_transcript(self, code, start_pos, length, start_pos_res, tx_proteome_off: int, tx_ref_len: int,
                       alt: np.ndarray, res_len: int, header_off: int = 0, header_len: int = 0):
        """One TranscriptInstruction::get_g_rep result (per-transcript offsets); step 5 happens in the builder."""
        code = np.ascontiguousarray(code, dtype=np.uint8)
        sp = np.ascontiguousarray(start_pos, dtype=np.uint64)
        ln = np.ascontiguousarray(length, dtype=np.uint64)
        sr = np.ascontiguousarray(start_pos_res, dtype=np.uint64)
        alt = np.ascontiguousarray(alt, dtype=np.uint8)
        self.ctx._check(self._lib.v2p_batch_add_transcript(self._h, _p(code), _p(sp), _p(ln), _p(sr), code.size,
                                                           tx_proteome_off, tx_ref_len, _p(alt), alt.size, res_len,
                                                           header_off, header_len))

    def end_haplotype(self):
        self.ctx._check(self._lib.v2p_batch_end_haplotype(self._h))

    def set_packed(self, desc: np.ndarray, chunks: np.ndarray, payload: np.ndarray, hap_out_begin: np.ndarray):
        desc = np.ascontiguousarray(desc, dtype=np.uint64)
        chunks = np.ascontiguousarray(chunks, dtype=np.uint64).reshape(-1, 2)
        payload = np.ascontiguousarray(payload, dtype=np.uint8)
        hb = np.ascontiguousarray(hap_out_begin, dtype=np.uint64)
        self.ctx._check(self._lib.v2p_batch_set_packed(self._h, _p(desc), desc.size, _p(chunks), chunks.shape[0],
                                                       _p(payload), payload.size, hb.ctypes.data, hb.size - 1))

    def finalize(self):
        self.ctx._check(self._lib.v2p_batch_finalize(self._h))

    def build_on_device(self, stream, window_bytes: int, kernel: int = 2) -> float:
        """Step 5 and the image packing as kernels (v2p_batch_build_on_device): `stream` is a cohort.TxStream (or anything with a
        `.struct` laid out like v2p_txstream).  Returns the time of the build kernels in ms; the batch is finalized."""
        ms = ctypes.c_float(0.0)
        self.ctx._check(self._lib.v2p_batch_build_on_device(self._h, ctypes.byref(stream.struct), window_bytes, kernel, ctypes.byref(ms)))
        return float(ms.value)

    def build_from_stream(self, rs: "ResidentStream", kernel: int = 0) -> float:
        """v2p_batch_build_from_stream: the one-piece device builder on a resident stream (no H2D).  Returns the build kernels' ms."""
        ms = ctypes.c_float(0.0)
        self.ctx._check(self._lib.v2p_batch_build_from_stream(self._h, rs._h, kernel, ctypes.byref(ms)))
        self._stream_ref = rs                      # (the image's payload descriptors read the stream's alt bytes)
        return float(ms.value)

    def build_and_execute(self, rs: "ResidentStream", kernel: int = 0, n_slices: int = 0):
        """v2p_batch_build_and_execute: Task vectors -> result bytes in one call, the image built slice by slice while the slice
        before it is stitched.  Asynchronous like execute(); sync() collects the status."""
        self.ctx._check(self._lib.v2p_batch_build_and_execute(self._h, rs._h, kernel, n_slices))
        self._stream_ref = rs

    def image_form(self) -> dict:
        """v2p_batch_image_form: how the image sits on the device right now."""
        f = self._lib.v2p_batch_image_form(self._h)
        if f < 0:
            self.ctx._check(f)
        return {"padded": bool(f & 1), "pieces": bool(f & 2), "staging_buffers": bool(f & 4), "tiles": bool(f & 8)}

    def oneshot_info(self) -> dict:
        info = N.OneShotInfo()
        self.ctx._check(self._lib.v2p_batch_oneshot_info(self._h, ctypes.byref(info)))
        return {"kernel": int(info.kernel), "n_slices": int(info.n_slices), "total_ms": float(info.total_ms), "build_ms": float(info.build_ms),
                "call_wall_ms": float(info.call_wall_ms), "tables_ms": float(info.tables_ms), "slice_build_ms": [float(info.slice_build_ms[j]) for j in range(int(info.n_slices))]}

    def reset(self):
        """v2p_batch_reset: back to empty, device buffers kept (the next build recycles them)."""
        self.ctx._check(self._lib.v2p_batch_reset(self._h))

    def download_patch_image(self):
        """A PATCH image (kernel 8) as it sits on the device: (segments [n_chunks, 1024] u64, patches [n_chunks, 1024] u32, chunk table in launch
        order [n_chunks, 2] u64, total segments, total patches) -- chunk k = arena offset / 8192 owns row k of both arrays."""
        n = self.counts()["n_chunks"]
        seg = np.zeros((n, 1024), dtype=np.uint64)
        patch = np.zeros((n, 1024), dtype=np.uint32)
        chunks = np.zeros((n, 2), dtype=np.uint64)
        ns, npat = ctypes.c_uint64(), ctypes.c_uint64()
        self.ctx._check(self._lib.v2p_batch_download_patch_image(self._h, seg.ctypes.data if n else None, patch.ctypes.data if n else None,
                                                                 chunks.ctypes.data if n else None, ctypes.byref(ns), ctypes.byref(npat)))
        return seg, patch, chunks, int(ns.value), int(npat.value)

    def download_image(self):
        """(desc, chunks, hap_out_begin) as they sit on the device."""
        cn = self.counts()
        n_haps, n_desc, n_chunks = cn["n_haps"], cn["n_desc"], cn["n_chunks"]
        desc = np.zeros(n_desc, dtype=np.uint64)
        chunks = np.zeros((n_chunks, 2), dtype=np.uint64)
        hb = np.zeros(n_haps + 1, dtype=np.uint64)
        self.ctx._check(self._lib.v2p_batch_download_image(self._h, desc.ctypes.data if n_desc else None,
                                                           chunks.ctypes.data if n_chunks else None, hb.ctypes.data))
        return desc, chunks, hb

    def execute(self):
        self.ctx._check(self._lib.v2p_batch_execute(self._h))

    def sync(self):
        self.ctx._check(self._lib.v2p_batch_sync(self._h))

    def counts(self) -> Dict[str, int]:
        v = [ctypes.c_uint64() for _ in range(5)]
        self.ctx._check(self._lib.v2p_batch_counts(self._h, *[ctypes.byref(x) for x in v]))
        return dict(zip(("n_haps", "n_desc", "n_chunks", "out_bytes", "payload_bytes"), (int(x.value) for x in v)))

    def hap_range(self, h: int) -> Tuple[int, int]:
        b, ln = ctypes.c_uint64(), ctypes.c_uint64()
        self.ctx._check(self._lib.v2p_batch_hap_range(self._h, h, ctypes.byref(b), ctypes.byref(ln)))
        return int(b.value), int(ln.value)

    def download(self, begin: int, length: int) -> np.ndarray:
        out = np.empty(length, dtype=np.uint8)
        self.ctx._check(self._lib.v2p_batch_download(self._h, begin, length, _p(out)))
        return out

    def download_hap(self, h: int) -> np.ndarray:
        return self.download(*self.hap_range(h))

    def digests(self) -> np.ndarray:
        n = self.counts()["n_haps"]
        out = np.zeros(n, dtype=np.uint64)
        self.ctx._check(self._lib.v2p_batch_digests(self._h, _p(out), n))
        return out

    def device_out(self) -> int:
        return int(self._lib.v2p_batch_device_out(self._h) or 0)

    def scribble(self, byte: int = 0xEE):
        """v2p_batch_scribble: the whole arena overwritten (checkers call it before every re-execute they verify)."""
        self.ctx._check(self._lib.v2p_batch_scribble(self._h, byte))


_default_ctx: Optional[Context] = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None or _default_ctx._h is None:
        _default_ctx = Context(0)
    return _default_ctx


class GIR:
    """gir.rs:15-23.  Tapes are sequences of characters, like the reference's Vec<char>."""

    def __init__(self, g_rep: Sequence[Task], annotation: Dict[str, Tuple[int, int]], alt_stream: Sequence[str],
                 ref_stream: Sequence[str], res_array: Sequence[str]):
        self.g_rep = list(g_rep)
        self.annotation = dict(annotation)
        self.alt_stream = list(alt_stream)
        self.ref_stream = list(ref_stream)
        self.res_array = list(res_array)

    def get_tasks(self) -> List[Task]:
        return self.g_rep

    def get_annotation(self):
        return self.annotation

    def get_results_max(self) -> int:  # gir.rs:156-167
        return max((v[1] for v in self.annotation.values()), default=0)

    def execute(self, engine: Engine, ctx: Optional[Context] = None):
        """gir.rs:197-241 -> (res_array, annotation).  Only Engine.GPU lives in this package."""
        if engine is not Engine.GPU:
            raise NotImplementedError("the st/mt engines are the reference's own CPU code; this package is the gpu engine")
        ctx = ctx or default_context()
        code, sp, ln, sr = _soa(self.g_rep)
        ref = np.fromiter((ord(c) for c in self.ref_stream), dtype=np.uint32, count=len(self.ref_stream))
        alt = np.fromiter((ord(c) for c in self.alt_stream), dtype=np.uint32, count=len(self.alt_stream))
        res = np.fromiter((ord(c) for c in self.res_array), dtype=np.uint32, count=len(self.res_array))
        ctx.execute_gir(code, sp, ln, sr, ref, alt, res)
        return [chr(int(c)) for c in res], self.annotation


class Pipeline:
    """Streamed execution with H2D / kernel / D2H overlap (v2p_pipeline_*): the way results reach the host.  submit_stream: slices of the
    transcript stream (Task vectors in, host bytes out, nothing packed on the host); submit: host-packed images."""

    def __init__(self, ctx: Context, n_slots: int = 3):
        self.ctx = ctx
        self._lib = ctx._lib
        h = ctypes.c_void_p()
        ctx._check(self._lib.v2p_pipeline_create(ctx._h, n_slots, ctypes.byref(h)))
        self._h = h
        self.n_slots = n_slots

    def close(self):
        if getattr(self, "_h", None) and getattr(self.ctx, "_h", None):
            self._lib.v2p_pipeline_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, desc: np.ndarray, chunks: np.ndarray, payload: np.ndarray, out_bytes: int) -> int:
        desc = np.ascontiguousarray(desc, dtype=np.uint64)
        chunks = np.ascontiguousarray(chunks, dtype=np.uint64).reshape(-1, 2)
        payload = np.ascontiguousarray(payload, dtype=np.uint8)
        t = ctypes.c_uint32()
        self.ctx._check(self._lib.v2p_pipeline_submit(self._h, _p(desc), desc.size, _p(chunks), chunks.shape[0],
                                                      _p(payload), payload.size, out_bytes, ctypes.byref(t)))
        return int(t.value)

    def reserve(self, stream_bytes: int, out_bytes: int, copy_threads: int = 0):
        """v2p_pipeline_reserve: pin the slots' staging / result buffers now; how many threads copy a slice into its staging."""
        self.ctx._check(self._lib.v2p_pipeline_reserve(self._h, stream_bytes, out_bytes, copy_threads))

    def submit_stream(self, stream, kernel: int = 0, digests: bool = False) -> int:
        """v2p_pipeline_submit_stream: a slice of the transcript stream (anything with a `.struct` laid out like v2p_txstream) -- checked and
        staged on this thread, uploaded, built + executed by the pipeline's runner, its arena copied back.  The slice may be freed on return.
        Returns the ticket, or -1 when every slot is in use (V2P_BUSY: wait for a ticket, release it, submit again)."""
        t = ctypes.c_uint32()
        rc = self._lib.v2p_pipeline_submit_stream(self._h, ctypes.byref(stream.struct), kernel, 1 if digests else 0, ctypes.byref(t))
        if rc == 1:                                          # V2P_BUSY: every slot is in use
            return -1
        self.ctx._check(rc)
        return int(t.value)

    def result_info(self, ticket: int) -> dict:
        """v2p_pipeline_result_info of a stream slice that has been waited for: hap_out_begin (copy), digests (copy or None), times."""
        hb, dg, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint64()
        times = (ctypes.c_double * 2)()
        self.ctx._check(self._lib.v2p_pipeline_result_info(self._h, ticket, ctypes.byref(hb), ctypes.byref(n), ctypes.byref(dg), ctypes.byref(times)))
        nh = int(n.value)
        hob = np.ctypeslib.as_array(ctypes.cast(hb, ctypes.POINTER(ctypes.c_uint64)), shape=(nh + 1,)).copy()
        dig = np.ctypeslib.as_array(ctypes.cast(dg, ctypes.POINTER(ctypes.c_uint64)), shape=(nh,)).copy() if dg.value and nh else (np.zeros(0, np.uint64) if dg.value else None)
        return {"hap_out_begin": hob, "digests": dig, "stage_ms": float(times[0]), "runner_ms": float(times[1])}

    def wait(self, ticket: int) -> np.ndarray:
        """View of the slot's pinned result buffer (valid until release(ticket))."""
        ptr, n = ctypes.c_void_p(), ctypes.c_uint64()
        self.ctx._check(self._lib.v2p_pipeline_wait(self._h, ticket, ctypes.byref(ptr), ctypes.byref(n)))
        if n.value == 0:
            return np.zeros(0, dtype=np.uint8)
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(int(n.value),))

    def release(self, ticket: int):
        self.ctx._check(self._lib.v2p_pipeline_release(self._h, ticket))
