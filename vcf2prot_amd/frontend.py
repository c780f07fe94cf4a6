"""Host mirror of the VCF front-end pieces of include/v2p_frontend.h.

    idx    = VcfIndex(vcf_bytes)                     # readers.rs:151-231 + vcf_ds.rs:67-87, linear time, host
    lists  = decode_bitmasks(ctx, idx)               # VCFRecords::get_csq_per_patient on the GPU (vcf_ds.rs:192-329)
    groups = group_per_transcript(idx, lists)        # vcf_tools.rs:82-96 + vcf_ds.rs:387-420, host

The decode has no CPU path: without the HIP library / a GPU it raises.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_uint8, c_uint16, c_uint32, c_uint64, c_void_p

import numpy as np

from . import _native as N

V2P_ERR_MASK_NEGATIVE, V2P_ERR_MASK_PARSE, V2P_ERR_MASK_INDEX, V2P_ERR_COLUMNS = -20, -21, -22, -23
V2P_ERR_FIELD_TOO_LONG, V2P_ERR_CAPACITY, V2P_ERR_VCF_FORMAT, V2P_ERR_DUPLICATE_POS = -24, -25, -26, -27
N.ERR_NAMES.update({-20: "V2P_ERR_MASK_NEGATIVE", -21: "V2P_ERR_MASK_PARSE", -22: "V2P_ERR_MASK_INDEX", -23: "V2P_ERR_COLUMNS",
                    -24: "V2P_ERR_FIELD_TOO_LONG", -25: "V2P_ERR_CAPACITY", -26: "V2P_ERR_VCF_FORMAT", -27: "V2P_ERR_DUPLICATE_POS"})


class v2p_mutation(ctypes.Structure):
    _fields_ = [("transcript", c_uint32), ("ref_aa_position", c_uint16), ("mut_aa_position", c_uint16),
                ("type", c_uint8), ("valid", c_uint8), ("pad_", c_uint8 * 2)]


# symbols of include/v2p_frontend.h that live in libvcf2prot_hip.so
DECODE_API = {
    "v2p_decode_run": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_uint64, c_uint64, c_void_p, c_void_p, POINTER(c_void_p)]),
    "v2p_decode_counts": (c_int, [c_void_p, c_void_p]),
    "v2p_decode_download": (c_int, [c_void_p, c_void_p]),
    "v2p_decode_device": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p)]),
    "v2p_decode_timing": (c_int, [c_void_p, POINTER(c_float), POINTER(c_float), POINTER(c_float), POINTER(c_float)]),
    "v2p_decode_destroy": (None, [c_void_p]),
    "v2p_decode_workspace_bytes": (c_uint64, [c_uint64, c_uint64, c_uint64]),
    "v2p_decode_launch": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_uint64, c_uint64, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_uint64, c_void_p, c_void_p, c_uint64, c_void_p, ctypes.c_uint]),
}
# ... and in libv2p_cohort.so (plain C++)
HOST_API = {
    "v2p_vcf_index_build": (c_int, [c_void_p, c_uint64, POINTER(c_void_p)]),
    "v2p_vcf_index_destroy": (None, [c_void_p]),
    "v2p_vcf_index_error": (c_char_p, [c_void_p]),
    "v2p_vcf_index_n_samples": (c_uint64, [c_void_p]),
    "v2p_vcf_index_n_records": (c_uint64, [c_void_p]),
    "v2p_vcf_index_n_consequences": (c_uint64, [c_void_p]),
    "v2p_vcf_index_sample": (c_int, [c_void_p, c_uint64, POINTER(c_uint64), POINTER(c_uint64)]),
    "v2p_vcf_index_row_begin": (POINTER(c_uint64), [c_void_p]),
    "v2p_vcf_index_row_end": (POINTER(c_uint64), [c_void_p]),
    "v2p_vcf_index_csq_begin": (POINTER(c_uint32), [c_void_p]),
    "v2p_vcf_index_csq_supported": (POINTER(c_uint8), [c_void_p]),
    "v2p_vcf_index_csq_text_begin": (POINTER(c_uint64), [c_void_p]),
    "v2p_vcf_index_csq_text_len": (POINTER(c_uint32), [c_void_p]),
    "v2p_groups_build": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_uint32, POINTER(c_void_p)]),
    "v2p_groups_destroy": (None, [c_void_p]),
    "v2p_groups_error": (c_char_p, [c_void_p]),
    "v2p_groups_error_haplotype": (c_int64, [c_void_p]),
    "v2p_groups_n_transcripts": (c_uint64, [c_void_p]),
    "v2p_groups_transcript": (c_int, [c_void_p, c_uint64, POINTER(c_uint64), POINTER(c_uint64)]),
    "v2p_groups_mutations": (POINTER(v2p_mutation), [c_void_p]),
    "v2p_groups_hap_group_begin": (POINTER(c_uint64), [c_void_p]),
    "v2p_groups_group_transcript": (POINTER(c_uint32), [c_void_p]),
    "v2p_groups_group_member_begin": (POINTER(c_uint64), [c_void_p]),
    "v2p_groups_member_ids": (POINTER(c_uint32), [c_void_p]),
}

_bound = {}


def _hip():
    lib = N.hip_lib()
    if "hip" not in _bound:
        N._bind(lib, DECODE_API)
        _bound["hip"] = True
    return lib


def _host():
    lib = N.cohort_lib()
    if "host" not in _bound:
        N._bind(lib, HOST_API)
        _bound["host"] = True
    return lib


def _arr(ptr, n, dtype):
    """A copy of a library-owned array (the handle may be closed while the array is still in use)."""
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).copy()


class VcfIndex:
    """Supported records, their sample-column ranges and the consequence table of one VCF text."""

    def __init__(self, text: bytes):
        self.text = np.frombuffer(text, dtype=np.uint8)
        self._bytes = text
        self._lib = _host()
        h = c_void_p()
        rc = self._lib.v2p_vcf_index_build(self.text.ctypes.data, self.text.size, ctypes.byref(h))
        self._h = h
        if rc != 0:
            msg = self._lib.v2p_vcf_index_error(h).decode() if h else "index build failed"
            self.close()
            raise N.V2PError(rc, msg)
        L = self._lib
        self.n_samples = int(L.v2p_vcf_index_n_samples(h))
        self.n_records = int(L.v2p_vcf_index_n_records(h))
        self.n_consequences = int(L.v2p_vcf_index_n_consequences(h))
        self.row_begin = _arr(L.v2p_vcf_index_row_begin(h), self.n_records, np.uint64)
        self.row_end = _arr(L.v2p_vcf_index_row_end(h), self.n_records, np.uint64)
        self.csq_begin = _arr(L.v2p_vcf_index_csq_begin(h), self.n_records + 1, np.uint32)
        self.csq_supported = _arr(L.v2p_vcf_index_csq_supported(h), self.n_consequences, np.uint8)
        self.csq_text_begin = _arr(L.v2p_vcf_index_csq_text_begin(h), self.n_consequences, np.uint64)
        self.csq_text_len = _arr(L.v2p_vcf_index_csq_text_len(h), self.n_consequences, np.uint32)

    def sample_names(self):
        out, b, n = [], c_uint64(), c_uint64()
        for i in range(self.n_samples):
            self._lib.v2p_vcf_index_sample(self._h, i, ctypes.byref(b), ctypes.byref(n))
            out.append(self._bytes[b.value:b.value + n.value].decode())
        return out

    def consequence(self, i: int) -> str:
        b = int(self.csq_text_begin[i])
        return self._bytes[b:b + int(self.csq_text_len[i])].decode()

    def record_of(self, csq_id: int) -> int:
        return int(np.searchsorted(self.csq_begin, csq_id, side="right") - 1)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.v2p_vcf_index_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


class HaplotypeLists:
    """Result of the decode: list 2*s + (h-1) holds the consequence ids of haplotype h of sample s."""

    def __init__(self, hap_begin: np.ndarray, ids: np.ndarray, timing_ms=None):
        self.hap_begin, self.ids, self.timing_ms = hap_begin, ids, timing_ms

    @property
    def n_haplotypes(self):
        return self.hap_begin.size - 1

    def of(self, h: int) -> np.ndarray:
        return self.ids[int(self.hap_begin[h]):int(self.hap_begin[h + 1])]


def decode_bitmasks(ctx, idx: VcfIndex) -> HaplotypeLists:
    """VCFRecords::get_csq_per_patient (vcf_ds.rs:192-211) for every proband, on the GPU of `ctx` (engine.Context)."""
    lib = _hip()
    h = c_void_p()
    rc = lib.v2p_decode_run(ctx._h, idx.text.ctypes.data, idx.text.size, idx.row_begin.ctypes.data, idx.row_end.ctypes.data,
                            idx.n_records, idx.n_samples, idx.csq_begin.ctypes.data, idx.csq_supported.ctypes.data, ctypes.byref(h))
    if rc != 0:
        raise N.V2PError(rc, lib.v2p_last_error(ctx._h).decode(), int(lib.v2p_last_error_index(ctx._h)))
    try:
        hap_begin = np.zeros(2 * idx.n_samples + 1, dtype=np.uint64)
        lib.v2p_decode_counts(h, hap_begin.ctypes.data)
        ids = np.zeros(int(hap_begin[-1]), dtype=np.uint32)
        rc = lib.v2p_decode_download(h, ids.ctypes.data if ids.size else None)
        if rc != 0:
            raise N.V2PError(rc, lib.v2p_last_error(ctx._h).decode())
        t = [c_float() for _ in range(4)]
        lib.v2p_decode_timing(h, *[ctypes.byref(x) for x in t])
        return HaplotypeLists(hap_begin, ids, dict(zip(("parse", "count", "scan", "emit"), (x.value for x in t))))
    finally:
        lib.v2p_decode_destroy(h)


class Groups:
    """IntMap in id space (Map.rs:5-31): per haplotype the AltTranscripts, members as consequence ids."""

    def __init__(self, idx: VcfIndex, lists: HaplotypeLists, n_threads: int = 0):
        self._lib = _host()
        self._idx = idx
        h = c_void_p()
        rc = self._lib.v2p_groups_build(idx._h, idx.text.ctypes.data, lists.hap_begin.ctypes.data,
                                        lists.ids.ctypes.data if lists.ids.size else None, lists.n_haplotypes, n_threads, ctypes.byref(h))
        self._h = h
        if rc != 0:
            msg = self._lib.v2p_groups_error(h).decode() if h else "grouping failed"
            hap = int(self._lib.v2p_groups_error_haplotype(h)) if h else -1
            self.close()
            raise N.V2PError(rc, msg, hap)
        L = self._lib
        self.n_transcripts = int(L.v2p_groups_n_transcripts(h))
        self.hap_group_begin = _arr(L.v2p_groups_hap_group_begin(h), lists.n_haplotypes + 1, np.uint64)
        n_groups = int(self.hap_group_begin[-1])
        self.group_transcript = _arr(L.v2p_groups_group_transcript(h), n_groups, np.uint32)
        self.group_member_begin = _arr(L.v2p_groups_group_member_begin(h), n_groups + 1, np.uint64)
        self.member_ids = _arr(L.v2p_groups_member_ids(h), int(self.group_member_begin[-1]), np.uint32)
        m = L.v2p_groups_mutations(h)
        self.mutations = np.ctypeslib.as_array(ctypes.cast(m, POINTER(c_uint8)), shape=(idx.n_consequences * ctypes.sizeof(v2p_mutation),)).view(
            np.dtype([("transcript", "<u4"), ("ref_aa_position", "<u2"), ("mut_aa_position", "<u2"), ("type", "u1"), ("valid", "u1"), ("pad", "u1", 2)])).copy() \
            if idx.n_consequences else None

    def transcript_name(self, rank: int) -> str:
        b, n = c_uint64(), c_uint64()
        self._lib.v2p_groups_transcript(self._h, rank, ctypes.byref(b), ctypes.byref(n))
        return self._idx._bytes[b.value:b.value + n.value].decode()

    def of(self, hap: int):
        """[(transcript name, [consequence ids])] of one haplotype, in the reference's order."""
        out = []
        for k in range(int(self.hap_group_begin[hap]), int(self.hap_group_begin[hap + 1])):
            a, b = int(self.group_member_begin[k]), int(self.group_member_begin[k + 1])
            out.append((self.transcript_name(int(self.group_transcript[k])), self.member_ids[a:b].tolist()))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.v2p_groups_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()


def group_per_transcript(idx: VcfIndex, lists: HaplotypeLists, n_threads: int = 0) -> Groups:
    return Groups(idx, lists, n_threads)
