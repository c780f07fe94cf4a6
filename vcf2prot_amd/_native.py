"""ctypes bindings of the C ABI (include/vcf2prot_hip.h, include/v2p_cohort.h).

The HIP library is the product: if it is missing this module raises -- there is
no Python or CPU fallback for the engine.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_size_t, c_uint8, c_uint32, c_uint64, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_PKG, "lib", "libvcf2prot_hip.so")
COHORT_LIB_PATH = os.path.join(_PKG, "lib", "libv2p_cohort.so")

V2P_OK = 0
ERR_NAMES = {
    0: "V2P_OK", -1: "V2P_ERR_INVALID_ARG", -2: "V2P_ERR_HIP", -3: "V2P_ERR_BAD_CODE", -4: "V2P_ERR_RES_OOB",
    -5: "V2P_ERR_SRC_OOB", -6: "V2P_ERR_NOT_CONTIGUOUS", -7: "V2P_ERR_NOT_CANONICAL", -8: "V2P_ERR_NON_BYTE_CHAR",
    -9: "V2P_ERR_UNSUPPORTED", -10: "V2P_ERR_STATE",
}
V2P_ERR_INVALID_ARG, V2P_ERR_HIP, V2P_ERR_BAD_CODE, V2P_ERR_RES_OOB, V2P_ERR_SRC_OOB = -1, -2, -3, -4, -5
V2P_ERR_NOT_CONTIGUOUS, V2P_ERR_NOT_CANONICAL, V2P_ERR_NON_BYTE_CHAR, V2P_ERR_UNSUPPORTED, V2P_ERR_STATE = -6, -7, -8, -9, -10
V2P_FLAG_DEBUG_GPU, V2P_FLAG_TEMPORAL, V2P_FLAG_RESULT_ORDER = 1, 2, 4


class v2p_chunk(ctypes.Structure):
    _fields_ = [("task_begin", c_uint64), ("dst_n", c_uint64)]


# name -> (restype, argtypes); every symbol include/vcf2prot_hip.h declares
HIP_API = {
    "v2p_version": (c_char_p, []),
    "v2p_device_count": (c_int, []),
    "v2p_engine_from_str": (c_int, [c_char_p, POINTER(c_int)]),
    "v2p_init": (c_int, [c_int, ctypes.c_uint, POINTER(c_void_p)]),
    "v2p_destroy": (None, [c_void_p]),
    "v2p_last_error": (c_char_p, [c_void_p]),
    "v2p_last_error_index": (c_int64, [c_void_p]),
    "v2p_set_stream": (c_int, [c_void_p, c_void_p]),
    "v2p_execute_gir": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64,
                                c_void_p, c_uint64, c_void_p, c_uint64, c_void_p, c_uint64]),
    "v2p_execute_gir_shared": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64,
                                       c_void_p, c_uint64, c_void_p, c_uint64, c_void_p, c_uint64, POINTER(c_int64)]),
    "v2p_gir_submit": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_uint64, c_void_p, c_uint64, c_void_p, c_uint64, POINTER(c_void_p)]),
    "v2p_gir_collect": (c_int, [c_void_p, c_void_p, POINTER(ctypes.c_int64)]),
    "v2p_coalesce_stats": (c_int, [c_void_p, POINTER(c_uint64), POINTER(c_uint64)]),
    "v2p_validate_gir": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64,
                                 c_uint64, c_uint64, c_uint64, POINTER(c_int64), POINTER(c_int)]),
    "v2p_upload_proteome": (c_int, [c_void_p, c_void_p, c_uint64]),
    "v2p_upload_reference": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_uint64]),
    "v2p_batch_add_haplotype_fasta": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64,
                                              c_void_p, c_void_p, c_uint64, c_void_p, c_uint64, c_uint64,
                                              c_void_p, c_void_p, c_void_p, c_uint64]),
    "v2p_batch_begin_haplotype": (c_int, [c_void_p]),
    "v2p_batch_add_transcript": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64, c_uint64, c_uint64,
                                         c_void_p, c_uint64, c_uint64, c_uint64, c_uint32]),
    "v2p_batch_end_haplotype": (c_int, [c_void_p]),
    "v2p_batch_create": (c_int, [c_void_p, POINTER(c_void_p)]),
    "v2p_batch_destroy": (None, [c_void_p]),
    "v2p_batch_add_gir": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64,
                                  c_void_p, c_uint64, c_void_p, c_uint64, c_uint64]),
    "v2p_batch_add_haplotype": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64,
                                        c_void_p, c_void_p, c_uint64, c_void_p, c_uint64, c_uint64]),
    "v2p_batch_build_on_device": (c_int, [c_void_p, c_void_p, c_uint32, c_int, POINTER(ctypes.c_float)]),
    "v2p_batch_download_image": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "v2p_stream_upload": (c_int, [c_void_p, c_void_p, POINTER(c_void_p)]),
    "v2p_stream_destroy": (None, [c_void_p]),
    "v2p_stream_counts": (c_int, [c_void_p, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64)]),
    "v2p_batch_build_from_stream": (c_int, [c_void_p, c_void_p, c_int, POINTER(ctypes.c_float)]),
    "v2p_batch_build_and_execute": (c_int, [c_void_p, c_void_p, c_int, c_uint32]),
    "v2p_batch_oneshot_info": (c_int, [c_void_p, c_void_p]),
    "v2p_batch_reset": (c_int, [c_void_p]),
    "v2p_batch_set_packed": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_uint64, c_void_p, c_uint64,
                                     c_void_p, c_uint64]),
    "v2p_batch_finalize": (c_int, [c_void_p]),
    "v2p_batch_execute": (c_int, [c_void_p]),
    "v2p_batch_sync": (c_int, [c_void_p]),
    "v2p_batch_image_form": (c_int, [c_void_p]),
    "v2p_batch_counts": (c_int, [c_void_p, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64),
                                 POINTER(c_uint64), POINTER(c_uint64)]),
    "v2p_batch_hap_range": (c_int, [c_void_p, c_uint64, POINTER(c_uint64), POINTER(c_uint64)]),
    "v2p_batch_download": (c_int, [c_void_p, c_uint64, c_uint64, c_void_p]),
    "v2p_batch_digests": (c_int, [c_void_p, c_void_p, c_uint64]),
    "v2p_batch_device_out": (c_void_p, [c_void_p]),
    "v2p_batch_scribble": (c_int, [c_void_p, c_int]),
    "v2p_pipeline_create": (c_int, [c_void_p, c_uint32, POINTER(c_void_p)]),
    "v2p_pipeline_destroy": (None, [c_void_p]),
    "v2p_pipeline_submit": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_uint64, c_void_p, c_uint64, c_uint64,
                                    POINTER(c_uint32)]),
    "v2p_pipeline_wait": (c_int, [c_void_p, c_uint32, POINTER(c_void_p), POINTER(c_uint64)]),
    "v2p_pipeline_reserve": (c_int, [c_void_p, c_uint64, c_uint64, c_uint32]),
    "v2p_pipeline_submit_stream": (c_int, [c_void_p, c_void_p, c_int, ctypes.c_uint, POINTER(c_uint32)]),
    "v2p_pipeline_result_info": (c_int, [c_void_p, c_uint32, POINTER(c_void_p), POINTER(c_uint64), POINTER(c_void_p), c_void_p]),
    "v2p_pipeline_release": (c_int, [c_void_p, c_uint32]),
    "v2p_stitch_launch_opts": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_uint32, c_void_p, c_uint64, c_void_p, c_uint64,
                                       c_void_p, c_uint64, c_void_p, c_void_p]),
    "v2p_set_launch_opts": (c_int, [c_void_p, c_void_p]),
    "v2p_stitch_launch_bits": (c_int, [c_void_p, c_uint64]),
    "v2p_routing_rules": (c_int, [c_uint64, c_uint64, c_uint64, c_uint64, c_int, c_void_p]),
    "v2p_order_chunks_for_xcds": (c_int, [c_void_p, c_uint64, c_void_p, c_uint64, c_uint64]),
    "v2p_digest_launch": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_uint64, c_void_p]),
}

_hip = None
_cohort = None


class OneShotInfo(ctypes.Structure):
    """v2p_oneshot_info (include/vcf2prot_hip.h)"""
    _fields_ = [("kernel", ctypes.c_int32), ("n_slices", ctypes.c_uint32), ("total_ms", ctypes.c_float), ("build_ms", ctypes.c_float),
                ("call_wall_ms", ctypes.c_double), ("slice_build_ms", ctypes.c_float * 32), ("tables_ms", ctypes.c_float)]


class Routing(ctypes.Structure):
    """v2p_routing (include/vcf2prot_hip.h)"""
    _fields_ = [("wave_bytes_per_task", ctypes.c_uint32), ("rich", ctypes.c_uint32), ("phased", ctypes.c_uint32), ("store_sc1", ctypes.c_uint32),
                ("phase_bytes", ctypes.c_uint64), ("order_blocks", ctypes.c_uint32), ("order_windows", ctypes.c_uint32)]


def routing_rules(n_desc: int, n_chunks: int, result_bytes: int, proteome_len: int, wave_image: bool = True) -> dict:
    """v2p_routing_rules: the library's routing rules for an image of these sizes (host-side, no GPU work)."""
    r = Routing()
    rc = hip_lib().v2p_routing_rules(n_desc, n_chunks, result_bytes, proteome_len, 1 if wave_image else 0, ctypes.byref(r))
    if rc != 0:
        raise V2PError(rc, "v2p_routing_rules")
    return {k: int(getattr(r, k)) for k, _ in Routing._fields_ if k != "reserved"}


class LaunchOpts(ctypes.Structure):
    """v2p_launch_opts (include/vcf2prot_hip.h): zero = the library's choice (store_sc1: -1)."""
    _fields_ = [("nontemporal", c_uint32), ("routing", c_uint32), ("phase_bytes", c_uint64), ("phase_min_chunks", c_uint32),
                ("store_sc1", ctypes.c_int32), ("max_blocks", c_uint32), ("reserved", c_uint32)]

    def __init__(self, nontemporal=1, routing=0, phase_bytes=0, phase_min_chunks=0, store_sc1=-1, max_blocks=0, reserved=0):
        super().__init__(nontemporal, routing, phase_bytes, phase_min_chunks, store_sc1, max_blocks, reserved)       # reserved: 0 (libv2p_bench.so: 3 / 8, csrc/bench/v2p_bench.h)


def stitch_launch(lib, stream, d_desc, n_desc, d_chunks, n_chunks, d_src0, src0_len, d_src1, src1_len, d_out, out_len, d_status, opts: "LaunchOpts") -> int:
    """v2p_stitch_launch_opts on raw device pointers (ints)."""
    return int(lib.v2p_stitch_launch_opts(ctypes.c_void_p(stream) if stream else None, d_desc, n_desc, d_chunks, n_chunks, d_src0, src0_len, d_src1, src1_len,
                                          d_out, out_len, d_status, ctypes.byref(opts)))


def _bind(lib, api):
    for name, (res, args) in api.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def hip_lib():
    """The HIP engine library.  Raises if it has not been built (no fallback)."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise RuntimeError(
                f"{HIP_LIB_PATH} is missing: build it with `python -m vcf2prot_amd.build` "
                "(the gpu engine has no CPU fallback)")
        _hip = _bind(ctypes.CDLL(HIP_LIB_PATH), HIP_API)
    return _hip


BENCH_LIB_PATH = os.path.join(_PKG, "lib", "libv2p_bench.so")
BENCH_API = {
    "v2p_copy_prefetch_launch": (c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_void_p, c_uint64, c_int, c_int, c_uint32]),
    "v2p_copy_mix_launch": (c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_void_p, c_uint64, c_void_p, c_uint32, c_uint32, c_uint32, c_uint32]),
    "v2p_copy_bench_launch": (c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_void_p, c_uint64, c_int, c_void_p]),
    "v2p_gather_bench_launch": (c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_uint32, c_uint32, c_void_p]),
    "v2p_fill_launch": (c_int, [c_void_p, c_void_p, c_uint64, c_uint32, c_int]),
    # the launcher with the packed flag word (kernel variants, timing-only ablations): csrc/bench/v2p_bench.h
    "v2p_stitch_launch": (c_int, [c_void_p, c_void_p, c_uint64, c_void_p, c_uint32, c_void_p, c_uint64, c_void_p, c_uint64,
                                  c_void_p, c_uint64, c_void_p, c_int, c_uint32]),
    # what no routing rule of the product picks (V2P_BENCH_VARIANTS build of the engine)
    "v2p_batch_download_patch_image": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_uint64), POINTER(c_uint64)]),
    "v2p_bench_set_variant": (c_int, [c_void_p, c_uint32]),
}
_bench = None


def bench_lib():
    """libv2p_bench.so (development tools): the micro-benchmarks of csrc/bench/ plus a V2P_BENCH_VARIANTS build of the engine, whose
    v2p_stitch_launch takes the timing-only ablation bits.  Binds the engine's API too, so a tool can use it in place of hip_lib()."""
    global _bench
    if _bench is None:
        if not os.path.exists(BENCH_LIB_PATH):
            from . import build
            build.build_bench()
        _bench = _bind(_bind(ctypes.CDLL(BENCH_LIB_PATH), HIP_API), BENCH_API)
    return _bench


def cohort_lib():
    global _cohort
    if _cohort is None:
        if not os.path.exists(COHORT_LIB_PATH):
            raise RuntimeError(f"{COHORT_LIB_PATH} is missing: build it with `python -m vcf2prot_amd.build`")
        from ._cohort_api import COHORT_API  # noqa: WPS433
        _cohort = _bind(ctypes.CDLL(COHORT_LIB_PATH), COHORT_API)
    return _cohort


class V2PError(RuntimeError):
    """A non-zero status from the C ABI; stands for the reference's panic!()."""

    def __init__(self, code: int, message: str, index: int = -1):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {message}")
        self.code = code
        self.index = index
