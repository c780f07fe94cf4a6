"""ctypes view of include/v2p_cohort.h."""
import ctypes
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_uint8, c_uint32, c_uint64, c_void_p

from ._native import v2p_chunk

ALT_KINDS = 6


class CohortParams(ctypes.Structure):
    _fields_ = [("seed_proteome", c_uint64), ("seed_cohort", c_uint64), ("n_samples", c_uint32),
                ("n_transcripts", c_uint32), ("mean_len", c_double), ("len_model", c_uint32), ("fixed_len", c_uint32),
                ("altered_per_hap", c_uint32), ("alts_fixed", c_uint32), ("alts_poisson", c_double),
                ("mix", c_double * ALT_KINDS), ("max_ins", c_uint32), ("max_del", c_uint32),
                ("max_fs_tail", c_uint32), ("max_sl_ext", c_uint32), ("p_start_lost", c_double),
                ("p_empty_hap", c_double)]


class HapView(ctypes.Structure):
    _fields_ = [("n_tasks", c_uint64), ("code", POINTER(c_uint8)), ("start_pos", POINTER(c_uint64)),
                ("length", POINTER(c_uint64)), ("start_pos_res", POINTER(c_uint64)),
                ("n_alt", c_uint64), ("alt", POINTER(c_uint8)), ("n_res", c_uint64), ("n_ref", c_uint64),
                ("n_seg", c_uint64), ("seg_ref_begin", POINTER(c_uint64)), ("seg_proteome_off", POINTER(c_uint64)),
                ("n_tx", c_uint64), ("tx_id", POINTER(c_uint32)), ("tx_res_begin", POINTER(c_uint64)),
                ("tx_res_end", POINTER(c_uint64))]


class PackedImage(ctypes.Structure):
    _fields_ = [("desc", POINTER(c_uint64)), ("n_desc", c_uint64), ("chunks", POINTER(v2p_chunk)), ("n_chunks", c_uint64),
                ("payload", POINTER(c_uint8)), ("n_payload", c_uint64), ("hap_out_begin", POINTER(c_uint64)),
                ("n_haps", c_uint64), ("n_tasks", c_uint64), ("n_copy_bytes", c_uint64), ("max_chunk_tasks", c_uint64)]


class TxStreamBuf(ctypes.Structure):
    """v2p_txstream_buf (include/v2p_cohort.h) == the field order of v2p_txstream (include/vcf2prot_hip.h)"""
    _fields_ = [("n_haps", c_uint64), ("n_tx", c_uint64), ("n_tasks", c_uint64), ("n_alt", c_uint64),
                ("hap_tx_begin", POINTER(c_uint64)), ("tx_proteome_off", POINTER(c_uint64)), ("tx_ref_len", POINTER(c_uint32)),
                ("tx_res_len", POINTER(c_uint32)), ("tx_task_begin", POINTER(c_uint64)), ("tx_alt_begin", POINTER(c_uint64)),
                ("code", POINTER(c_uint8)), ("start_pos", POINTER(c_uint32)), ("length", POINTER(c_uint32)),
                ("start_pos_res", POINTER(c_uint32)), ("alt", POINTER(c_uint8)),
                ("tx_header_off", POINTER(c_uint64)), ("tx_header_len", POINTER(c_uint32))]


class Instruction(ctypes.Structure):
    """instruction.rs:6-15 (include/v2p_step4b.h)"""
    _fields_ = [("code", ctypes.c_char), ("s_state", c_uint8), ("pos_ref", c_uint64), ("pos_res", c_uint64),
                ("len", c_uint64), ("data", c_char_p), ("data_len", c_uint64)]


class PatchImage(ctypes.Structure):
    """v2p_patch_image (include/v2p_cohort.h)"""
    _fields_ = [("seg", POINTER(c_uint64)), ("patch", POINTER(ctypes.c_uint32)), ("chunks", c_void_p), ("hap_out_begin", POINTER(c_uint64)),
                ("n_chunks", c_uint64), ("n_haps", c_uint64), ("out_bytes", c_uint64), ("n_seg", c_uint64), ("n_patch", c_uint64)]


COHORT_API = {
    "v2p_inspect_transcript_tasks": (c_int, [c_void_p, c_void_p, c_uint64, c_uint64, POINTER(ctypes.c_int64)]),
    "v2p_transcript_g_rep": (c_int, [POINTER(Instruction), c_uint64, c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_uint64,
                                     POINTER(c_uint64), c_void_p, c_uint64, POINTER(c_uint64), POINTER(c_uint64)]),
    "v2p_cohort_preset": (c_int, [c_char_p, POINTER(CohortParams)]),
    "v2p_cohort_create": (c_int, [POINTER(CohortParams), POINTER(c_void_p)]),
    "v2p_cohort_destroy": (None, [c_void_p]),
    "v2p_cohort_n_haplotypes": (c_uint64, [c_void_p]),
    "v2p_cohort_n_transcripts": (c_uint32, [c_void_p]),
    "v2p_cohort_proteome_len": (c_uint64, [c_void_p]),
    "v2p_cohort_proteome": (POINTER(c_uint8), [c_void_p]),
    "v2p_cohort_tx_offsets": (POINTER(c_uint64), [c_void_p]),
    "v2p_hapbuf_create": (c_void_p, []),
    "v2p_hapbuf_destroy": (None, [c_void_p]),
    "v2p_cohort_generate": (c_int, [c_void_p, c_uint64, c_void_p, POINTER(HapView)]),
    "v2p_cohort_ref_tape_u32": (c_int, [c_void_p, POINTER(HapView), c_void_p]),
    "v2p_cohort_describe": (c_int64, [c_void_p, c_uint64, c_void_p, c_uint64]),
    "v2p_cohort_pack": (c_int, [c_void_p, c_uint64, c_uint64, c_int, c_uint32, c_uint32, c_uint32, POINTER(PackedImage)]),
    "v2p_cohort_fasta_headers": (c_uint64, [c_void_p, c_void_p, c_uint64]),
    "v2p_cohort_result_sizes": (c_int, [c_void_p, c_uint64, c_uint64, c_int, c_void_p]),
    "v2p_packed_free": (None, [POINTER(PackedImage)]),
    "v2p_cohort_launch_bits": (c_int, [c_void_p, c_uint64]),
    "v2p_cohort_txstream": (c_int, [c_void_p, c_uint64, c_uint64, c_int, POINTER(TxStreamBuf)]),
    "v2p_txstream_free": (None, [POINTER(TxStreamBuf)]),
    "v2p_txstream_pack_rows": (c_int, [POINTER(TxStreamBuf), c_uint64, c_int, c_uint32, POINTER(PackedImage), POINTER(c_uint64)]),
    "v2p_txstream_pack_patch": (c_int, [POINTER(TxStreamBuf), c_uint64, POINTER(PatchImage), POINTER(c_uint64)]),
    "v2p_patch_image_free": (None, [POINTER(PatchImage)]),
    "v2p_patch_interpret": (c_int, [c_void_p, c_void_p, c_void_p, c_uint64, c_void_p, c_uint64, c_void_p, c_uint64, c_void_p, c_uint64]),
    "v2p_cohort_pack_grid": (c_int, [c_void_p, c_uint64, c_uint64, c_uint32, c_int, POINTER(PackedImage)]),
}
