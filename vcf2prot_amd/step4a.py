"""Binding of include/v2p_step4a.h: sorted mutations of one AltTranscript -> Instruction list (step 4a,
instruction.rs:20-760 + transcript_instructions.rs:33-160)."""
from __future__ import annotations

import ctypes
from ctypes import POINTER, c_char_p, c_int, c_uint8, c_uint16, c_uint32, c_uint64, c_void_p
from typing import Dict, List, Optional, Sequence, Tuple

from . import _native as N
from ._cohort_api import Instruction

OK, SKIP, PANIC, CAPACITY = 0, 1, 2, 3
INSPECT_INS_GEN, PANIC_INSPECT_ERR = 1, 2
DEFAULT_FLAGS = INSPECT_INS_GEN | PANIC_INSPECT_ERR          # cli.rs:337-368: on unless NO_TEST is exported

SUP_TYPE = ["missense", "*missense", "frameshift", "*frameshift", "inframe_insertion", "*inframe_insertion", "inframe_deletion",
            "*inframe_deletion", "stop_gained", "stop_lost", "*missense&inframe_altering", "*frameshift&stop_retained",
            "*stop_gained&inframe_altering", "frameshift&stop_retained", "inframe_deletion&stop_retained",
            "inframe_insertion&stop_retained", "stop_gained&inframe_altering", "start_lost", "*stop_gained", "stop_lost&frameshift",
            "missense&inframe_altering", "start_lost&splice_region"]


class MutationView(ctypes.Structure):
    _fields_ = [("type", c_uint8), ("ref_aa_position", c_uint16), ("mut_aa_position", c_uint16),
                ("ref_aa", c_char_p), ("ref_aa_len", c_uint32), ("mut_aa", c_char_p), ("mut_aa_len", c_uint32)]


STEP4A_API = {
    "v2p_transcript_instructions": (c_int, [POINTER(MutationView), c_uint64, c_uint32, POINTER(Instruction), c_uint64, POINTER(c_uint64)]),
    "v2p_groups_mutation_view": (c_int, [c_void_p, c_uint32, POINTER(MutationView)]),
}
_bound = []


def _lib():
    lib = N.cohort_lib()
    if not _bound:
        N._bind(lib, STEP4A_API)
        _bound.append(True)
    return lib


def _run(views, n, flags):
    lib = _lib()
    out = (Instruction * max(n, 1))()
    k = c_uint64()
    rc = lib.v2p_transcript_instructions(views, n, flags, out, n, ctypes.byref(k))
    if rc != OK:
        return rc, None
    res = []
    for i in range(k.value):
        o = out[i]
        data = ctypes.string_at(o.data, o.data_len).decode() if o.data_len else ""
        res.append(dict(code=o.code.decode(), s_state=bool(o.s_state), pos_ref=int(o.pos_ref), pos_res=int(o.pos_res), len=int(o.len), data=data))
    return rc, res


def transcript_instructions(mutations: Sequence[Tuple[str, int, int, str, str]], flags: int = DEFAULT_FLAGS):
    """mutations: (type name, ref_aa_position, mut_aa_position, ref_aa, mut_aa), sorted by mut_aa_position.
    Returns (status, [instruction dicts] or None)."""
    n = len(mutations)
    views = (MutationView * max(n, 1))()
    keep = []
    for i, (t, rp, mp, ra, ma) in enumerate(mutations):
        ra_b, ma_b = ra.encode(), ma.encode()
        keep.append((ra_b, ma_b))
        views[i] = MutationView(SUP_TYPE.index(t), rp, mp, ra_b, len(ra_b), ma_b, len(ma_b))
    return _run(views, n, flags)


def group_instructions(groups, member_ids: Sequence[int], flags: int = DEFAULT_FLAGS):
    """Instruction list of one group of a frontend.Groups object (its members are consequence ids)."""
    lib = _lib()
    n = len(member_ids)
    views = (MutationView * max(n, 1))()
    for i, cid in enumerate(member_ids):
        if lib.v2p_groups_mutation_view(groups._h, int(cid), ctypes.byref(views[i])) != 0:
            raise ValueError(f"consequence {cid} is not a valid Mutation")
    return _run(views, n, flags)
