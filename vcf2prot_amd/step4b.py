"""Binding of include/v2p_step4b.h: Instruction list of one transcript -> Vec<Task> (step 4b,
transcript_instructions.rs:335-780)."""
from __future__ import annotations

import ctypes
from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import _native as N
from ._cohort_api import Instruction

STATUS = {0: "ok", 1: "must_be_last", 2: "unsupported", 3: "arithmetic", 4: "capacity"}


def transcript_g_rep(instructions: Sequence[Dict], ref_len: int) -> Tuple[int, np.ndarray, bytes, int]:
    """instructions: dicts with code, s_state, pos_ref, pos_res, len, data (instruction.rs:6-15).
    Returns (status, tasks[n,4] as (exe_code, start_pos, length, start_pos_res), alt tape, result length)."""
    lib = N.cohort_lib()
    n = len(instructions)
    arr = (Instruction * max(n, 1))()
    keep = []
    for i, ins in enumerate(instructions):
        data = ins["data"].encode()
        keep.append(data)
        arr[i] = Instruction(ins["code"].encode(), int(bool(ins["s_state"])), ins["pos_ref"], ins["pos_res"], ins["len"], data, len(data))
    cap_t = 3 * n + 4
    cap_a = 2 * sum(len(d) for d in keep) + 8
    code = np.zeros(cap_t, np.uint8)
    sp, ln, sr = (np.zeros(cap_t, np.uint64) for _ in range(3))
    alt = np.zeros(cap_a, np.uint8)
    nt, na, rl = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    rc = lib.v2p_transcript_g_rep(arr, n, ref_len, code.ctypes.data, sp.ctypes.data, ln.ctypes.data, sr.ctypes.data, cap_t,
                                  ctypes.byref(nt), alt.ctypes.data, cap_a, ctypes.byref(na), ctypes.byref(rl))
    k = int(nt.value)
    tasks = np.stack([code[:k].astype(np.uint64), sp[:k], ln[:k], sr[:k]], axis=1) if k else np.zeros((0, 4), np.uint64)
    return rc, tasks, alt[:int(na.value)].tobytes(), int(rl.value)


def inspect_transcript_tasks(tasks: np.ndarray, res_len: int) -> Tuple[int, int]:
    """INSPECT_TXP (transcript_instructions.rs:386-421) on tasks[n,4] = (exe_code, start_pos, length, start_pos_res):
    (0, -1) ok, (1, i) task i does not start where task i-1 ended, (2, -1) the lengths do not add up to res_len."""
    lib = N.cohort_lib()
    t = np.ascontiguousarray(tasks, dtype=np.uint64).reshape(-1, 4)
    ln, sr = np.ascontiguousarray(t[:, 2]), np.ascontiguousarray(t[:, 3])
    bad = ctypes.c_int64(-1)
    rc = lib.v2p_inspect_transcript_tasks(ln.ctypes.data, sr.ctypes.data, t.shape[0], int(res_len), ctypes.byref(bad))
    return int(rc), int(bad.value)
