"""CPU oracle for vcf2prot's step-6 SIR executor -- TEST INFRASTRUCTURE ONLY.

Nothing under ``vcf2prot_amd/`` may import this module.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, as
the checker / the reported CPU baseline.

Parity status: PINNED against the reference's own known-answer tests and
against Task vectors + FASTA records harvested from the reference's prebuilt
CPU binary (``oracle/make_golden.py`` -> ``tests/golden/``).

Two restatements live here:

* pure-Python loops (``task_execute`` / ``gir_execute``) that read like the Rust
  (paths relative to /root/reference/src/data_structures/InternalRep):
  ``task.rs:38-50``, ``gir.rs:197-241``; plus the step-5 concatenation
  (``haplotype_instruction.rs:75-158``) and the FASTA record format
  (``personalized_genome.rs:90-113``) that are needed to compare against the
  binary's end-to-end output;
* ``COracle``: ctypes binding of ``oracle/sir_oracle.c`` (same semantics in C,
  fast enough for 10^8 residues, and the timed CPU baseline).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

DOT = ord(".")


class OraclePanic(Exception):
    """Stands for a Rust ``panic!`` on the reference's CPU path."""

    def __init__(self, msg: str, index: int = -1):
        super().__init__(msg)
        self.index = index


@dataclass(frozen=True)
class Task:
    """task.rs:2-9"""

    exe_code: int
    start_pos: int
    length: int
    start_pos_res: int

    def as_tuple(self) -> Tuple[int, int, int, int]:
        return (self.exe_code, self.start_pos, self.length, self.start_pos_res)


def task_execute(task: Task, results_tape: list, ref_tape: Sequence, alt_tape: Sequence) -> None:
    """task.rs:38-50 -- slice copy with Rust's bounds panics."""
    end_bound_res = task.start_pos_res + task.length
    end_bound_stream = task.start_pos + task.length
    src = ref_tape if task.exe_code == 0 else alt_tape
    if end_bound_res > len(results_tape):
        raise OraclePanic(f"range end index {end_bound_res} out of range for slice of length {len(results_tape)}")
    if end_bound_stream > len(src):
        raise OraclePanic(f"range end index {end_bound_stream} out of range for slice of length {len(src)}")
    results_tape[task.start_pos_res:end_bound_res] = src[task.start_pos:end_bound_stream]


def validate_contiguity(tasks: Sequence[Task]) -> int:
    """gir.rs:208-226 predicate.  Returns the first failing index or -1."""
    for idx in range(1, len(tasks)):
        if tasks[idx].start_pos_res != tasks[idx - 1].start_pos_res + tasks[idx - 1].length:
            return idx
    return -1


def gir_execute(tasks: Sequence[Task], ref_stream: Sequence, alt_stream: Sequence, res_array: list,
                engine: str = "st", debug_cpu_exec: bool = False) -> list:
    """gir.rs:197-241.  ``engine`` follows engines.rs:17-29."""
    if engine in ("gpu", "GPU"):
        raise OraclePanic("You are on the CPU version and GPU is not supported !!!")  # gir.rs:238
    if engine not in ("st", "ST", "mt", "MT"):
        raise ValueError(f"{engine} is not a supported engine")  # engines.rs:27
    if debug_cpu_exec:
        bad = validate_contiguity(tasks)
        if bad >= 0:
            raise OraclePanic(
                "Critical failure in the calculations was encountered: position: "
                f"{bad} the sum {tasks[bad].start_pos_res} does not equal previous inputs: "
                f"{tasks[bad - 1].start_pos_res} and {tasks[bad - 1].length} \n", bad)
    for i, t in enumerate(tasks):
        try:
            task_execute(t, res_array, ref_stream, alt_stream)
        except OraclePanic as e:
            raise OraclePanic(str(e), i) from None
    return res_array


# ----------------------------------------------------------------------------
# step 5 (producer of the hot path's inputs) and emit -- needed only to compare
# against the reference binary's end-to-end FASTA output.
# ----------------------------------------------------------------------------
@dataclass
class TranscriptGIR:
    """What TranscriptInstruction::get_g_rep returns (transcript_instructions.rs:335-427)."""

    name: str
    tasks: List[Task]
    alt: str
    ref: str
    res_len: int


def haplotype_concat(girs: Iterable[TranscriptGIR]):
    """haplotype_instruction.rs:75-137 (loop-and-reindex) + update_task :140-158.

    Returns (tasks, annotation{name:(start,end)}, alt_stream, ref_stream, res_array)."""
    g_rep: List[Task] = []
    annotation: Dict[str, Tuple[int, int]] = {}
    alt_array: List[str] = []
    reference_array: List[str] = []
    ref_counter = alt_counter = res_counter = 0
    for g in girs:
        for t in g.tasks:
            if t.exe_code == 0:
                g_rep.append(Task(0, t.start_pos + ref_counter, t.length, t.start_pos_res + res_counter))
            elif t.exe_code == 1:
                g_rep.append(Task(1, t.start_pos + alt_counter, t.length, t.start_pos_res + res_counter))
            else:  # haplotype_instruction.rs:154
                raise OraclePanic(f"Unsupported Stream code: {t.exe_code} from task: {t}")
        # an empty (start-lost) GIR carries empty tapes: transcript_instructions.rs:338-343
        alt_array.extend(g.alt)
        reference_array.extend(g.ref)
        annotation[g.name] = (res_counter, res_counter + g.res_len)
        ref_counter += len(g.ref)
        alt_counter += len(g.alt)
        res_counter += g.res_len
    res_array = ["."] * res_counter  # haplotype_instruction.rs:78
    return g_rep, annotation, alt_array, reference_array, res_array


def fasta_records(res: Sequence[str], annotation: Dict[str, Tuple[int, int]], hap: int) -> List[Tuple[str, str]]:
    """personalized_genome.rs:90-113: '>{name}_{hap}' / seq[start..end]; returned
    as a sorted list because the reference iterates a HashMap (random order)."""
    s = "".join(res)
    out = []
    for name, (a, b) in annotation.items():
        if b > len(s):  # sequence_tape.rs:33-41
            raise OraclePanic(f"Bad Tape Encountered, the provided maximum index is {b} while tape length is {len(s)} ")
        out.append((f"{name}_{hap}", s[a:b]))
    return sorted(out)


# ----------------------------------------------------------------------------
# C restatement binding
# ----------------------------------------------------------------------------
SIR_TASK_DTYPE = np.dtype([("start_pos", "<u8"), ("length", "<u8"), ("start_pos_res", "<u8"),
                           ("exe_code", "u1"), ("_pad", "u1", (7,))])
assert SIR_TASK_DTYPE.itemsize == 32

PANIC_NAMES = {0: "ok", 1: "res_oob", 2: "src_oob", 3: "not_contiguous", 4: "overflow"}


class _SirJob(ctypes.Structure):
    _fields_ = [("tasks", ctypes.c_void_p), ("n_tasks", ctypes.c_uint64),
                ("ref", ctypes.c_void_p), ("n_ref", ctypes.c_uint64),
                ("alt", ctypes.c_void_p), ("n_alt", ctypes.c_uint64),
                ("res", ctypes.c_void_p), ("n_res", ctypes.c_uint64)]


def build_c_oracle(force: bool = False) -> str:
    """Compile oracle/sir_oracle.c -> oracle/libsir_oracle.so (gcc)."""
    src = os.path.join(_HERE, "sir_oracle.c")
    so = os.path.join(_HERE, "libsir_oracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-march=x86-64-v2", "-fPIC", "-shared", "-pthread",
                               "-o", so, src])
    return so


class COracle:
    def __init__(self):
        self.lib = ctypes.CDLL(build_c_oracle())
        L = self.lib
        L.sir_sizeof_task.restype = ctypes.c_size_t
        L.sir_sizeof_job.restype = ctypes.c_size_t
        assert L.sir_sizeof_task() == 32 and L.sir_sizeof_job() == ctypes.sizeof(_SirJob)
        L.sir_validate_contiguity.restype = ctypes.c_int64
        L.sir_validate_contiguity.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        L.sir_gir_execute.restype = ctypes.c_int
        L.sir_gir_execute.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                      ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                      ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
        L.sir_gir_execute_u8.restype = ctypes.c_int
        L.sir_gir_execute_u8.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                         ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64,
                                         ctypes.POINTER(ctypes.c_int64)]
        L.sir_mt_execute.restype = ctypes.c_int
        L.sir_mt_execute.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_double)]
        L.sir_digest_u8.restype = ctypes.c_uint64
        L.sir_digest_u8.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        L.sir_digest_u32.restype = ctypes.c_uint64
        L.sir_digest_u32.argtypes = [ctypes.c_void_p, ctypes.c_uint64]

    @staticmethod
    def pack_tasks(code, start_pos, length, start_pos_res) -> np.ndarray:
        n = len(code)
        t = np.zeros(n, dtype=SIR_TASK_DTYPE)
        t["exe_code"] = np.asarray(code, dtype=np.uint8)
        t["start_pos"] = np.asarray(start_pos, dtype=np.uint64)
        t["length"] = np.asarray(length, dtype=np.uint64)
        t["start_pos_res"] = np.asarray(start_pos_res, dtype=np.uint64)
        return t

    def validate(self, tasks: np.ndarray) -> int:
        return int(self.lib.sir_validate_contiguity(tasks.ctypes.data, len(tasks)))

    def gir_execute(self, tasks: np.ndarray, ref: np.ndarray, alt: np.ndarray, res: np.ndarray,
                    debug_cpu_exec: bool = False) -> np.ndarray:
        """Reference-faithful (uint32 tapes).  ``res`` is modified in place; raises OraclePanic."""
        assert tasks.dtype == SIR_TASK_DTYPE and ref.dtype == np.uint32 and alt.dtype == np.uint32 and res.dtype == np.uint32
        assert res.flags.c_contiguous and ref.flags.c_contiguous and alt.flags.c_contiguous
        bad = ctypes.c_int64(-1)
        rc = self.lib.sir_gir_execute(tasks.ctypes.data, len(tasks), ref.ctypes.data, len(ref),
                                      alt.ctypes.data, len(alt), res.ctypes.data, len(res),
                                      int(debug_cpu_exec), ctypes.byref(bad))
        if rc != 0:
            raise OraclePanic(PANIC_NAMES.get(rc, str(rc)), int(bad.value))
        return res

    def gir_execute_u8(self, tasks: np.ndarray, ref: np.ndarray, alt: np.ndarray, res: np.ndarray) -> np.ndarray:
        assert tasks.dtype == SIR_TASK_DTYPE and ref.dtype == np.uint8 and alt.dtype == np.uint8 and res.dtype == np.uint8
        bad = ctypes.c_int64(-1)
        rc = self.lib.sir_gir_execute_u8(tasks.ctypes.data, len(tasks), ref.ctypes.data, len(ref),
                                         alt.ctypes.data, len(alt), res.ctypes.data, len(res), ctypes.byref(bad))
        if rc != 0:
            raise OraclePanic(PANIC_NAMES.get(rc, str(rc)), int(bad.value))
        return res

    def mt_execute(self, jobs: Sequence[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]],
                   n_threads: int, wide: bool, reps: int = 1) -> float:
        """jobs: (tasks, ref, alt, res) per haplotype.  Returns wall seconds for ``reps`` passes."""
        arr = (_SirJob * len(jobs))()
        for j, (t, ref, alt, res) in enumerate(jobs):
            arr[j] = _SirJob(t.ctypes.data, len(t), ref.ctypes.data, len(ref), alt.ctypes.data, len(alt),
                             res.ctypes.data, len(res))
        secs = ctypes.c_double(0.0)
        rc = self.lib.sir_mt_execute(ctypes.cast(arr, ctypes.c_void_p), len(jobs), n_threads, int(wide), reps,
                                     ctypes.byref(secs))
        if rc != 0:
            raise OraclePanic(PANIC_NAMES.get(rc, str(rc)))
        return float(secs.value)

    def digest_u8(self, a: np.ndarray) -> int:
        a = np.ascontiguousarray(a, dtype=np.uint8)
        return int(self.lib.sir_digest_u8(a.ctypes.data, a.size))

    def digest_u32(self, a: np.ndarray) -> int:
        a = np.ascontiguousarray(a, dtype=np.uint32)
        return int(self.lib.sir_digest_u32(a.ctypes.data, a.size))


def str_to_u32(s: Sequence[str] | str) -> np.ndarray:
    return np.array([ord(c) for c in s], dtype=np.uint32)


def u32_to_str(a: np.ndarray) -> str:
    return "".join(chr(int(c)) for c in a)
