"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's VCF front end between the record
lines and the per-haplotype AltTranscript lists (BCSQ bitmask decode and
group_muts_per_transcript).  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of the
decode bench may import this module; the product (vcf2prot_amd/) never does.

It follows the reference string for string, including its quirks, so that it can be pinned against
the reference's own unit-test vectors (tests/test_frontend_oracle.py) and against FASTA written by
the reference binary for handcrafted VCFs (tests/golden/decode_cases.json, made by
oracle/make_decode_golden.py):

  readers.rs:185-231            return_if_supported / is_supported_csq     (record filter)
  vcf_ds.rs:67-87               get_consequences_vector                    (text after "BCSQ=")
  vcf_ds.rs:126-190             get_patient_fields                         (drop 9 columns, transpose)
  text_parser.rs:163-252        get_bit_mask / parse_fields / remove_leading_zeros
  MaskDecoder.rs:33-153         BitMask::from_string / get_indices / parse_single_field / parse_concat_values
  vcf_ds.rs:213-329             decode_back / extract_effects             (index, then SUP_TYPE filter)
  text_parser.rs:27-66          split_csq_string
  text_parser.rs:84-145         parse_amino_acid_field / parse_amino_acid_seq_position
  mutation_ds.rs:15-131         MutationType::from_str, MutatedString::from_str, Mutation::new
  vcf_tools.rs:82-131           group_muts_per_transcript / get_unique_transcript
  vcf_ds.rs:351-420             AltTranscript::new / sort_alterations / drop_replicate

A Rust panic is raised as ReferencePanic.  Parity pinned (see above).
"""
from __future__ import annotations

from dataclasses import dataclass

# Constants.rs:2-8
DEF_CONSEQ = ""
SUP_TYPE = ["missense", "*missense", "frameshift", "*frameshift",
            "inframe_insertion", "*inframe_insertion", "inframe_deletion", "*inframe_deletion",
            "stop_gained", "stop_lost", "*missense&inframe_altering", "*frameshift&stop_retained",
            "*stop_gained&inframe_altering", "frameshift&stop_retained", "inframe_deletion&stop_retained",
            "inframe_insertion&stop_retained", "stop_gained&inframe_altering", "start_lost", "*stop_gained",
            "stop_lost&frameshift", "missense&inframe_altering", "start_lost&splice_region"]


class ReferencePanic(Exception):
    """The reference would abort the process here."""


# ---------------------------------------------------------------- Rust integer parsing
def _rust_parse_int(s: str, lo: int, hi: int, allow_minus: bool):
    """core::num from_str_radix(10): optional sign ('+' always, '-' only for signed types), at least one
    ASCII digit, nothing else, value inside [lo, hi].  Returns None for Err."""
    if not s:
        return None
    body = s
    neg = False
    if s[0] == "+":
        body = s[1:]
    elif s[0] == "-":
        if not allow_minus:
            return None
        neg, body = True, s[1:]
    if not body or any(c < "0" or c > "9" for c in body):
        return None
    v = -int(body) if neg else int(body)
    return v if lo <= v <= hi else None


def parse_i32(s):
    return _rust_parse_int(s, -2 ** 31, 2 ** 31 - 1, True)


def parse_u32(s):
    return _rust_parse_int(s, 0, 2 ** 32 - 1, False)


def parse_u16(s):
    return _rust_parse_int(s, 0, 65535, False)


# ---------------------------------------------------------------- readers.rs:185-231
def is_supported_csq(csq: str) -> bool:
    if csq.count("|") != 6:
        return False
    return csq.split("|")[0] in SUP_TYPE


def return_if_supported(line: str) -> bool:
    cols = line.split("\t")
    if len(cols) < 8:
        raise ReferencePanic("index 7 out of range")                     # readers.rs:187
    bcsq = [x for x in cols[7].split(";") if x.startswith("BCSQ=")]
    if not bcsq:
        return False
    parts = bcsq[0].split("=")
    val = parts[1]                                                          # text between the first and the second '='
    if "," in val:
        return any(is_supported_csq(e) for e in val.split(","))
    return is_supported_csq(val)


# ---------------------------------------------------------------- vcf_ds.rs:67-87
def consequences_of(line: str) -> str:
    info = line.split("\t")[7]
    parts = info.split("BCSQ=")
    if len(parts) < 2:
        raise ReferencePanic("index 1 out of range")
    return parts[1]                                                         # up to the next "BCSQ=" or the end of INFO


# ---------------------------------------------------------------- vcf_ds.rs:126-190
def get_patient_fields(records, num_probands):
    res = [[] for _ in range(num_probands)]
    for rec in records:
        fields = rec.split("\t")
        if len(fields) < 9:
            raise ReferencePanic("drain(0..9) past the end")
        fields = fields[9:]
        for i, f in enumerate(fields):
            if i >= num_probands:
                raise ReferencePanic("more sample columns than probands")
            res[i].append(f)
    return res


# ---------------------------------------------------------------- text_parser.rs:163-252
def parse_fields(fields: str) -> str:
    v = parse_i32(fields)
    if v is None:
        return DEF_CONSEQ
    if v < 0:
        raise ReferencePanic(f"An invalid bit mask was encountered: {fields}")
    return fields + "$"


def remove_leading_zeros(fields: str) -> str:
    parts = fields.split(",")
    while parts and parts[-1] == "0":
        parts.pop()
    if not parts:
        return DEF_CONSEQ
    if "-" in fields:
        raise ReferencePanic(f"An invalid bit mask was encountered: {fields}")
    return ",".join(parts)


def get_bit_mask(field: str) -> str:
    n = field.count(":")
    if n == 0:
        return DEF_CONSEQ
    tail = field.split(":")[n]
    if tail == ".":
        return DEF_CONSEQ
    if tail.count(",") == 0:
        return parse_fields(tail)
    tail = remove_leading_zeros(tail)
    if tail == DEF_CONSEQ:
        return tail
    if tail.count(",") == 0:
        return parse_fields(tail)
    return tail


# ---------------------------------------------------------------- MaskDecoder.rs:33-153
def bitmask_from_string(s: str):
    if s == DEF_CONSEQ or s == "0$":
        return None
    if s.endswith("$"):
        v = parse_u32(s[:-1])
        if v is None:
            raise ReferencePanic(f"unwrap on parse::<u32>({s[:-1]!r})")
        return [v]
    out = []
    for e in s.split(","):
        v = parse_u32(e)
        if v is None:
            raise ReferencePanic(f"unwrap on parse::<u32>({e!r})")
        out.append(v)
    return out


def get_indices(words):
    if words is None:
        return None
    h1, h2 = [], []
    if len(words) == 1:
        m, idx = words[0], 0
        while m:
            if m & 1:
                h1.append(idx)
            if (m >> 1) & 1:
                h2.append(idx)
            m >>= 2
            idx += 1
        return h1, h2
    base = 0
    for m in words:
        idx = 0
        while m:
            if m & 1:
                h1.append(base + idx)
            if (m >> 1) & 1:
                h2.append(base + idx)
            m >>= 2
            idx += 1
        base += 15
    return h1, h2


# ---------------------------------------------------------------- vcf_ds.rs:213-329
def extract_effect_indices(n_csq: int, bitmask: str):
    """extract_effects up to the indexing: the (h1, h2) consequence indices, with the bounds panic."""
    ind = get_indices(bitmask_from_string(bitmask))
    if ind is None:
        return [], []
    for lst in ind:
        for i in lst:
            if i >= n_csq:
                raise ReferencePanic(f"index out of bounds: the len is {n_csq} but the index is {i}")
    return ind


def get_type(csq: str) -> str:
    return csq.split("|")[0]


def decode_back_indices(consequences, proband_fields):
    """decode_back for one proband, in index space: two lists of (record, consequence index)."""
    out = ([], [])
    for r, (csq, field) in enumerate(zip(consequences, proband_fields)):
        parts = csq.split(",")
        h = extract_effect_indices(len(parts), get_bit_mask(field))
        for k in (0, 1):
            for i in h[k]:
                if get_type(parts[i]) in SUP_TYPE:
                    out[k].append((r, i))
    return out


def get_csq_per_patient(records, num_probands):
    """vcf_ds.rs:192-211.  Returns (consequences, [(h1, h2) per proband]) with h* = lists of (record, index)."""
    consequences = [consequences_of(r) for r in records]
    table = get_patient_fields(records, num_probands)
    return consequences, [decode_back_indices(consequences, donor) for donor in table]


def read_vcf_text(text: str):
    """readers.rs:8-33 on an in-memory file: (proband names, supported record lines)."""
    lines = text.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    lines = [ln[:-1] if ln.endswith("\r") else ln for ln in lines]         # str::lines strips "\r\n" too
    header = next((ln for ln in lines if ln.startswith("#CHROM")), None)
    if header is None:
        raise ValueError("Could not find a header line")
    if header.endswith("\t"):
        header = header[:-1]
    cols = header.split("\t")
    if len(cols) < 8:
        raise ValueError("minimum number of columns")
    if len(cols) < 9:
        raise ReferencePanic("drain(0..9) past the end")
    names = cols[9:]
    if not names:
        raise ValueError("The file does not contain any patients")
    recs = [ln for ln in lines if not ln.startswith("#")]
    recs = [ln for ln in recs if return_if_supported(ln)]
    if not recs:
        raise ValueError("Could not extract any records from the provided file!!")
    return names, recs


# ---------------------------------------------------------------- text_parser.rs:27-66
def split_csq_string(s: str):
    """Ok -> [type, transcript, aa field]; Err -> None."""
    n = s.count("|")
    res = s.split("|")
    if n == 6:
        if res[3] in ("protein_coding", "NMD"):
            return [res[0], res[2], res[5]]
        return None
    if res[0] == "start_lost":
        if len(res) < 3:
            raise ReferencePanic("index 2 out of range")
        return [res[0], res[2], "1M>1*"]
    return None


# ---------------------------------------------------------------- text_parser.rs:84-145, mutation_ds.rs
MUTATION_TYPES = SUP_TYPE            # MutationType::from_str accepts exactly the SUP_TYPE spellings (mutation_ds.rs:19-46)


def parse_amino_acid_seq_position(s: str):
    if "-" in s:
        return None
    digits = "".join(c for c in s if "0" <= c <= "9")
    pos = parse_u16(digits)
    if pos is None:
        return None
    seq = "".join(c for c in s if not ("0" <= c <= "9"))
    return pos, (seq if seq else "*")


@dataclass
class Mutation:
    transcript_name: str
    mut_type: str
    ref_aa_position: int
    mut_aa_position: int
    ref_aa: str
    mut_aa: str
    source: str = ""

    def identity(self):
        return (self.mut_type, self.ref_aa_position, self.mut_aa_position, self.ref_aa, self.mut_aa)


def mutation_new(csq: str):
    """Mutation::new(split_csq_string(csq)).ok()"""
    info = split_csq_string(csq)
    if info is None:
        return None
    if info[0] not in MUTATION_TYPES:
        return None
    parts = info[2].split(">")
    if len(parts) != 2:
        return None
    a = parse_amino_acid_seq_position(parts[0])
    if a is None:
        return None
    b = parse_amino_acid_seq_position(parts[1])
    if b is None:
        return None
    # MutationInfo::new subtracts one from u16 positions (mutation_ds.rs:96-97); position 0 wraps in a release build
    return Mutation(info[1], info[0], (a[0] - 1) & 0xFFFF, (b[0] - 1) & 0xFFFF, a[1], b[1], csq)


# ---------------------------------------------------------------- vcf_tools.rs:82-131, vcf_ds.rs:351-420
def get_unique_transcript(muts):
    names = []
    for m in muts:
        r = split_csq_string(m)
        if r is not None:
            names.append(r[1])
    return sorted(set(names), key=lambda s: s.encode())                    # Vec<String>::sort is bytewise


def drop_replicate(name, alts):
    # sort_unstable_by(mut_aa_position): for the slice lengths seen here the reference's pdqsort is an insertion
    # sort, i.e. stable; ties between *different* mutations end in the panic below either way
    alts = sorted(alts, key=lambda m: m.mut_aa_position)
    unique = {m.ref_aa_position for m in alts}
    if len(unique) < len(alts):
        ded = []
        for m in alts:
            if ded and ded[-1].identity() == m.identity():
                continue
            ded.append(m)
        alts = ded
        if len(unique) != len(alts):
            raise ReferencePanic(f"Encountered a logical error with analyzing mutations in transcript: {name}")
    return alts


def group_muts_per_transcript(muts):
    """The reference's quadratic grouping: [(transcript, [Mutation...])] in sorted transcript order."""
    res = []
    for t in get_unique_transcript(muts):
        mine = [m for m in muts if t in m]                                  # substring match on the whole csq string
        alts = [x for x in (mutation_new(m) for m in mine) if x is not None]
        res.append((t, drop_replicate(t, alts)))
    return res


def parse_vcf(text: str):
    """parts/io.rs:13-26 in one call: [(proband, groups of haplotype 1, groups of haplotype 2)]."""
    names, recs = read_vcf_text(text)
    consequences, per = get_csq_per_patient(recs, len(names))
    split = [c.split(",") for c in consequences]
    out = []
    for name, (h1, h2) in zip(names, per):
        m1 = [split[r][i] for r, i in h1]
        m2 = [split[r][i] for r, i in h2]
        out.append((name, group_muts_per_transcript(m1), group_muts_per_transcript(m2)))
    return out


# ---------------------------------------------------------------- C restatement (oracle/frontend_oracle.c): CPU baseline
def build_c_frontend(force: bool = False) -> str:
    import os
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    src, so = os.path.join(here, "frontend_oracle.c"), os.path.join(here, "libfrontend_oracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O3", "-march=x86-64-v2", "-fPIC", "-shared", "-pthread", "-o", so, src])
    return so


class CFrontend:
    """ctypes wrapper of fe_decode: same inputs as the device decode (record ranges, consequence table)."""
    CODES = {1: "negative", 2: "parse", 3: "index", 4: "columns", 5: "nomem"}

    def __init__(self):
        import ctypes
        self.ct = ctypes

        class Input(ctypes.Structure):
            _fields_ = [("text", ctypes.c_void_p), ("row_begin", ctypes.c_void_p), ("row_end", ctypes.c_void_p),
                        ("n_rows", ctypes.c_uint64), ("n_samples", ctypes.c_uint64), ("csq_begin", ctypes.c_void_p),
                        ("csq_supported", ctypes.c_void_p)]
        self.Input = Input
        self.lib = ctypes.CDLL(build_c_frontend())
        self.lib.fe_decode.restype = ctypes.c_int
        self.lib.fe_decode.argtypes = [ctypes.POINTER(Input), ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.POINTER(ctypes.c_int64)]
        self.lib.fe_free.argtypes = [ctypes.c_void_p]

    def decode(self, text, row_begin, row_end, n_samples, csq_begin, csq_supported, n_threads=1):
        """numpy arrays in (uint8 text, uint64 ranges, uint32 csq_begin, uint8 supported) -> (rc, hap_begin, ids, err_field)"""
        import numpy as np
        ct = self.ct
        inp = self.Input(text.ctypes.data, row_begin.ctypes.data, row_end.ctypes.data, row_begin.size, n_samples,
                         csq_begin.ctypes.data, csq_supported.ctypes.data)
        hap_begin = np.zeros(2 * n_samples + 1, dtype=np.uint64)
        out, err = ct.c_void_p(), ct.c_int64(-1)
        rc = self.lib.fe_decode(ct.byref(inp), int(n_threads), hap_begin.ctypes.data, ct.byref(out), ct.byref(err))
        ids = None
        if rc == 0:
            n = int(hap_begin[-1])
            ids = np.ctypeslib.as_array(ct.cast(out, ct.POINTER(ct.c_uint32)), shape=(max(n, 1),))[:n].copy()
            self.lib.fe_free(out)
        return rc, hap_begin, ids, err.value


# ================================================================ step 4a: Mutation -> Instruction (instruction.rs)
# instruction.rs:20-53 (from_mutation dispatch), :86-...(interpret_*), validate_s_state (last fn of the impl),
# transcript_instructions.rs:33-160 (TranscriptInstruction::from_alt_transcript with the INSPECT_INS_GEN checks that
# cli.rs:337-368 switches on by default).
@dataclass
class Instruction:
    code: str
    s_state: bool
    pos_ref: int
    pos_res: int
    len: int
    data: str

    def as_dict(self):
        return dict(code=self.code, s_state=self.s_state, pos_ref=self.pos_ref, pos_res=self.pos_res, len=self.len, data=self.data)


def _kind(aa: str) -> str:
    """MutatedString::from_str (mutation_ds.rs:58-76)."""
    if aa == "*":
        return "NotSeq"
    return "End" if "*" in aa else "Seq"


def _chars(aa: str) -> str:
    """Sequence -> all characters; EndSequence -> all but the last one (data.remove(data.len()-1))."""
    return aa if _kind(aa) == "Seq" else aa[:-1]


def _phi():
    return Instruction("E", False, 0, 0, 0, "")


def validate_s_state(m: Mutation, vec) -> bool:
    index = next(i for i, e in enumerate(vec) if e.mut_aa_position == m.mut_aa_position)    # PartialEq compares mut_aa_position only
    for e in vec[:index]:
        if e.mut_type in ("stop_gained", "frameshift", "*stop_gained"):
            return False
        if e.mut_type in ("inframe_insertion", "inframe_deletion") and _kind(e.mut_aa) in ("NotSeq", "End"):
            return False
    return True


def _stop_gained(m):
    return Instruction("G", False, m.ref_aa_position, m.mut_aa_position, 0, "")


def _stop_lost(m):
    if _kind(m.mut_aa) == "NotSeq":
        raise ReferencePanic("Something went wrong, interpreting (stop_lost)")
    d = _chars(m.mut_aa)
    return Instruction("L", False, m.ref_aa_position, m.mut_aa_position, len(d), d)


def _frameshift(m):
    if _kind(m.mut_aa) == "NotSeq":
        return _phi()
    d = _chars(m.mut_aa)
    return Instruction("F", False, m.ref_aa_position, m.mut_aa_position, len(d), d)


def _block_substitution(m, on_mut_notseq, on_ref_notseq):
    """the '2' / '3' branch shared by insertion, deletion and missense&inframe_altering: positions are taken crosswise"""
    pos_res, pos_ref = m.ref_aa_position, m.mut_aa_position
    if _kind(m.mut_aa) == "NotSeq":
        return on_mut_notseq()
    data = _chars(m.mut_aa)
    if _kind(m.ref_aa) == "NotSeq":
        return on_ref_notseq()
    ref_seq = _chars(m.ref_aa)
    if len(data) != len(ref_seq):
        return Instruction("3", False, pos_ref, pos_res, len(ref_seq), data)
    return Instruction("2", False, pos_ref, pos_res, len(data), data)


def _panic(what):
    def f():
        raise ReferencePanic(what)
    return f


def _missense(m):
    if _kind(m.mut_aa) == "NotSeq":
        raise ReferencePanic("Something went wrong, interpreting (missense)")
    return Instruction("M", False, m.ref_aa_position, m.mut_aa_position, 1, _chars(m.mut_aa))


def _inframe_insertion(m):
    k = _kind(m.ref_aa)
    if k == "Seq":
        if len(m.ref_aa) != 1:
            return _block_substitution(m, lambda: _stop_gained(m), lambda: _stop_lost(m))
    elif k == "End":
        return _frameshift(m)
    else:
        raise ReferencePanic("In interpreting an inframe insertion the reference amino acids was just an asterisk")
    km = _kind(m.mut_aa)
    if km == "End":
        return _frameshift(m)
    if km == "NotSeq":
        return _stop_gained(m)
    return Instruction("I", False, m.ref_aa_position, m.mut_aa_position, len(m.mut_aa), m.mut_aa)


def _inframe_deletion(m):
    if _kind(m.ref_aa) == "NotSeq":
        return _stop_gained(m)
    ln = len(_chars(m.ref_aa))
    km = _kind(m.mut_aa)
    if km == "Seq":
        if len(m.mut_aa) == 1:
            data = m.mut_aa
        else:
            return _block_substitution(m, _panic("interpreting failed"), _panic("interpreting failed"))
    elif km == "End":
        data = m.mut_aa[:-1]
        if len(data) != 1:
            return _frameshift(m)
    else:
        return _stop_gained(m)
    return Instruction("D", False, m.ref_aa_position, m.mut_aa_position, ln - len(data), data)


def _s(m, vec, inner, code):
    if not validate_s_state(m, vec):
        return _phi()
    n = inner(m)
    n.code, n.s_state = code, True
    return n


def _s_frameshift(m, vec):
    if not validate_s_state(m, vec):
        return _phi()
    if _kind(m.mut_aa) == "NotSeq":
        return _stop_gained(m)
    n = _frameshift(m)
    n.code, n.s_state = "R", True
    return n


def _recode(n, code):
    if n.code != "E":
        n.code = code
    return n


def instruction_from_mutation(m: Mutation, vec) -> Instruction:
    t = m.mut_type
    if t == "missense":
        return _missense(m)
    if t == "*missense":
        return _s(m, vec, _missense, "N")
    if t == "frameshift":
        return _frameshift(m)
    if t == "*frameshift":
        return _s_frameshift(m, vec)
    if t == "inframe_insertion":
        return _inframe_insertion(m)
    if t == "*inframe_insertion":
        if not validate_s_state(m, vec):
            return _phi()
        n = _inframe_insertion(m)
        if n.code == "I":
            n.code, n.s_state = "J", True
        return n
    if t == "inframe_deletion":
        return _inframe_deletion(m)
    if t == "*inframe_deletion":
        return _s(m, vec, _inframe_deletion, "C")                      # recoded to 'C' whatever the inner call returned
    if t == "start_lost":
        return Instruction("0", False, 0, 0, 0, "")
    if t == "stop_lost":
        return _stop_lost(m)
    if t == "stop_gained":
        return _stop_gained(m)
    if t == "*stop_gained":
        return _s(m, vec, _stop_gained, "X")
    if t == "*missense&inframe_altering":
        return _recode(_s_frameshift(m, vec), "K")
    if t == "*frameshift&stop_retained":
        if _kind(m.mut_aa) == "NotSeq":
            if validate_s_state(m, vec):
                return Instruction("Q", True, m.ref_aa_position, m.mut_aa_position, 0, "")
            return _phi()
        return _s_frameshift(m, vec)
    if t == "*stop_gained&inframe_altering":
        return _recode(_s(m, vec, _stop_gained, "X"), "A")
    if t == "frameshift&stop_retained":
        return _recode(_frameshift(m), "B")
    if t == "inframe_deletion&stop_retained":
        n = _stop_gained(m)
        n.code = "P"
        if _kind(m.ref_aa) == "End":
            n.len = len(m.ref_aa) - 1
        return n
    if t == "inframe_insertion&stop_retained":
        return _phi()
    if t == "stop_gained&inframe_altering":
        return _recode(_stop_gained(m), "T")
    if t == "stop_lost&frameshift":
        return _stop_lost(m) if _kind(m.ref_aa) == "NotSeq" else _frameshift(m)
    if t == "missense&inframe_altering":
        if _kind(m.mut_aa) == "NotSeq":
            return _recode(_frameshift(m), "Y")
        return _block_substitution(m, _panic("unreachable"), _panic("interpreting failed"))
    if t == "start_lost&splice_region":
        return Instruction("U", False, 0, 0, 0, "")
    raise ReferencePanic(f"unknown mutation type {t}")


U64 = (1 << 64) - 1


def transcript_instructions(name, alts, inspect=True, panic_inspect=True):
    """TranscriptInstruction::from_alt_transcript after the reference lookup: list of Instructions, or None where the
    reference returns Err (the transcript is skipped).  alts must already be sorted (drop_replicate's order)."""
    alts = sorted(alts, key=lambda m: m.mut_aa_position)
    ins = [i for i in (instruction_from_mutation(m, alts) for m in alts) if i.code != "E"]
    if not ins:
        return None

    def trouble(msg):
        if panic_inspect:
            raise ReferencePanic(f"Critical error was encountered: for transcript: {name}: {msg}")
        return None
    if inspect:
        if len({i.pos_ref for i in ins}) != len(ins):
            return trouble("some mutations at the same position")
        if len(ins) > 1 and not any(i.code == "0" for i in ins):
            for a, b in zip(ins[:-1], ins[1:]):
                if b.pos_res <= ((a.pos_res + len(a.data) - 1) & U64):
                    return trouble("some mutations overlap")
                if a.code in "CD" and b.pos_ref <= ((a.pos_res + a.len - 1) & U64):
                    return trouble("some mutations overlap")
    return ins
