// instructions.cpp -- include/v2p_step4a.h: Mutation -> Instruction, function by function after
// /root/reference/src/data_structures/InternalRep/instruction.rs, and the per-transcript wrapper of
// transcript_instructions.rs:33-160.
#include <cstdint>
#include <cstring>
#include <string_view>
#include <unordered_set>
#include <vector>

#include "../../../include/v2p_step4a.h"

namespace {

enum Type : uint8_t {      // Constants.rs:3-8, same order
    MisSense, SMisSense, FrameShift, SFrameShift, InframeInsertion, SInframeInsertion, InframeDeletion, SInframeDeletion,
    StopGained, StopLost, SMisSenseAndInframeAltering, SFrameShiftAndStopRetained, SStopGainedAndInframeAltering,
    FrameShiftAndStopRetained, InframeDeletionAndStopRetained, InframeInsertionAndStopRetained, StopGainedAndInframeAltering,
    StartLost, SStopGained, StopLostAndFrameShift, MissenseAndInframeAltering, StartLostAndSpliceRegion
};

enum Kind { Seq, End, NotSeq };                      // MutatedString (mutation_ds.rs:50-76)

Kind kind(const char* s, uint32_t n)
{
    if (n == 1 && s[0] == '*') return NotSeq;
    return memchr(s, '*', n) ? End : Seq;
}

struct Panic {};

struct M {
    const v2p_mutation_view& v;
    Kind ref_kind() const { return kind(v.ref_aa, v.ref_aa_len); }
    Kind mut_kind() const { return kind(v.mut_aa, v.mut_aa_len); }
    // Sequence -> every character, EndSequence -> all but the last (data.remove(data.len()-1))
    uint64_t mut_chars() const { return mut_kind() == Seq ? v.mut_aa_len : v.mut_aa_len - 1; }
    uint64_t ref_chars() const { return ref_kind() == Seq ? v.ref_aa_len : v.ref_aa_len - 1; }
};

v2p_instruction make(char code, bool s, uint64_t pos_ref, uint64_t pos_res, uint64_t len, const char* data, uint64_t data_len)
{
    v2p_instruction i{};
    i.code = code; i.s_state = s; i.pos_ref = pos_ref; i.pos_res = pos_res; i.len = len; i.data = data; i.data_len = data_len;
    return i;
}

v2p_instruction phi() { return make('E', false, 0, 0, 0, nullptr, 0); }                                   // generate_phi_instruction

v2p_instruction stop_gained(const M& m) { return make('G', false, m.v.ref_aa_position, m.v.mut_aa_position, 0, nullptr, 0); }

v2p_instruction stop_lost(const M& m)
{
    if (m.mut_kind() == NotSeq) throw Panic{};
    return make('L', false, m.v.ref_aa_position, m.v.mut_aa_position, m.mut_chars(), m.v.mut_aa, m.mut_chars());
}

v2p_instruction frameshift(const M& m)
{
    if (m.mut_kind() == NotSeq) return phi();
    return make('F', false, m.v.ref_aa_position, m.v.mut_aa_position, m.mut_chars(), m.v.mut_aa, m.mut_chars());
}

v2p_instruction missense(const M& m)
{
    if (m.mut_kind() == NotSeq) throw Panic{};
    return make('M', false, m.v.ref_aa_position, m.v.mut_aa_position, 1, m.v.mut_aa, m.mut_chars());
}

// the '2' / '3' branch of insertion, deletion and missense&inframe_altering; positions are taken crosswise there
template <class OnMut, class OnRef>
v2p_instruction block_substitution(const M& m, OnMut on_mut_notseq, OnRef on_ref_notseq)
{
    const uint64_t pos_res = m.v.ref_aa_position, pos_ref = m.v.mut_aa_position;
    if (m.mut_kind() == NotSeq) return on_mut_notseq();
    const uint64_t nd = m.mut_chars();
    if (m.ref_kind() == NotSeq) return on_ref_notseq();
    const uint64_t nr = m.ref_chars();
    if (nd != nr) return make('3', false, pos_ref, pos_res, nr, m.v.mut_aa, nd);
    return make('2', false, pos_ref, pos_res, nd, m.v.mut_aa, nd);
}

v2p_instruction inframe_insertion(const M& m)
{
    switch (m.ref_kind()) {
        case Seq:
            if (m.v.ref_aa_len != 1)
                return block_substitution(m, [&] { return stop_gained(m); }, [&] { return stop_lost(m); });
            break;
        case End: return frameshift(m);
        case NotSeq: throw Panic{};
    }
    switch (m.mut_kind()) {
        case End: return frameshift(m);
        case NotSeq: return stop_gained(m);
        default: break;
    }
    return make('I', false, m.v.ref_aa_position, m.v.mut_aa_position, m.v.mut_aa_len, m.v.mut_aa, m.v.mut_aa_len);
}

v2p_instruction inframe_deletion(const M& m)
{
    if (m.ref_kind() == NotSeq) return stop_gained(m);
    const uint64_t len = m.ref_chars();
    uint64_t nd = 0;
    switch (m.mut_kind()) {
        case Seq:
            if (m.v.mut_aa_len == 1) nd = 1;
            else return block_substitution(m, []() -> v2p_instruction { throw Panic{}; }, []() -> v2p_instruction { throw Panic{}; });
            break;
        case End:
            nd = m.v.mut_aa_len - 1;
            if (nd != 1) return frameshift(m);
            break;
        case NotSeq: return stop_gained(m);
    }
    return make('D', false, m.v.ref_aa_position, m.v.mut_aa_position, len - nd, m.v.mut_aa, nd);
}

// instruction.rs validate_s_state: Mutation's PartialEq compares mut_aa_position only (mutation_ds.rs:174-180)
bool validate_s_state(const v2p_mutation_view* muts, uint64_t n, uint64_t self)
{
    uint64_t index = 0;
    while (muts[index].mut_aa_position != muts[self].mut_aa_position) ++index;
    for (uint64_t k = 0; k < index; ++k) {
        const uint8_t t = muts[k].type;
        if (t == StopGained || t == FrameShift || t == SStopGained) return false;
        if (t == InframeInsertion || t == InframeDeletion) {
            const Kind mk = kind(muts[k].mut_aa, muts[k].mut_aa_len);
            if (mk == NotSeq || mk == End) return false;
        }
    }
    return true;
}

v2p_instruction s_frameshift(const M& m, bool ok)
{
    if (!ok) return phi();
    if (m.mut_kind() == NotSeq) return stop_gained(m);
    v2p_instruction i = frameshift(m);
    i.code = 'R'; i.s_state = 1;
    return i;
}

v2p_instruction recode(v2p_instruction i, char code)
{
    if (i.code != 'E') i.code = code;
    return i;
}

v2p_instruction from_mutation(const v2p_mutation_view* muts, uint64_t n, uint64_t k)
{
    const M m{muts[k]};
    auto valid = [&] { return validate_s_state(muts, n, k); };
    auto starred = [&](v2p_instruction inner, char code) { inner.code = code; inner.s_state = 1; return inner; };
    switch (muts[k].type) {
        case MisSense: return missense(m);
        case SMisSense: return valid() ? starred(missense(m), 'N') : phi();
        case FrameShift: return frameshift(m);
        case SFrameShift: return s_frameshift(m, valid());
        case InframeInsertion: return inframe_insertion(m);
        case SInframeInsertion: {
            if (!valid()) return phi();
            v2p_instruction i = inframe_insertion(m);
            return i.code == 'I' ? starred(i, 'J') : i;
        }
        case InframeDeletion: return inframe_deletion(m);
        case SInframeDeletion: return valid() ? starred(inframe_deletion(m), 'C') : phi();          // 'C' whatever the inner call returned
        case StartLost: return make('0', false, 0, 0, 0, nullptr, 0);
        case StopLost: return stop_lost(m);
        case StopGained: return stop_gained(m);
        case SStopGained: return valid() ? starred(stop_gained(m), 'X') : phi();
        case SMisSenseAndInframeAltering: return recode(s_frameshift(m, valid()), 'K');
        case SFrameShiftAndStopRetained:
            if (m.mut_kind() == NotSeq) return valid() ? make('Q', true, m.v.ref_aa_position, m.v.mut_aa_position, 0, nullptr, 0) : phi();
            return s_frameshift(m, valid());
        case SStopGainedAndInframeAltering: return recode(valid() ? starred(stop_gained(m), 'X') : phi(), 'A');
        case FrameShiftAndStopRetained: return recode(frameshift(m), 'B');
        case InframeDeletionAndStopRetained: {
            v2p_instruction i = stop_gained(m);
            i.code = 'P';
            if (m.ref_kind() == End) i.len = m.v.ref_aa_len - 1;
            return i;
        }
        case InframeInsertionAndStopRetained: return phi();
        case StopGainedAndInframeAltering: return recode(stop_gained(m), 'T');
        case StopLostAndFrameShift: return m.ref_kind() == NotSeq ? stop_lost(m) : frameshift(m);
        case MissenseAndInframeAltering:
            if (m.mut_kind() == NotSeq) return recode(frameshift(m), 'Y');
            return block_substitution(m, []() -> v2p_instruction { throw Panic{}; }, []() -> v2p_instruction { throw Panic{}; });
        case StartLostAndSpliceRegion: return make('U', false, 0, 0, 0, nullptr, 0);
        default: throw Panic{};
    }
}

}  // namespace

extern "C" int v2p_transcript_instructions(const v2p_mutation_view* muts, uint64_t n, uint32_t flags,
                                           v2p_instruction* out, uint64_t cap, uint64_t* n_out)
{
    if (!n_out || (n && (!muts || !out))) return V2P_4A_CAPACITY;
    *n_out = 0;
    uint64_t k = 0;
    try {
        for (uint64_t i = 0; i < n; ++i) {
            if (muts[i].type > StartLostAndSpliceRegion || !muts[i].ref_aa_len || !muts[i].mut_aa_len) return V2P_4A_PANIC;
            const v2p_instruction ins = from_mutation(muts, n, i);
            if (ins.code == 'E') continue;                                   // transcript_instructions.rs:46-50
            if (k == cap) return V2P_4A_CAPACITY;
            out[k++] = ins;
        }
    } catch (const Panic&) {
        return V2P_4A_PANIC;
    }
    if (!k) return V2P_4A_SKIP;                                              // :52-55
    const int trouble = (flags & V2P_4A_PANIC_INSPECT_ERR) ? V2P_4A_PANIC : V2P_4A_SKIP;
    if (flags & V2P_4A_INSPECT_INS_GEN) {
        std::unordered_set<uint64_t> starts;
        for (uint64_t i = 0; i < k; ++i) starts.insert(out[i].pos_ref);
        if (starts.size() != k) return trouble;                              // :62-82
        bool start_lost = false;
        for (uint64_t i = 0; i < k; ++i) start_lost |= out[i].code == '0';
        if (k > 1 && !start_lost) {
            for (uint64_t i = 0; i + 1 < k; ++i) {
                const v2p_instruction &a = out[i], &b = out[i + 1];
                if (b.pos_res <= a.pos_res + a.data_len - 1) return trouble;                                     // :99 (usize arithmetic wraps in a release build)
                if ((a.code == 'C' || a.code == 'D') && b.pos_ref <= a.pos_res + a.len - 1) return trouble;      // :119-121
            }
        }
    }
    *n_out = k;
    return V2P_4A_OK;
}
