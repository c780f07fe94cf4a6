// transcript_tasks.cpp -- step 4b restated: Instruction list of one transcript -> Vec<Task>.
//
// SURVEY.md section 8f rank 3.  Follows /root/reference/src/data_structures/InternalRep/
// transcript_instructions.rs function by function (line numbers in the comments):
//   compute_expected_results_array_size :214-321      get_g_rep            :335-427
//   to_task                             :452-505      add_till_next_ins    :508-629
//   add_last_instruction                :633-651      get_task_from_*      :654-780
//   build_base_instruction              :713-736
// The Instruction struct is the reference's (instruction.rs:6-15); producing Instructions from
// mutations (instruction.rs:64-1098) stays in the Rust host.  Pinned by tests/test_step4b.py
// against the Instruction lists and Task vectors the reference binary printed for the golden
// transcripts (29 of its own unit tests + 7 more shapes).
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/v2p_step4b.h"

namespace {

struct Ins {
    char code; bool s_state; uint64_t pos_ref, pos_res, len; std::string data;
    bool operator==(const Ins& o) const {       // #[derive(PartialEq)] on Instruction
        return code == o.code && s_state == o.s_state && pos_ref == o.pos_ref && pos_res == o.pos_res && len == o.len && data == o.data;
    }
};
struct Tk { uint8_t code; uint64_t start_pos, length, start_pos_res; };

const Tk PHI{2, 0, 0, 0};

bool in(char c, const char* set) { return std::strchr(set, c) != nullptr && c != 0; }

size_t position(const std::vector<Ins>& v, const Ins& x) {   // iter().position(|i| i == x)
    for (size_t i = 0; i < v.size(); ++i) if (v[i] == x) return i;
    return v.size();
}
bool any_gf_before(const std::vector<Ins>& v, const Ins& x) {
    const size_t idx = position(v, x);
    for (size_t i = 0; i < idx; ++i) if (v[i].code == 'G' || v[i].code == 'F') return true;
    return false;
}

// :214-321
int expected_size(const std::vector<Ins>& ins, uint64_t ref_len, uint64_t* out)
{
    int64_t e = 0;
    const int64_t R = int64_t(ref_len);
    for (const Ins& i : ins) {
        const int64_t p = int64_t(i.pos_ref), dl = int64_t(i.data.size()), ln = int64_t(i.len);
        bool stop = false;
        switch (i.code) {
            case 'U': case '0': e -= R; stop = true; break;                                  // :221
            case 'F': e += dl - (R - p); break;                                              // :222
            case 'R': if (!any_gf_before(ins, i)) e += dl - (R - p); break;                  // :223-232
            case 'G': case 'X': e -= R - p; break;                                           // :233
            case 'M': case 'N': case '2': break;                                             // :234
            case 'L': if (i.pos_ref + 1 == ref_len || i.pos_ref == ref_len) e += dl; else e += dl - (R - p); break;   // :235-245
            case 'I': e += dl - 1; break;                                                    // :246
            case 'J': if (!any_gf_before(ins, i)) e += dl - 1; break;                        // :247-256
            case 'D': e -= ln; break;                                                        // :257
            case 'C': if (!any_gf_before(ins, i)) e -= ln; break;                            // :258-267
            case 'K': case 'Q': if (!any_gf_before(ins, i)) e += dl - (R - p); break;        // :268-287
            case 'A': if (!any_gf_before(ins, i)) e -= R - p; break;                         // :288-297
            case 'B': e -= R - p - ln; break;                                                // :298
            case 'P': e -= ln; break;                                                        // :299
            case 'Z': break;                                                                 // :300
            case 'T': e -= R - p; break;                                                     // :301
            case 'W': e += dl; break;                                                        // :302
            case 'Y': e += dl - (R - p) + 1; break;                                          // :303
            case '3': e += dl - ln; break;                                                   // :304
            default: return V2P_4B_UNSUPPORTED;                                              // :305 panic
        }
        if (stop) break;
    }
    if (R + e < 0) return V2P_4B_ARITHMETIC;
    *out = uint64_t(R + e);
    return V2P_4B_OK;
}

// :713-736
Tk build_base_instruction(const Ins& i, uint64_t ref_len)
{
    switch (i.code) {
        case 'Z': case 'Y': return Tk{0, 0, i.pos_ref + 1, 0};
        case 'L':
            if (i.pos_ref + 1 == ref_len) return Tk{0, 0, i.pos_ref + 1, 0};
            if (i.pos_ref == ref_len) return Tk{0, 0, i.pos_ref, 0};
            return Tk{0, 0, i.pos_res, 0};
        default: return Tk{0, 0, i.pos_ref, 0};
    }
}

#define CHK_SUB(a, b) do { if ((a) < (b)) return V2P_4B_ARITHMETIC; } while (0)   /* usize underflow panics in a debug build */

// :508-629
int add_till_next_ins(const Ins& ins, const std::vector<Ins>& all, const Tk& last, uint64_t ref_len, Tk* out)
{
    const size_t pos = position(all, ins);
    if (pos + 1 >= all.size()) return V2P_4B_UNSUPPORTED;
    const Ins& nx = all[pos + 1];
    const uint64_t at = last.start_pos_res + last.length;
    switch (ins.code) {
        case 'D': case 'C': {
            if (nx.pos_ref == ins.pos_ref) { *out = PHI; return V2P_4B_OK; }
            if (ins.pos_ref + ins.len == nx.pos_ref) { *out = PHI; return V2P_4B_OK; }
            const uint64_t start = ins.pos_ref + ins.len + 1;
            if (nx.code == 'L' && nx.pos_ref + 1 == ref_len && start == nx.pos_ref) { *out = Tk{0, start, 1, at}; return V2P_4B_OK; }   // :524-529
            CHK_SUB(nx.pos_ref, start);
            *out = Tk{0, start, nx.pos_ref - start, at};
            return V2P_4B_OK;
        }
        case '2': case '3': {
            if (nx.pos_ref == ins.pos_ref) { *out = PHI; return V2P_4B_OK; }
            if (ins.pos_ref + ins.len == nx.pos_ref) { *out = PHI; return V2P_4B_OK; }
            const uint64_t start = ins.pos_ref + ins.len;
            CHK_SUB(nx.pos_ref, start);
            *out = Tk{0, start, nx.pos_ref - start, at};
            return V2P_4B_OK;
        }
        default: {
            if (nx.pos_ref == ins.pos_ref) { *out = PHI; return V2P_4B_OK; }
            if (nx.code == 'L' && nx.pos_ref + 1 == ref_len) {                                 // :595-602
                CHK_SUB(nx.pos_ref, ins.pos_ref);
                *out = Tk{0, ins.pos_ref + 1, nx.pos_ref - ins.pos_ref, at};
                return V2P_4B_OK;
            }
            CHK_SUB(nx.pos_ref, ins.pos_ref + 1);
            *out = Tk{0, ins.pos_ref + 1, nx.pos_ref - 1 - ins.pos_ref, at};
            return V2P_4B_OK;
        }
    }
}

// :633-651
int add_last_instruction(uint64_t ref_len, const Ins& i, uint64_t at, Tk* out)
{
    switch (i.code) {
        case 'D': case 'C':
            CHK_SUB(ref_len, i.pos_ref + i.len + 1);
            *out = Tk{0, i.pos_ref + i.len + 1, ref_len - i.pos_ref - i.len - 1, at};
            return V2P_4B_OK;
        case '2': case '3':
            CHK_SUB(ref_len, i.pos_ref + i.len);
            *out = Tk{0, i.pos_ref + i.len, ref_len - i.pos_ref - i.len, at};
            return V2P_4B_OK;
        default:
            CHK_SUB(ref_len, i.pos_ref + 1);
            *out = Tk{0, i.pos_ref + 1, ref_len - i.pos_ref - 1, at};
            return V2P_4B_OK;
    }
}

// :452-505
int to_task(const Ins& ins, const std::vector<Ins>& all, std::string& alt, const std::vector<Tk>& tasks, uint64_t ref_len, Tk* t1, Tk* t2)
{
    const Tk& last = tasks.back();
    const uint64_t at = last.start_pos_res + last.length;
    Tk it = PHI;
    switch (ins.code) {
        case 'M': case 'N': {                                       // get_task_from_missense :654-663
            alt += ins.data; alt += ins.data;
            it = Tk{1, alt.size() - ins.data.size(), 1, at};
            break;
        }
        case 'F': case 'R': case 'K': case 'B': case 'Y': {         // get_task_from_frameshift :666-679
            alt += ins.data;
            it = Tk{1, alt.size() - ins.data.size(), ins.len, at};
            break;
        }
        case 'G': case 'X': case 'A': case 'T':                     // stop gained family: phi (:682-693)
        case 'Q': case 'Z': case 'P':                               // :471
            it = PHI;
            break;
        case 'L': case 'W': {                                       // get_task_from_stop_lost :696-710
            alt += ins.data;
            it = Tk{1, alt.size() - ins.data.size(), ins.data.size(), at};
            break;
        }
        case 'I': case 'J': { const uint64_t a = alt.size(); alt += ins.data; it = Tk{1, a, ins.len, at}; break; }            // :739-747
        case 'D': case 'C': { const uint64_t a = alt.size(); alt += ins.data; it = Tk{1, a, ins.data.size(), at}; break; }    // :750-758
        case '2': { const uint64_t a = alt.size(); alt += ins.data; it = Tk{1, a, ins.len, at}; break; }                      // :761-769
        case '3': { const uint64_t a = alt.size(); alt += ins.data; it = Tk{1, a, ins.data.size(), at}; break; }              // :772-780
        default: return V2P_4B_UNSUPPORTED;                          // :479 panic
    }
    *t1 = it;
    const bool is_last = all.back() == ins;                          // :481
    if (is_last) {
        if (in(ins.code, "KYQABPZTWGFRLX")) { *t2 = PHI; return V2P_4B_OK; }               // :486-490
        return add_last_instruction(ref_len, ins, it.start_pos_res + it.length, t2);        // :491
    }
    if (in(ins.code, "KQABPZTWGFRL")) return V2P_4B_MUST_BE_LAST;                          // :496-499
    return add_till_next_ins(ins, all, it, ref_len, t2);                                    // :500
}

}  // namespace

extern "C" int v2p_transcript_g_rep(const v2p_instruction* ins, uint64_t n_ins, uint64_t ref_len,
                                    uint8_t* code, uint64_t* start_pos, uint64_t* length, uint64_t* start_pos_res,
                                    uint64_t cap_tasks, uint64_t* n_tasks,
                                    uint8_t* alt, uint64_t cap_alt, uint64_t* n_alt, uint64_t* res_len)
{
    if ((n_ins && !ins) || !n_tasks || !n_alt || !res_len) return V2P_4B_UNSUPPORTED;
    std::vector<Ins> v(n_ins);
    for (uint64_t i = 0; i < n_ins; ++i)
        v[i] = Ins{ins[i].code, ins[i].s_state != 0, ins[i].pos_ref, ins[i].pos_res, ins[i].len,
                   std::string(ins[i].data ? ins[i].data : "", ins[i].data ? size_t(ins[i].data_len) : 0)};
    *n_tasks = 0; *n_alt = 0; *res_len = 0;
    bool empty = v.empty();                                          // :338-343
    for (const Ins& i : v) if (i.code == '0' || i.code == 'U') empty = true;
    if (empty) return V2P_4B_OK;
    std::vector<Tk> tasks;
    std::string alt_s;
    tasks.push_back(build_base_instruction(v[0], ref_len));          // :353
    for (const Ins& i : v) {                                         // :355-371
        Tk t1, t2;
        const int rc = to_task(i, v, alt_s, tasks, ref_len, &t1, &t2);
        if (rc != V2P_4B_OK) return rc;
        if (t1.code != 2) tasks.push_back(t1);
        if (t2.code != 2) tasks.push_back(t2);
    }
    uint64_t size = 0;
    const int rc = expected_size(v, ref_len, &size);                 // :348,385
    if (rc != V2P_4B_OK) return rc;
    if (tasks.size() > cap_tasks || alt_s.size() > cap_alt) return V2P_4B_CAPACITY;
    for (size_t i = 0; i < tasks.size(); ++i) {
        code[i] = tasks[i].code; start_pos[i] = tasks[i].start_pos; length[i] = tasks[i].length; start_pos_res[i] = tasks[i].start_pos_res;
    }
    if (!alt_s.empty()) memcpy(alt, alt_s.data(), alt_s.size());
    *n_tasks = tasks.size(); *n_alt = alt_s.size(); *res_len = size;
    return V2P_4B_OK;
}

// transcript_instructions.rs:386-421
extern "C" int v2p_inspect_transcript_tasks(const uint64_t* length, const uint64_t* start_pos_res, uint64_t n_tasks, uint64_t res_len,
                                            int64_t* first_bad)
{
    if (first_bad) *first_bad = -1;
    if (n_tasks == 0) return V2P_4B_INSPECT_OK;                 // the empty GIR returns before the validation (:338-343)
    uint64_t counter = 0;
    for (uint64_t i = 1; i < n_tasks; ++i) {
        if (start_pos_res[i] != start_pos_res[i - 1] + length[i - 1]) {
            if (first_bad) *first_bad = int64_t(i);
            return V2P_4B_INSPECT_NOT_CONTIGUOUS;
        }
        counter += length[i];
    }
    counter += length[0];
    return counter == res_len ? V2P_4B_INSPECT_OK : V2P_4B_INSPECT_SIZE_MISMATCH;
}
