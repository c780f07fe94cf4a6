// group_muts.cpp -- include/v2p_frontend.h part (3): per-haplotype grouping of consequence ids by transcript.
//
// The reference (vcf_tools.rs:82-96) walks, for every unique transcript of a haplotype, over ALL of the haplotype's
// consequence strings with str::contains -- transcripts x mutations x string length per haplotype.  Here every
// consequence string is parsed once per file (text_parser.rs:27-66,84-145; mutation_ds.rs:78-131), the substring
// matches of other transcript ids are found once per file with a rolling hash, and a haplotype is then one sort of
// (transcript rank, mut_aa_position) keys.  Same groups, same member order, same aborts (vcf_ds.rs:387-420).
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../../include/v2p_frontend.h"
#include "../../../include/v2p_step4a.h"
#include "frontend_common.hpp"

struct ParsedText { std::string ref_aa, mut_aa; };

struct v2p_groups {
    std::string error;
    std::vector<ParsedText> aa;                                       // per consequence: the two amino-acid strings of a valid Mutation
    int64_t error_hap = -1;
    std::vector<uint64_t> tx_begin; std::vector<uint32_t> tx_len;      // unique transcript ids (text ranges), sorted
    std::vector<v2p_mutation> muts;
    std::vector<uint64_t> hap_group_begin;
    std::vector<uint32_t> group_transcript;
    std::vector<uint64_t> group_member_begin;
    std::vector<uint32_t> member_ids;
};

namespace {

using v2p_frontend::sup_type_index;

struct Parsed {
    bool split_ok = false;          // split_csq_string returned Ok
    bool poison = false;            // split_csq_string would index out of range (start_lost with fewer than three fields)
    std::string_view tx;
    bool mut_ok = false;            // Mutation::new returned Ok
    int type = -1;
    uint16_t ref_pos = 0, mut_pos = 0;
    std::string ref_aa, mut_aa;
};

// text_parser.rs:118-145
bool seq_position(std::string_view s, uint16_t& pos, std::string& seq)
{
    if (s.find('-') != std::string_view::npos) return false;
    uint64_t v = 0;
    size_t nd = 0;
    seq.clear();
    for (char c : s) {
        if (c >= '0' && c <= '9') { v = v * 10 + uint64_t(c - '0'); if (v > 1000000) v = 1000000; ++nd; }
        else seq.push_back(c);
    }
    if (!nd || v > 65535) return false;
    pos = uint16_t(v);
    if (seq.empty()) seq = "*";
    return true;
}

Parsed parse_csq(std::string_view s)
{
    Parsed p;
    std::vector<std::string_view> f;
    size_t b = 0;
    while (b <= s.size()) {
        size_t e = s.find('|', b);
        if (e == std::string_view::npos) e = s.size();
        f.push_back(s.substr(b, e - b));
        b = e + 1;
    }
    std::string_view type, aa;
    if (f.size() == 7) {                                         // six separators (text_parser.rs:33-45)
        if (f[3] != "protein_coding" && f[3] != "NMD") return p;
        type = f[0]; p.tx = f[2]; aa = f[5];
    } else if (f[0] == "start_lost") {                           // text_parser.rs:48-57
        if (f.size() < 3) { p.poison = true; return p; }
        type = f[0]; p.tx = f[2]; aa = "1M>1*";
    } else {
        return p;
    }
    p.split_ok = true;
    p.type = sup_type_index(type);                               // mutation_ds.rs:149-156
    if (p.type < 0) return p;
    const size_t gt = aa.find('>');                              // text_parser.rs:87-91: exactly two parts
    if (gt == std::string_view::npos || aa.find('>', gt + 1) != std::string_view::npos) return p;
    uint16_t a = 0, m = 0;
    if (!seq_position(aa.substr(0, gt), a, p.ref_aa)) return p;
    if (!seq_position(aa.substr(gt + 1), m, p.mut_aa)) return p;
    p.ref_pos = uint16_t(a - 1);                                 // mutation_ds.rs:96-97 (u16 arithmetic of a release build)
    p.mut_pos = uint16_t(m - 1);
    p.mut_ok = true;
    return p;
}

struct Table {
    std::vector<Parsed> parsed;
    std::vector<uint32_t> rank;                  // own transcript rank, ~0u if split failed
    std::vector<uint32_t> ident;                 // identity class of drop_replicate's dedup_by (vcf_ds.rs:399-406)
    std::vector<uint32_t> extra_begin, extra;    // other transcript ranks whose id occurs in the consequence text
    std::vector<std::string_view> names;
};

constexpr uint64_t HB = 0x100000001B3ull;

}  // namespace

extern "C" {

int v2p_groups_build(const v2p_vcf_index* x, const uint8_t* text_u8, const uint64_t* hap_begin, const uint32_t* ids,
                     uint64_t n_haps, uint32_t n_threads, v2p_groups** out)
{
    if (!out) return -1;
    *out = nullptr;
    if (!x || !text_u8 || !hap_begin || (!ids && hap_begin[n_haps])) return -1;
    v2p_groups* g = new (std::nothrow) v2p_groups();
    if (!g) return -1;
    *out = g;
    const char* text = reinterpret_cast<const char*>(text_u8);
    const uint64_t n_csq = v2p_vcf_index_n_consequences(x);
    const uint64_t* tb = v2p_vcf_index_csq_text_begin(x);
    const uint32_t* tl = v2p_vcf_index_csq_text_len(x);
    const uint8_t* sup = v2p_vcf_index_csq_supported(x);

    if (!n_threads) n_threads = std::max(1u, std::thread::hardware_concurrency());
    // run f(begin, end) over [0, n_csq) on the worker threads
    auto parallel_csq = [&](auto&& f) {
        const uint32_t nt = uint32_t(std::min<uint64_t>(n_threads, std::max<uint64_t>(1, n_csq / 4096)));
        std::vector<std::thread> th;
        for (uint32_t t = 1; t < nt; ++t) th.emplace_back([&, t] { f(n_csq * t / nt, n_csq * (t + 1) / nt, t); });
        f(0, n_csq / nt, 0u);
        for (auto& x : th) x.join();
        return nt;
    };

    Table T;
    T.parsed.resize(n_csq);
    parallel_csq([&](uint64_t b, uint64_t e, uint32_t) {
        for (uint64_t i = b; i < e; ++i)
            if (sup[i]) T.parsed[i] = parse_csq(std::string_view(text + tb[i], tl[i]));  // unsupported ones never reach a haplotype
    });

    // unique transcript ids of the file, bytewise sorted (Vec<String>::sort of vcf_tools.rs:126-128, file-wide)
    {
        std::vector<std::string_view> all;
        for (auto& p : T.parsed) if (p.split_ok) all.push_back(p.tx);
        std::sort(all.begin(), all.end());
        all.erase(std::unique(all.begin(), all.end()), all.end());
        T.names = std::move(all);
    }
    std::unordered_map<std::string_view, uint32_t> rank_of;
    rank_of.reserve(T.names.size() * 2);
    for (uint32_t r = 0; r < T.names.size(); ++r) {
        rank_of.emplace(T.names[r], r);
        g->tx_begin.push_back(uint64_t(T.names[r].data() - text));
        g->tx_len.push_back(uint32_t(T.names[r].size()));
    }
    T.rank.assign(n_csq, ~0u);
    T.ident.assign(n_csq, ~0u);
    g->muts.resize(n_csq);
    g->aa.resize(n_csq);
    {
        std::unordered_map<std::string, uint32_t> classes;
        std::string key;
        for (uint64_t i = 0; i < n_csq; ++i) {
            const Parsed& p = T.parsed[i];
            v2p_mutation m{};
            m.transcript = ~0u;
            if (p.split_ok) { T.rank[i] = rank_of[p.tx]; m.transcript = T.rank[i]; }
            if (p.mut_ok) {
                m.valid = 1; m.type = uint8_t(p.type); m.ref_aa_position = p.ref_pos; m.mut_aa_position = p.mut_pos;
                key.assign(1, char(p.type));
                key.append(reinterpret_cast<const char*>(&p.ref_pos), 2).append(reinterpret_cast<const char*>(&p.mut_pos), 2);
                key.append(p.ref_aa).push_back('>');
                key.append(p.mut_aa);
                T.ident[i] = classes.emplace(key, uint32_t(classes.size())).first->second;
            }
            g->muts[i] = m;
            if (p.mut_ok) { g->aa[i].ref_aa = p.ref_aa; g->aa[i].mut_aa = p.mut_aa; }
        }
    }
    // str::contains of vcf_tools.rs:91: which OTHER transcript ids occur somewhere in a consequence's text
    {
        std::unordered_map<size_t, std::unordered_map<uint64_t, std::vector<uint32_t>>> by_len;
        for (uint32_t r = 0; r < T.names.size(); ++r) {
            uint64_t h = 0;
            for (char c : T.names[r]) h = h * HB + uint8_t(c);
            by_len[T.names[r].size()][h].push_back(r);
        }
        T.extra_begin.assign(n_csq + 1, 0);
        std::vector<std::vector<uint32_t>> part_extra(n_threads), part_count(n_threads);
        std::vector<uint64_t> part_begin(n_threads, 0);
        const uint32_t used = parallel_csq([&](uint64_t cb, uint64_t ce, uint32_t t) {
            std::vector<uint32_t> found;
            part_begin[t] = cb;
            part_count[t].assign(ce - cb, 0);
            for (uint64_t i = cb; i < ce; ++i) {
                if (!T.parsed[i].split_ok) continue;              // a consequence that does not split never becomes a Mutation
                std::string_view s(text + tb[i], tl[i]);
                found.clear();
                for (auto& [len, table] : by_len) {
                    if (len == 0 || len > s.size()) continue;
                    uint64_t pw = 1, h = 0;
                    for (size_t k = 0; k + 1 < len; ++k) pw *= HB;
                    for (size_t k = 0; k < len; ++k) h = h * HB + uint8_t(s[k]);
                    for (size_t k = 0;; ++k) {
                        auto it = table.find(h);
                        if (it != table.end())
                            for (uint32_t r : it->second)
                                if (r != T.rank[i] && T.names[r] == s.substr(k, len)) found.push_back(r);
                        if (k + len >= s.size()) break;
                        h = (h - uint8_t(s[k]) * pw) * HB + uint8_t(s[k + len]);
                    }
                }
                std::sort(found.begin(), found.end());
                found.erase(std::unique(found.begin(), found.end()), found.end());
                part_count[t][i - cb] = uint32_t(found.size());
                part_extra[t].insert(part_extra[t].end(), found.begin(), found.end());
            }
        });
        for (uint32_t t = 0; t < used; ++t) {                     // parts are consecutive ranges: stitch them in order
            size_t o = 0;
            for (size_t k = 0; k < part_count[t].size(); ++k) {
                T.extra_begin[part_begin[t] + k] = uint32_t(T.extra.size());
                T.extra.insert(T.extra.end(), part_extra[t].begin() + o, part_extra[t].begin() + o + part_count[t][k]);
                o += part_count[t][k];
            }
        }
        T.extra_begin[n_csq] = uint32_t(T.extra.size());
    }

    // ---- per haplotype ----
    struct HapOut { std::vector<uint32_t> group_tx; std::vector<uint32_t> group_size; std::vector<uint32_t> members; };
    std::vector<HapOut> outs(n_haps);
    std::atomic<uint64_t> next{0};
    std::mutex err_mu;
    if (!n_threads) n_threads = std::max(1u, std::thread::hardware_concurrency());
    n_threads = uint32_t(std::min<uint64_t>(n_threads, std::max<uint64_t>(1, n_haps)));
    auto fail = [&](uint64_t h, const std::string& msg) {
        std::lock_guard<std::mutex> lk(err_mu);
        if (g->error_hap < 0 || int64_t(h) < g->error_hap) { g->error_hap = int64_t(h); g->error = msg; }
    };
    auto work = [&]() {
        struct Key { uint32_t rank; uint16_t mut_pos; uint32_t order; uint32_t id; };
        std::vector<Key> keys;
        std::vector<uint32_t> present;
        for (;;) {
            const uint64_t h = next.fetch_add(1);
            if (h >= n_haps) break;
            const uint32_t* L = ids + hap_begin[h];
            const uint32_t n = uint32_t(hap_begin[h + 1] - hap_begin[h]);
            keys.clear();
            present.clear();
            bool bad = false;
            for (uint32_t k = 0; k < n; ++k) {
                const uint32_t id = L[k];
                if (id >= n_csq) { fail(h, "consequence id out of range"); bad = true; break; }
                if (T.parsed[id].poison) { fail(h, "start_lost consequence with fewer than three fields (text_parser.rs:52 would abort)"); bad = true; break; }
                if (T.rank[id] != ~0u) present.push_back(T.rank[id]);
            }
            if (bad) continue;
            std::sort(present.begin(), present.end());
            present.erase(std::unique(present.begin(), present.end()), present.end());
            for (uint32_t k = 0; k < n; ++k) {
                const uint32_t id = L[k];
                if (T.rank[id] == ~0u) continue;
                const uint16_t mp = T.parsed[id].mut_pos;
                keys.push_back(Key{T.rank[id], mp, k, id});
                for (uint32_t e = T.extra_begin[id]; e < T.extra_begin[id + 1]; ++e)
                    if (std::binary_search(present.begin(), present.end(), T.extra[e])) keys.push_back(Key{T.extra[e], mp, k, id});
            }
            // groups in sorted transcript order; inside a group Mutation::new failures drop out (vcf_ds.rs:360-362), then
            // sort_alterations by mut_aa_position (ties keep list order) and drop_replicate
            std::sort(keys.begin(), keys.end(), [&](const Key& a, const Key& b) {
                if (a.rank != b.rank) return a.rank < b.rank;
                const bool va = T.parsed[a.id].mut_ok, vb = T.parsed[b.id].mut_ok;
                if (va != vb) return va;                              // invalid ones to the back of the group
                if (va && a.mut_pos != b.mut_pos) return a.mut_pos < b.mut_pos;
                return a.order < b.order;
            });
            HapOut& o = outs[h];
            size_t i = 0;
            std::vector<uint16_t> refs;
            for (uint32_t r : present) {
                const size_t m0 = o.members.size();
                size_t j = i;
                while (j < keys.size() && keys[j].rank == r) ++j;
                size_t v = i;
                while (v < j && T.parsed[keys[v].id].mut_ok) ++v;     // [i, v) are the group's Mutations
                refs.clear();
                for (size_t k = i; k < v; ++k) refs.push_back(T.parsed[keys[k].id].ref_pos);
                std::sort(refs.begin(), refs.end());
                const size_t n_unique = size_t(std::unique(refs.begin(), refs.end()) - refs.begin());
                if (n_unique < v - i) {
                    for (size_t k = i; k < v; ++k)
                        if (k == i || T.ident[keys[k].id] != T.ident[keys[k - 1].id]) o.members.push_back(keys[k].id);
                    if (o.members.size() - m0 != n_unique) {
                        fail(h, "Encountered a logical error with analyzing mutations in transcript: " + std::string(T.names[r]));
                        bad = true;
                        break;
                    }
                } else {
                    for (size_t k = i; k < v; ++k) o.members.push_back(keys[k].id);
                }
                o.group_tx.push_back(r);
                o.group_size.push_back(uint32_t(o.members.size() - m0));
                i = j;
            }
            if (bad) { o = HapOut(); }
        }
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < n_threads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    if (g->error_hap >= 0) return V2P_ERR_DUPLICATE_POS;

    g->hap_group_begin.assign(n_haps + 1, 0);
    g->group_member_begin.push_back(0);
    for (uint64_t h = 0; h < n_haps; ++h) {
        const HapOut& o = outs[h];
        size_t m = 0;
        for (size_t k = 0; k < o.group_tx.size(); ++k) {
            g->group_transcript.push_back(o.group_tx[k]);
            g->member_ids.insert(g->member_ids.end(), o.members.begin() + m, o.members.begin() + m + o.group_size[k]);
            m += o.group_size[k];
            g->group_member_begin.push_back(g->member_ids.size());
        }
        g->hap_group_begin[h + 1] = g->group_transcript.size();
    }
    return 0;
}

void v2p_groups_destroy(v2p_groups* g) { delete g; }
const char* v2p_groups_error(const v2p_groups* g) { return g ? g->error.c_str() : ""; }
int64_t v2p_groups_error_haplotype(const v2p_groups* g) { return g ? g->error_hap : -1; }
uint64_t v2p_groups_n_transcripts(const v2p_groups* g) { return g ? g->tx_begin.size() : 0; }
int v2p_groups_transcript(const v2p_groups* g, uint64_t rank, uint64_t* begin, uint64_t* len)
{
    if (!g || rank >= g->tx_begin.size() || !begin || !len) return -1;
    *begin = g->tx_begin[rank];
    *len = g->tx_len[rank];
    return 0;
}
const v2p_mutation* v2p_groups_mutations(const v2p_groups* g) { return g ? g->muts.data() : nullptr; }
int v2p_groups_mutation_view(const v2p_groups* g, uint32_t id, v2p_mutation_view* out)
{
    if (!g || !out || id >= g->muts.size() || !g->muts[id].valid) return -1;
    out->type = g->muts[id].type;
    out->ref_aa_position = g->muts[id].ref_aa_position;
    out->mut_aa_position = g->muts[id].mut_aa_position;
    out->ref_aa = g->aa[id].ref_aa.data(); out->ref_aa_len = uint32_t(g->aa[id].ref_aa.size());
    out->mut_aa = g->aa[id].mut_aa.data(); out->mut_aa_len = uint32_t(g->aa[id].mut_aa.size());
    return 0;
}
const uint64_t* v2p_groups_hap_group_begin(const v2p_groups* g) { return g ? g->hap_group_begin.data() : nullptr; }
const uint32_t* v2p_groups_group_transcript(const v2p_groups* g) { return g ? g->group_transcript.data() : nullptr; }
const uint64_t* v2p_groups_group_member_begin(const v2p_groups* g) { return g ? g->group_member_begin.data() : nullptr; }
const uint32_t* v2p_groups_member_ids(const v2p_groups* g) { return g ? g->member_ids.data() : nullptr; }

}  // extern "C"
