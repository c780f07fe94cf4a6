// cohort_gen.cpp -- synthetic cohorts at the Task boundary (include/v2p_cohort.h).
//
// For every haplotype this restates, for the alteration kinds it draws, what the
// reference's steps 4b and 5 produce (paths under
// /root/reference/src/data_structures/InternalRep):
//   transcript_instructions.rs:335-427  get_g_rep: base task, then per instruction an
//                                       alt task and a "copy reference until next/last" task
//   transcript_instructions.rs:508-651  add_till_next_ins / add_last_instruction
//   transcript_instructions.rs:654-780  alt-tape pushes (missense payload pushed twice)
//   transcript_instructions.rs:713-736  build_base_instruction
//   haplotype_instruction.rs:75-158     concatenation + rebasing by ref/alt/res counters
// It does not parse VCF and it is not the general step-4 compiler: only the shapes
// listed in SURVEY.md Appendix A are generated, and they are pinned against the
// reference binary by tests/golden/c1_example.json.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/v2p_cohort.h"
#include "sir_pack.hpp"
#include "rows_image.hpp"
#include "patch_image_host.hpp"

namespace {

const char AA20[21] = "ACDEFGHIKLMNPQRSTVWY";

struct Rng {   // splitmix64-seeded xoshiro256**
    uint64_t s[4];
    static uint64_t sm(uint64_t& x) {
        uint64_t z = (x += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed, uint64_t stream = 0) {
        uint64_t x = seed ^ (0xD1B54A32D192ED03ull * (stream + 1));
        for (auto& v : s) v = sm(x);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        const uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    double uniform() { return double(next() >> 11) * (1.0 / 9007199254740992.0); }
    uint64_t below(uint64_t n) { return n ? uint64_t((__uint128_t(next()) * n) >> 64) : 0; }
    uint64_t range(uint64_t lo, uint64_t hi) { return lo + below(hi - lo + 1); }   // inclusive
    double normal() {
        double u1 = uniform(), u2 = uniform();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
    uint32_t poisson(double lambda) {
        if (lambda <= 0) return 0;
        const double L = std::exp(-lambda);
        double p = 1.0; uint32_t k = 0;
        do { ++k; p *= uniform(); } while (p > L && k < 1000);
        return k - 1;
    }
    uint8_t residue() { return uint8_t(AA20[below(20)]); }
};

struct Alteration {
    int kind;            // V2P_ALT_* or -1 for start_lost
    uint32_t P;          // 1-based position in the csq amino-acid field
    uint32_t dl;         // residues deleted (deletion only)
    std::string data;    // payload residues (kind specific, see emit_transcript)
};

}  // namespace

struct v2p_cohort {
    v2p_cohort_params p;
    std::vector<uint8_t> proteome;
    std::vector<uint64_t> tx_off;
    double mix_cdf[V2P_ALT_KINDS];
};

struct v2p_hapbuf {
    std::vector<uint8_t> code;
    std::vector<uint64_t> start_pos, length, start_pos_res;
    std::vector<uint8_t> alt;
    std::vector<uint64_t> seg_ref_begin, seg_proteome_off;
    std::vector<uint32_t> tx_id;
    std::vector<uint64_t> tx_res_begin, tx_res_end;
    std::vector<uint32_t> picked;          // scratch
    std::vector<Alteration> alts;          // scratch
};

namespace {

// Alterations of one altered transcript, sorted and made mutually compatible.
void draw_alterations(const v2p_cohort& c, Rng& rng, const uint8_t* ref, uint32_t R, std::vector<Alteration>& out)
{
    const v2p_cohort_params& p = c.p;
    out.clear();
    if (p.p_start_lost > 0 && rng.uniform() < p.p_start_lost) {
        Alteration a; a.kind = -1; a.P = 1; a.dl = 0;
        uint8_t y; do { y = rng.residue(); } while (y == ref[0]);
        a.data.assign(1, char(y));
        out.push_back(a);
        return;
    }
    const uint32_t k = p.alts_fixed ? p.alts_fixed : 1 + rng.poisson(p.alts_poisson);
    std::vector<Alteration> cand(k);
    for (uint32_t i = 0; i < k; ++i) {
        const double u = rng.uniform();
        int kind = 0;
        while (kind < V2P_ALT_KINDS - 1 && u >= c.mix_cdf[kind]) ++kind;
        Alteration& a = cand[i];
        a.kind = kind; a.dl = 0;
        switch (kind) {
            case V2P_ALT_STOP_LOST: a.P = R + 1; break;
            case V2P_ALT_FRAMESHIFT: case V2P_ALT_STOP_GAINED: a.P = uint32_t(rng.range(2, R)); break;
            default: a.P = uint32_t(rng.range(1, R));
        }
    }
    std::stable_sort(cand.begin(), cand.end(), [](const Alteration& x, const Alteration& y) { return x.P < y.P; });
    uint32_t min_P = 1;
    for (Alteration& a : cand) {
        if (a.P < min_P) continue;                      // overlaps the previous alteration: dropped
        const uint32_t pz = a.P - 1;                    // 0-based
        if (a.kind == V2P_ALT_DELETION) {
            uint32_t dl = uint32_t(rng.range(1, p.max_del ? p.max_del : 1));
            if (a.P + dl > R) dl = R - a.P;
            if (dl == 0) a.kind = V2P_ALT_MISSENSE; else a.dl = dl;
        }
        switch (a.kind) {
            case V2P_ALT_MISSENSE: {
                uint8_t y; do { y = rng.residue(); } while (y == ref[pz]);
                a.data.assign(1, char(y));
                min_P = a.P + 1;
                break;
            }
            case V2P_ALT_INSERTION: {                   // data = anchor residue + inserted residues
                const uint32_t n = uint32_t(rng.range(1, p.max_ins ? p.max_ins : 1));
                a.data.assign(1, char(ref[pz]));
                for (uint32_t i = 0; i < n; ++i) a.data.push_back(char(rng.residue()));
                min_P = a.P + 1;
                break;
            }
            case V2P_ALT_DELETION:                      // data = anchor residue, dl residues after it vanish
                a.data.assign(1, char(ref[pz]));
                min_P = a.P + a.dl + 1;
                break;
            case V2P_ALT_FRAMESHIFT: {                  // data = anchor residue + new tail
                const uint32_t n = uint32_t(rng.range(1, p.max_fs_tail ? p.max_fs_tail : 1));
                a.data.assign(1, char(ref[pz]));
                for (uint32_t i = 0; i < n; ++i) a.data.push_back(char(rng.residue()));
                break;
            }
            case V2P_ALT_STOP_GAINED:
                a.data.clear();
                break;
            case V2P_ALT_STOP_LOST: {                   // extension after the last residue
                const uint32_t n = uint32_t(rng.range(1, p.max_sl_ext ? p.max_sl_ext : 1));
                a.data.clear();
                for (uint32_t i = 0; i < n; ++i) a.data.push_back(char(rng.residue()));
                break;
            }
        }
        out.push_back(a);
        if (a.kind == V2P_ALT_FRAMESHIFT || a.kind == V2P_ALT_STOP_GAINED || a.kind == V2P_ALT_STOP_LOST) break;  // must be last
    }
}

// Appends the rebased tasks of one transcript (step 4b shapes + step 5 rebasing).
// Returns the transcript's result length.
template <class TaskSink>
uint64_t emit_transcript(const std::vector<Alteration>& alts, uint32_t R, std::vector<uint8_t>& alt_tape,
                         uint64_t ref_counter, uint64_t res_counter, TaskSink&& task)
{
    if (alts.empty() || alts[0].kind == -1) return 0;           // start_lost: empty GIR (transcript_instructions.rs:338-343)
    uint64_t e = 0;                                             // running end of the transcript's result
    auto ref_task = [&](uint64_t start, uint64_t len) { task(0, ref_counter + start, len, res_counter + e); e += len; };
    auto alt_task = [&](uint64_t start, uint64_t len) { task(1, start, len, res_counter + e); e += len; };
    ref_task(0, alts[0].P - 1);                                 // build_base_instruction :713-736
    for (size_t i = 0; i < alts.size(); ++i) {
        const Alteration& a = alts[i];
        const uint64_t p = a.P - 1;
        const uint64_t at = alt_tape.size();                    // alt_counter + per-transcript offset: the tape is shared
        switch (a.kind) {
            case V2P_ALT_MISSENSE:                              // :654-663, payload pushed twice, task reads the second copy
                alt_tape.push_back(uint8_t(a.data[0])); alt_tape.push_back(uint8_t(a.data[0]));
                alt_task(at + 1, 1);
                break;
            case V2P_ALT_INSERTION:                             // :739-747
            case V2P_ALT_FRAMESHIFT:                            // :666-679
            case V2P_ALT_STOP_LOST:                             // :696-710
                alt_tape.insert(alt_tape.end(), a.data.begin(), a.data.end());
                alt_task(at, a.data.size());
                break;
            case V2P_ALT_DELETION:                              // :750-758
                alt_tape.push_back(uint8_t(a.data[0]));
                alt_task(at, 1);
                break;
            case V2P_ALT_STOP_GAINED:                           // phi task, never pushed (:363-370, :682-686)
                break;
        }
        const bool terminal = a.kind == V2P_ALT_FRAMESHIFT || a.kind == V2P_ALT_STOP_GAINED || a.kind == V2P_ALT_STOP_LOST;
        if (terminal) break;                                    // to_task :486-492: no follow-up copy
        const uint64_t from = a.kind == V2P_ALT_DELETION ? p + a.dl + 1 : p + 1;
        if (i + 1 < alts.size()) {                              // add_till_next_ins :508-629 (zero-length tasks are emitted)
            const Alteration& nx = alts[i + 1];
            const uint64_t until = nx.kind == V2P_ALT_STOP_LOST ? R : nx.P - 1;
            ref_task(from, until - from);
        } else {                                                // add_last_instruction :633-651
            ref_task(from, R - from);
        }
    }
    return e;
}

void generate_into(const v2p_cohort& c, uint64_t hap, v2p_hapbuf& b, bool keep_alts_text, std::string* text)
{
    const v2p_cohort_params& p = c.p;
    const uint32_t T = p.n_transcripts;
    b.code.clear(); b.start_pos.clear(); b.length.clear(); b.start_pos_res.clear(); b.alt.clear();
    b.seg_ref_begin.clear(); b.seg_proteome_off.clear(); b.tx_id.clear(); b.tx_res_begin.clear(); b.tx_res_end.clear();
    b.picked.clear();
    Rng rng(p.seed_cohort, hap);
    const bool empty = p.p_empty_hap > 0 && rng.uniform() < p.p_empty_hap;
    if (!empty) {
        if (p.altered_per_hap == 0 || p.altered_per_hap >= T) {
            b.picked.resize(T);
            for (uint32_t t = 0; t < T; ++t) b.picked[t] = t;
        } else {                                               // selection sampling: sorted, without replacement
            uint32_t need = p.altered_per_hap;
            for (uint32_t t = 0; t < T && need; ++t)
                if (rng.below(T - t) < need) { b.picked.push_back(t); --need; }
        }
    }
    uint64_t ref_counter = 0, res_counter = 0;
    b.seg_ref_begin.push_back(0);
    auto sink = [&](uint8_t code, uint64_t sp, uint64_t len, uint64_t sr) {
        b.code.push_back(code); b.start_pos.push_back(sp); b.length.push_back(len); b.start_pos_res.push_back(sr);
    };
    for (uint32_t t : b.picked) {
        const uint32_t R = uint32_t(c.tx_off[t + 1] - c.tx_off[t]);
        const uint8_t* ref = c.proteome.data() + c.tx_off[t];
        draw_alterations(c, rng, ref, R, b.alts);
        if (keep_alts_text && text) {
            for (const Alteration& a : b.alts) {
                char head[64];
                std::string line;
                snprintf(head, sizeof head, "%u\t", t);
                line = head;
                const std::string P = std::to_string(a.P);
                const uint32_t pz = a.P - 1;
                switch (a.kind) {
                    case -1: line += "start_lost\t" + P + char(ref[0]) + ">" + P + a.data; break;
                    case V2P_ALT_MISSENSE: line += "missense\t" + P + char(ref[pz]) + ">" + P + a.data; break;
                    case V2P_ALT_INSERTION: line += "inframe_insertion\t" + P + char(ref[pz]) + ">" + P + a.data; break;
                    case V2P_ALT_DELETION:
                        line += "inframe_deletion\t" + P + std::string(reinterpret_cast<const char*>(ref + pz), a.dl + 1) + ">" + P + a.data;
                        break;
                    case V2P_ALT_FRAMESHIFT:
                        line += "frameshift\t" + P + std::string(reinterpret_cast<const char*>(ref + pz), R - pz) + "*>" + P + a.data + "*";
                        break;
                    case V2P_ALT_STOP_GAINED: line += "stop_gained\t" + P + char(ref[pz]) + ">" + P + "*"; break;
                    case V2P_ALT_STOP_LOST: line += "stop_lost\t" + P + "*>" + P + a.data; break;
                }
                *text += line + "\n";
            }
        }
        const bool nonempty = !b.alts.empty() && b.alts[0].kind != -1;
        const uint64_t res_len = emit_transcript(b.alts, R, b.alt, ref_counter, res_counter, sink);
        b.tx_id.push_back(t);
        b.tx_res_begin.push_back(res_counter);
        b.tx_res_end.push_back(res_counter + res_len);
        if (nonempty) {                                        // haplotype_instruction.rs:116-132
            b.seg_proteome_off.push_back(c.tx_off[t]);
            ref_counter += R;
            b.seg_ref_begin.push_back(ref_counter);
        }
        res_counter += res_len;
    }
}

void fill_view(const v2p_hapbuf& b, v2p_hap_view* v)
{
    v->n_tasks = b.code.size();
    v->code = b.code.data(); v->start_pos = b.start_pos.data(); v->length = b.length.data(); v->start_pos_res = b.start_pos_res.data();
    v->n_alt = b.alt.size(); v->alt = b.alt.data();
    v->n_res = b.tx_res_end.empty() ? 0 : b.tx_res_end.back();
    v->n_ref = b.seg_ref_begin.back();
    v->n_seg = b.seg_proteome_off.size();
    v->seg_ref_begin = b.seg_ref_begin.data(); v->seg_proteome_off = b.seg_proteome_off.data();
    v->n_tx = b.tx_id.size(); v->tx_id = b.tx_id.data(); v->tx_res_begin = b.tx_res_begin.data(); v->tx_res_end = b.tx_res_end.data();
}

constexpr uint32_t HEADER_BYTES = 19;     // ">ENST%011u_h\n"

// appends one generated haplotype to a device image (resident-proteome form); with `fasta`
// the haplotype's arena range is file-ready: header, residues, line feed per record
int pack_hap(const v2p_cohort& c, uint64_t hap, const v2p_hapbuf& b, v2p::ImageBuilder& img, bool fasta)
{
    v2p::RefSegments segs{b.seg_ref_begin.data(), b.seg_proteome_off.data(), b.seg_proteome_off.size()};
    const uint64_t n_res = b.tx_res_end.empty() ? 0 : b.tx_res_end.back();
    const uint64_t off_alt = img.payload_alloc(b.alt.size());
    if (!b.alt.empty()) memcpy(&img.payload[off_alt], b.alt.data(), b.alt.size());
    auto emit_task = [&](uint64_t i) -> int {
        if (b.code[i] == 0) {
            uint64_t src = 0;
            if (b.length[i] && !segs.map(b.start_pos[i], b.length[i], &src)) return v2p::PACK_SRC_OOB;
            return img.add_task(v2p::SPACE_PROTEOME, src, b.length[i], b.start_pos_res[i], n_res);
        }
        return img.add_task(v2p::SPACE_PAYLOAD, off_alt + b.start_pos[i], b.length[i], b.start_pos_res[i], n_res);
    };
    int rc = v2p::PACK_OK;
    if (fasta) {
        std::vector<uint64_t> hdr_src(b.tx_id.size());
        std::vector<uint32_t> hdr_len(b.tx_id.size(), HEADER_BYTES);
        for (size_t r = 0; r < b.tx_id.size(); ++r)            // header table sits behind the proteome
            hdr_src[r] = c.proteome.size() + 1 + (2ull * b.tx_id[r] + (hap & 1ull)) * HEADER_BYTES;   // table starts with a line feed
        rc = v2p::interleave_fasta(img, b.start_pos_res.data(), b.length.data(), b.code.size(), b.tx_res_end.data(),
                                   hdr_src.data(), hdr_len.data(), b.tx_id.size(), v2p::SPACE_PROTEOME, true, emit_task);
    } else {
        for (size_t i = 0; i < b.code.size() && rc == v2p::PACK_OK; ++i) rc = emit_task(i);
    }
    if (rc != v2p::PACK_OK) return rc;
    img.end_haplotype(n_res);
    return v2p::PACK_OK;
}

}  // namespace

extern "C" {

int v2p_cohort_preset(const char* name, v2p_cohort_params* o)
{
    if (!name || !o) return -1;
    memset(o, 0, sizeof *o);
    o->seed_proteome = 7;
    o->max_ins = 5; o->max_del = 5; o->max_fs_tail = 60; o->max_sl_ext = 30;
    const double full_mix[V2P_ALT_KINDS] = {0.70, 0.08, 0.08, 0.06, 0.05, 0.03};
    if (!strcmp(name, "C1")) {            // plumbing example: every kind, small enough for the reference binary
        o->seed_cohort = 5; o->n_samples = 4; o->n_transcripts = 40; o->fixed_len = 60; o->mean_len = 60;
        o->altered_per_hap = 14; o->alts_poisson = 0.8;
        const double m[V2P_ALT_KINDS] = {0.30, 0.18, 0.18, 0.12, 0.11, 0.11};
        memcpy(o->mix, m, sizeof m);
        o->max_fs_tail = 12; o->max_sl_ext = 8; o->p_start_lost = 0.06; o->p_empty_hap = 0.0;
    } else if (!strcmp(name, "C2")) {     // 1 000 samples x 20 k transcripts x ~400 aa, one missense each
        o->seed_cohort = 9; o->n_samples = 1000; o->n_transcripts = 20000; o->mean_len = 400;
        o->alts_fixed = 1; o->mix[V2P_ALT_MISSENSE] = 1.0;
    } else if (!strcmp(name, "C3")) {     // 10 000 samples, full alteration mix
        o->seed_cohort = 11; o->n_samples = 10000; o->n_transcripts = 20000; o->mean_len = 400;
        o->altered_per_hap = 5000; o->alts_poisson = 1.0; memcpy(o->mix, full_mix, sizeof full_mix);
    } else if (!strcmp(name, "C4")) {     // 1000-Genomes scale
        o->seed_cohort = 12; o->n_samples = 2504; o->n_transcripts = 100000; o->mean_len = 400; o->len_model = 1;
        o->altered_per_hap = 12000; o->alts_poisson = 1.0; memcpy(o->mix, full_mix, sizeof full_mix);
    } else if (!strcmp(name, "C5")) {     // deep Task vectors: 64 alterations per 800-aa transcript
        o->seed_cohort = 13; o->n_samples = 50000; o->n_transcripts = 100; o->fixed_len = 800; o->mean_len = 800;
        o->alts_fixed = 64;
        o->mix[V2P_ALT_MISSENSE] = 0.6; o->mix[V2P_ALT_INSERTION] = 0.2; o->mix[V2P_ALT_DELETION] = 0.2;
    } else {
        return -1;
    }
    return 0;
}

int v2p_cohort_create(const v2p_cohort_params* p, v2p_cohort** out)
{
    if (!p || !out || p->n_transcripts == 0) return -1;
    v2p_cohort* c = new v2p_cohort();
    c->p = *p;
    double s = 0;
    for (int k = 0; k < V2P_ALT_KINDS; ++k) s += p->mix[k];
    if (s <= 0) { delete c; return -1; }
    double acc = 0;
    for (int k = 0; k < V2P_ALT_KINDS; ++k) { acc += p->mix[k] / s; c->mix_cdf[k] = acc; }
    c->mix_cdf[V2P_ALT_KINDS - 1] = 2.0;
    Rng rng(p->seed_proteome, 0x5052u);
    c->tx_off.resize(size_t(p->n_transcripts) + 1);
    c->tx_off[0] = 0;
    for (uint32_t t = 0; t < p->n_transcripts; ++t) {
        uint64_t len;
        if (p->fixed_len) len = p->fixed_len;
        else if (p->len_model == 1) len = uint64_t(std::llround(p->mean_len * std::exp(0.82 * rng.normal())));   // median L, mean ~1.4 L
        else len = uint64_t(std::llround(p->mean_len + 0.25 * p->mean_len * rng.normal()));
        if (len < 50 && !p->fixed_len) len = 50;
        if (len > 35000) len = 35000;
        c->tx_off[t + 1] = c->tx_off[t] + len;
    }
    c->proteome.resize(c->tx_off.back());
    for (uint32_t t = 0; t < p->n_transcripts; ++t) {
        for (uint64_t i = c->tx_off[t]; i < c->tx_off[t + 1]; ++i) c->proteome[i] = rng.residue();
        c->proteome[c->tx_off[t]] = 'M';
    }
    *out = c;
    return 0;
}

void v2p_cohort_destroy(v2p_cohort* c) { delete c; }
uint64_t v2p_cohort_n_haplotypes(const v2p_cohort* c) { return c ? 2ull * c->p.n_samples : 0; }
uint32_t v2p_cohort_n_transcripts(const v2p_cohort* c) { return c ? c->p.n_transcripts : 0; }
uint64_t v2p_cohort_proteome_len(const v2p_cohort* c) { return c ? c->proteome.size() : 0; }
const uint8_t* v2p_cohort_proteome(const v2p_cohort* c) { return c ? c->proteome.data() : nullptr; }
const uint64_t* v2p_cohort_tx_offsets(const v2p_cohort* c) { return c ? c->tx_off.data() : nullptr; }

v2p_hapbuf* v2p_hapbuf_create(void) { return new v2p_hapbuf(); }
void v2p_hapbuf_destroy(v2p_hapbuf* b) { delete b; }

int v2p_cohort_generate(const v2p_cohort* c, uint64_t hap, v2p_hapbuf* buf, v2p_hap_view* view)
{
    if (!c || !buf || !view) return -1;
    generate_into(*c, hap, *buf, false, nullptr);
    fill_view(*buf, view);
    return 0;
}

int v2p_cohort_ref_tape_u32(const v2p_cohort* c, const v2p_hap_view* v, uint32_t* out)
{
    if (!c || !v || (!out && v->n_ref)) return -1;
    for (uint64_t s = 0; s < v->n_seg; ++s) {
        const uint64_t n = v->seg_ref_begin[s + 1] - v->seg_ref_begin[s];
        const uint8_t* src = c->proteome.data() + v->seg_proteome_off[s];
        uint32_t* dst = out + v->seg_ref_begin[s];
        for (uint64_t i = 0; i < n; ++i) dst[i] = src[i];       // chars().collect::<Vec<char>>(), transcript_instructions.rs:349
    }
    return 0;
}

int64_t v2p_cohort_describe(const v2p_cohort* c, uint64_t hap, char* buf, uint64_t cap)
{
    if (!c) return -1;
    v2p_hapbuf b;
    std::string text;
    generate_into(*c, hap, b, true, &text);
    if (buf && cap) {
        const size_t n = text.size() < cap - 1 ? text.size() : size_t(cap - 1);
        memcpy(buf, text.data(), n);
        buf[n] = 0;
    }
    return int64_t(text.size());
}

int v2p_cohort_pack(const v2p_cohort* c, uint64_t h0, uint64_t h1, int n_threads,
                    uint32_t chunk_tasks, uint32_t chunk_bytes, uint32_t flags, v2p_packed_image* out)
{
    const bool fasta = (flags & V2P_PACK_FASTA) != 0;
    if (!c || !out || h1 < h0) return -1;
    memset(out, 0, sizeof *out);
    const uint64_t n = h1 - h0;
    if (n_threads < 1) n_threads = 1;
    if (uint64_t(n_threads) > n && n) n_threads = int(n);
    std::vector<v2p::ImageBuilder> parts(size_t(n_threads ? n_threads : 1));
    for (auto& im : parts) {
        if (flags & V2P_PACK_WAVE) {            // (explicit limits below stay inside what one wave takes)
            im.set_kernel(4);
            if (chunk_tasks > v2p::CHUNK_TASKS_WAVE) chunk_tasks = v2p::CHUNK_TASKS_WAVE;
            if (chunk_bytes > v2p::CHUNK_BYTES_WAVE) chunk_bytes = v2p::CHUNK_BYTES_WAVE;
        }
        if (chunk_tasks) { im.chunk_tasks = chunk_tasks; im.adaptive_tasks = false; }
        if (chunk_bytes) { im.chunk_bytes = chunk_bytes; im.adaptive_bytes = false; }
        if (flags & V2P_PACK_NO_IMM) im.inline_payload = false;
        if (flags & V2P_PACK_NO_FUSE) im.fuse_snv = false;
        if (flags & V2P_PACK_NO_DOUBLE) im.fuse_double = false;
        if (flags & V2P_PACK_NO_LINE_CUT) im.line_cut = false;
        if (flags & V2P_PACK_PER_BLOCK) im.kernel_choice = 2;
        if (flags & V2P_PACK_LONG_RUN) im.kernel_choice = 1;
        if (flags & V2P_PACK_DENSE) im.set_kernel(3);
        if ((flags >> 8) & 0x7FFF) im.cut_align = (flags >> 8) & 0x7FFF;   // experiment knobs: bits 8..22 cut alignment (bit 23: V2P_PACK_NO_LINE_CUT),
        if (flags >> 24) im.soft_window = flags >> 24;                     //                   bits 24..31 closing window
    }
    // one kernel per image (sir_pack.hpp): every thread's builder takes the decision the first haplotype's shape asks for
    if (n && !(flags & (V2P_PACK_PER_BLOCK | V2P_PACK_LONG_RUN | V2P_PACK_DENSE | V2P_PACK_WAVE)) && !chunk_tasks) {
        v2p_hapbuf b;
        generate_into(*c, h0, b, false, nullptr);
        uint64_t bytes = 0, tasks = 0;
        for (size_t i = 0; i < b.length.size(); ++i) { bytes += b.length[i]; tasks += b.length[i] ? 1 : 0; }
        const int choice = !tasks ? 2 : (bytes / tasks >= v2p::WAVE_BYTES_PER_TASK ? 4 : 3);
        for (auto& im : parts) {
            im.set_kernel(choice);
            if (choice == 4 && chunk_bytes) im.chunk_bytes = chunk_bytes < v2p::CHUNK_BYTES_WAVE ? chunk_bytes : v2p::CHUNK_BYTES_WAVE;
        }
    }
    std::vector<int> status(parts.size(), 0);
    // where every thread's part begins in the arena (its haplotypes' result sizes, FASTA text included): chunk cuts are aligned in
    // the arena, so a part must know its absolute offset before it cuts anything
    std::vector<uint64_t> origin(parts.size() + 1, 0);
    {
        auto size_work = [&](int w) {
            const uint64_t a = h0 + n * uint64_t(w) / uint64_t(n_threads), e = h0 + n * uint64_t(w + 1) / uint64_t(n_threads);
            v2p_hapbuf b;
            uint64_t sum = 0;
            for (uint64_t h = a; h < e; ++h) {
                generate_into(*c, h, b, false, nullptr);
                sum += (b.tx_res_end.empty() ? 0 : b.tx_res_end.back()) + (fasta ? uint64_t(HEADER_BYTES + 1) * b.tx_id.size() : 0);
            }
            origin[size_t(w) + 1] = sum;
        };
        std::vector<std::thread> ts;
        for (int w = 1; w < n_threads; ++w) ts.emplace_back(size_work, w);
        size_work(0);
        for (auto& t : ts) t.join();
        for (size_t w = 0; w < parts.size(); ++w) origin[w + 1] += origin[w];
        for (size_t w = 0; w < parts.size(); ++w) parts[w].set_origin(origin[w]);
    }
    auto work = [&](int w) {
        const uint64_t a = h0 + n * uint64_t(w) / uint64_t(n_threads), e = h0 + n * uint64_t(w + 1) / uint64_t(n_threads);
        v2p_hapbuf b;
        for (uint64_t h = a; h < e; ++h) {
            generate_into(*c, h, b, false, nullptr);
            const int rc = pack_hap(*c, h, b, parts[size_t(w)], fasta);
            if (rc) { status[size_t(w)] = rc; return; }
        }
        parts[size_t(w)].finish();
    };
    std::vector<std::thread> th;
    for (int w = 1; w < n_threads; ++w) th.emplace_back(work, w);
    work(0);
    for (auto& t : th) t.join();
    for (int s : status) if (s) return s;
    for (size_t w = 0; w < parts.size(); ++w) if (parts[w].out_size() != origin[w + 1] - origin[w]) return -3;   // (the size pass and the packer disagree)
    uint64_t nd = 0, nc = 0, np = 0;
    for (auto& im : parts) { nd += im.desc.size(); nc += im.chunks.size(); np += im.payload.size(); }
    out->desc = static_cast<uint64_t*>(malloc((nd ? nd : 1) * 8));
    out->chunks = static_cast<v2p_chunk*>(malloc((nc ? nc : 1) * sizeof(v2p_chunk)));
    out->payload = static_cast<uint8_t*>(malloc(np ? np : 1));
    out->hap_out_begin = static_cast<uint64_t*>(malloc((n + 1) * 8));
    if (!out->desc || !out->chunks || !out->payload || !out->hap_out_begin) { v2p_packed_free(out); return -2; }
    // concatenate the per-thread images, rebasing descriptor indices, payload and result offsets
    std::vector<uint64_t> bd(parts.size()), bc(parts.size()), bp(parts.size()), bo(parts.size()), bh(parts.size());
    uint64_t od = 0, oc = 0, op = 0, oo = 0, oh = 0;
    for (size_t w = 0; w < parts.size(); ++w) {
        bd[w] = od; bc[w] = oc; bp[w] = op; bo[w] = oo; bh[w] = oh;
        od += parts[w].desc.size(); oc += parts[w].chunks.size(); op += parts[w].payload.size();
        oo += parts[w].out_size(); oh += parts[w].n_haplotypes();
        out->n_tasks += parts[w].n_ref_tasks; out->n_copy_bytes += parts[w].n_copy_bytes;
        if (parts[w].max_chunk_tasks > out->max_chunk_tasks) out->max_chunk_tasks = parts[w].max_chunk_tasks;
        if (parts[w].max_long_tasks > out->max_chunk_tasks) out->max_chunk_tasks = parts[w].max_long_tasks;
    }
    auto merge = [&](int w) {
        v2p::ImageBuilder& im = parts[size_t(w)];
        uint64_t* d = out->desc + bd[size_t(w)];
        for (size_t i = 0; i < im.desc.size(); ++i) {
            uint64_t x = im.desc[i];
            if (v2p::desc_space(x) == v2p::SPACE_PAYLOAD) x += bp[size_t(w)];      // source offset sits in the low 40 bits
            d[i] = x;
        }
        v2p_chunk* ch = out->chunks + bc[size_t(w)];
        for (size_t i = 0; i < im.chunks.size(); ++i) {
            ch[i].task_begin = im.chunks[i].task_begin + bd[size_t(w)];
            const uint64_t dn = im.chunks[i].dst_n;
            ch[i].dst_n = dn;                                      // (absolute already: ImageBuilder::set_origin)
        }
        if (!im.payload.empty()) memcpy(out->payload + bp[size_t(w)], im.payload.data(), im.payload.size());
        for (size_t i = 0; i + 1 < im.hap_out_begin.size(); ++i) out->hap_out_begin[bh[size_t(w)] + i] = im.hap_out_begin[i];
        im = v2p::ImageBuilder();      // release early
    };
    th.clear();
    for (int w = 1; w < n_threads; ++w) th.emplace_back(merge, w);
    merge(0);
    for (auto& t : th) t.join();
    out->hap_out_begin[n] = oo;
    out->n_desc = nd; out->n_chunks = nc; out->n_payload = np; out->n_haps = n;
    return 0;
}

int v2p_cohort_result_sizes(const v2p_cohort* c, uint64_t h0, uint64_t h1, int n_threads, uint64_t* out)
{
    if (!c || !out || h1 < h0 || h1 > v2p_cohort_n_haplotypes(c)) return -1;
    const uint64_t n = h1 - h0;
    if (n_threads < 1) n_threads = 1;
    if (uint64_t(n_threads) > n && n) n_threads = int(n);
    auto work = [&](int w) {
        v2p_hapbuf b;
        for (uint64_t h = h0 + uint64_t(w); h < h1; h += uint64_t(n_threads)) {
            generate_into(*c, h, b, false, nullptr);
            out[h - h0] = b.tx_res_end.empty() ? 0 : b.tx_res_end.back();
        }
    };
    std::vector<std::thread> th;
    for (int w = 1; w < n_threads; ++w) th.emplace_back(work, w);
    work(0);
    for (auto& t : th) t.join();
    return 0;
}

int v2p_cohort_txstream(const v2p_cohort* c, uint64_t h0, uint64_t h1, int n_threads, v2p_txstream_buf* out)
{
    if (!c || !out || h1 < h0 || h1 > v2p_cohort_n_haplotypes(c)) return -1;
    memset(out, 0, sizeof *out);
    const uint64_t n = h1 - h0;
    if (n_threads < 1) n_threads = 1;
    if (uint64_t(n_threads) > n && n) n_threads = int(n);
    struct Part {
        std::vector<uint64_t> hap_ntx, tx_off; std::vector<uint32_t> tx_ref, tx_res, tx_ntask, tx_nalt;
        std::vector<uint8_t> code, alt; std::vector<uint32_t> sp, ln, sr;
    };
    std::vector<Part> parts(size_t(n_threads ? n_threads : 1));
    auto work = [&](int w) {
        Part& p = parts[size_t(w)];
        const uint64_t a = h0 + n * uint64_t(w) / uint64_t(n_threads), e = h0 + n * uint64_t(w + 1) / uint64_t(n_threads);
        const v2p_cohort_params& cp = c->p;
        const uint32_t T = cp.n_transcripts;
        std::vector<uint32_t> picked;
        std::vector<Alteration> alts;
        std::vector<uint8_t> tape;
        for (uint64_t hap = a; hap < e; ++hap) {                       // the draws of generate_into(), transcript by transcript
            Rng rng(cp.seed_cohort, hap);
            picked.clear();
            const bool empty = cp.p_empty_hap > 0 && rng.uniform() < cp.p_empty_hap;
            if (!empty) {
                if (cp.altered_per_hap == 0 || cp.altered_per_hap >= T) { picked.resize(T); for (uint32_t t = 0; t < T; ++t) picked[t] = t; }
                else { uint32_t need = cp.altered_per_hap; for (uint32_t t = 0; t < T && need; ++t) if (rng.below(T - t) < need) { picked.push_back(t); --need; } }
            }
            p.hap_ntx.push_back(picked.size());
            for (uint32_t t : picked) {
                const uint32_t R = uint32_t(c->tx_off[t + 1] - c->tx_off[t]);
                draw_alterations(*c, rng, c->proteome.data() + c->tx_off[t], R, alts);
                tape.clear();
                const size_t task0 = p.code.size();
                auto sink = [&](uint8_t code, uint64_t sp, uint64_t len, uint64_t sr) {   // ref_counter = res_counter = 0, own alt tape
                    p.code.push_back(code); p.sp.push_back(uint32_t(sp)); p.ln.push_back(uint32_t(len)); p.sr.push_back(uint32_t(sr));
                };
                const uint64_t res_len = emit_transcript(alts, R, tape, 0, 0, sink);
                p.tx_off.push_back(c->tx_off[t]); p.tx_ref.push_back(R); p.tx_res.push_back(uint32_t(res_len));
                p.tx_ntask.push_back(uint32_t(p.code.size() - task0)); p.tx_nalt.push_back(uint32_t(tape.size()));
                p.alt.insert(p.alt.end(), tape.begin(), tape.end());
            }
        }
    };
    std::vector<std::thread> th;
    for (int w = 1; w < n_threads; ++w) th.emplace_back(work, w);
    work(0);
    for (auto& t : th) t.join();
    uint64_t ntx = 0, ntask = 0, nalt = 0;
    for (auto& p : parts) { ntx += p.tx_off.size(); ntask += p.code.size(); nalt += p.alt.size(); }
    out->n_haps = n; out->n_tx = ntx; out->n_tasks = ntask; out->n_alt = nalt;
    auto A = [](uint64_t count, size_t sz) { return malloc((count ? count : 1) * sz); };
    out->hap_tx_begin = (uint64_t*)A(n + 1, 8); out->tx_proteome_off = (uint64_t*)A(ntx, 8); out->tx_ref_len = (uint32_t*)A(ntx, 4);
    out->tx_res_len = (uint32_t*)A(ntx, 4); out->tx_task_begin = (uint64_t*)A(ntx + 1, 8); out->tx_alt_begin = (uint64_t*)A(ntx + 1, 8);
    out->code = (uint8_t*)A(ntask, 1); out->start_pos = (uint32_t*)A(ntask, 4); out->length = (uint32_t*)A(ntask, 4);
    out->start_pos_res = (uint32_t*)A(ntask, 4); out->alt = (uint8_t*)A(nalt, 1);
    if (!out->hap_tx_begin || !out->tx_proteome_off || !out->tx_ref_len || !out->tx_res_len || !out->tx_task_begin || !out->tx_alt_begin ||
        !out->code || !out->start_pos || !out->length || !out->start_pos_res || !out->alt) { v2p_txstream_free(out); return -2; }
    uint64_t hx = 0, tx = 0, tk = 0, al = 0;
    out->hap_tx_begin[0] = 0;
    for (auto& p : parts) {
        for (uint64_t k : p.hap_ntx) { out->hap_tx_begin[hx + 1] = out->hap_tx_begin[hx] + k; ++hx; }
        for (size_t i = 0; i < p.tx_off.size(); ++i) {
            out->tx_proteome_off[tx] = p.tx_off[i]; out->tx_ref_len[tx] = p.tx_ref[i]; out->tx_res_len[tx] = p.tx_res[i];
            out->tx_task_begin[tx] = tk; out->tx_alt_begin[tx] = al;
            tk += p.tx_ntask[i]; al += p.tx_nalt[i]; ++tx;
        }
    }
    out->tx_task_begin[ntx] = tk; out->tx_alt_begin[ntx] = al;
    tk = 0; al = 0;
    for (auto& p : parts) {
        if (!p.code.empty()) {
            memcpy(out->code + tk, p.code.data(), p.code.size());
            memcpy(out->start_pos + tk, p.sp.data(), p.sp.size() * 4); memcpy(out->length + tk, p.ln.data(), p.ln.size() * 4);
            memcpy(out->start_pos_res + tk, p.sr.data(), p.sr.size() * 4);
        }
        if (!p.alt.empty()) memcpy(out->alt + al, p.alt.data(), p.alt.size());
        tk += p.code.size(); al += p.alt.size();
    }
    return 0;
}

void v2p_txstream_free(v2p_txstream_buf* s)
{
    if (!s) return;
    free(s->hap_tx_begin); free(s->tx_proteome_off); free(s->tx_ref_len); free(s->tx_res_len); free(s->tx_task_begin); free(s->tx_alt_begin);
    free(s->code); free(s->start_pos); free(s->length); free(s->start_pos_res); free(s->alt);
    memset(s, 0, sizeof *s);
}

// The host image the device builder must reproduce: the transcript stream through ImageBuilder's step-5 folding with grid cutting.
int v2p_cohort_pack_grid(const v2p_cohort* c, uint64_t h0, uint64_t h1, uint32_t window_bytes, int kernel, v2p_packed_image* out)
{
    if (!c || !out || h1 < h0 || window_bytes == 0 || window_bytes % (kernel == 4 ? 1024u : 4096u)) return -1;
    if (kernel == 4 && window_bytes > v2p::CHUNK_BYTES_WAVE) return -1;        // a wave chunk is at most ten 1 KiB rows
    memset(out, 0, sizeof *out);
    v2p_txstream_buf s;
    int rc = v2p_cohort_txstream(c, h0, h1, 8, &s);
    if (rc) return rc;
    v2p::ImageBuilder im;
    im.grid_bytes = window_bytes; im.kernel_choice = kernel == 1 ? 1 : (kernel == 3 ? 3 : (kernel == 4 ? 4 : 2)); im.adaptive_tasks = false;
    im.payload.assign(s.alt, s.alt + s.n_alt);                        // the alt tapes ARE the payload arena
    for (uint64_t h = 0; h < s.n_haps && rc == 0; ++h) {
        uint64_t res = 0;                                             // res_counter of haplotype_instruction.rs:90,132
        for (uint64_t t = s.hap_tx_begin[h]; t < s.hap_tx_begin[h + 1] && rc == 0; ++t) {
            const uint64_t n_res = res + s.tx_res_len[t];
            for (uint64_t i = s.tx_task_begin[t]; i < s.tx_task_begin[t + 1] && rc == 0; ++i) {
                if (s.code[i] == 0) rc = im.add_task(v2p::SPACE_PROTEOME, s.tx_proteome_off[t] + s.start_pos[i], s.length[i], res + s.start_pos_res[i], n_res);
                else rc = im.add_task(v2p::SPACE_PAYLOAD, s.tx_alt_begin[t] + s.start_pos[i], s.length[i], res + s.start_pos_res[i], n_res);
            }
            im.fill_to(n_res);
            res = n_res;
        }
        im.end_haplotype(res);
    }
    im.finish();
    v2p_txstream_free(&s);
    if (rc) return rc;
    if (im.grid_overflow) return v2p::PACK_TOO_LARGE;                      // some window holds more descriptors than its kernel takes
    out->n_desc = im.desc.size(); out->n_chunks = im.chunks.size(); out->n_payload = im.payload.size(); out->n_haps = im.n_haplotypes();
    out->desc = static_cast<uint64_t*>(malloc((out->n_desc ? out->n_desc : 1) * 8));
    out->chunks = static_cast<v2p_chunk*>(malloc((out->n_chunks ? out->n_chunks : 1) * sizeof(v2p_chunk)));
    out->payload = static_cast<uint8_t*>(malloc(out->n_payload ? out->n_payload : 1));
    out->hap_out_begin = static_cast<uint64_t*>(malloc((out->n_haps + 1) * 8));
    if (!out->desc || !out->chunks || !out->payload || !out->hap_out_begin) { v2p_packed_free(out); return -2; }
    memcpy(out->desc, im.desc.data(), out->n_desc * 8);
    memcpy(out->chunks, im.chunks.data(), out->n_chunks * sizeof(v2p_chunk));
    if (out->n_payload) memcpy(out->payload, im.payload.data(), out->n_payload);
    memcpy(out->hap_out_begin, im.hap_out_begin.data(), (out->n_haps + 1) * 8);
    out->n_tasks = im.n_ref_tasks; out->n_copy_bytes = im.n_copy_bytes;
    out->max_chunk_tasks = im.max_chunk_tasks > im.max_long_tasks ? im.max_chunk_tasks : im.max_long_tasks;
    return 0;
}

// ROWS image of a transcript stream on the host (rows_image.hpp): what v2p_batch_build_on_device(kernel 6 / 7) must reproduce.
int v2p_txstream_pack_rows(const v2p_txstream_buf* s, uint64_t proteome_len, int mode, uint32_t emulate_k, v2p_packed_image* out, uint64_t* status)
{
    if (!s || !out || (mode != v2p::ROWS_WAVE && mode != v2p::ROWS_DENSE) || emulate_k > 64) return -1;
    memset(out, 0, sizeof *out);
    v2p::TxStreamView v{s->n_haps, s->n_tx, s->n_tasks, s->n_alt, s->hap_tx_begin, s->tx_proteome_off, s->tx_ref_len, s->tx_res_len, s->tx_task_begin, s->tx_alt_begin,
                        s->code, s->start_pos, s->length, s->start_pos_res, s->alt, s->tx_header_off, s->tx_header_len};
    v2p::RowsImage im;
    std::vector<uint64_t> cover;
    if (emulate_k) v2p::rows_emulate(v, proteome_len, mode, emulate_k, im, &cover);
    else v2p::rows_reference(v, proteome_len, mode, im);
    if (status) *status = im.status;
    if (im.status != ~0ull) return v2p::PACK_RES_OOB;
    if (!v2p::rows_cut(im, mode, emulate_k ? &cover : nullptr)) { if (status) *status = im.status; return v2p::PACK_TOO_LARGE; }
    out->n_desc = im.desc.size(); out->n_chunks = im.chunks.size(); out->n_payload = s->n_alt; out->n_haps = s->n_haps;
    out->desc = static_cast<uint64_t*>(malloc((out->n_desc ? out->n_desc : 1) * 8));
    out->chunks = static_cast<v2p_chunk*>(malloc((out->n_chunks ? out->n_chunks : 1) * sizeof(v2p_chunk)));
    out->payload = static_cast<uint8_t*>(malloc(out->n_payload ? out->n_payload : 1));
    out->hap_out_begin = static_cast<uint64_t*>(malloc((out->n_haps + 1) * 8));
    if (!out->desc || !out->chunks || !out->payload || !out->hap_out_begin) { v2p_packed_free(out); return -2; }
    if (out->n_desc) memcpy(out->desc, im.desc.data(), out->n_desc * 8);
    if (out->n_chunks) memcpy(out->chunks, im.chunks.data(), out->n_chunks * sizeof(v2p_chunk));
    if (out->n_payload) memcpy(out->payload, s->alt, out->n_payload);
    memcpy(out->hap_out_begin, im.hap_out_begin.data(), (out->n_haps + 1) * 8);
    out->n_tasks = s->n_tasks;
    for (uint64_t i = 0; i < s->n_tasks; ++i) out->n_copy_bytes += s->length[i];
    return 0;
}

// PATCH image of a transcript stream on the host (patch_image_host.hpp): the sequential restatement of patch_build_kernel's rules.
int v2p_txstream_pack_patch(const v2p_txstream_buf* s, uint64_t proteome_len, v2p_patch_image* out, uint64_t* status)
{
    if (!s || !out) return -1;
    memset(out, 0, sizeof *out);
    v2p::TxStreamView v{s->n_haps, s->n_tx, s->n_tasks, s->n_alt, s->hap_tx_begin, s->tx_proteome_off, s->tx_ref_len, s->tx_res_len, s->tx_task_begin, s->tx_alt_begin,
                        s->code, s->start_pos, s->length, s->start_pos_res, s->alt, s->tx_header_off, s->tx_header_len};
    v2p::PatchImage im;
    v2p::patch_reference(v, proteome_len, im);
    if (status) *status = im.status;
    if (im.status != ~0ull) return (im.status & 0xFF) == v2p::STATUS_PATCH_DECLINED ? v2p::PACK_TOO_LARGE : v2p::PACK_RES_OOB;
    out->n_chunks = im.chunks.size(); out->n_haps = s->n_haps; out->out_bytes = im.out_bytes; out->n_seg = im.total_seg; out->n_patch = im.total_patch;
    out->seg = static_cast<uint64_t*>(malloc((im.seg.size() ? im.seg.size() : 1) * 8));
    out->patch = static_cast<uint32_t*>(malloc((im.patch.size() ? im.patch.size() : 1) * 4));
    out->chunks = static_cast<v2p_chunk*>(malloc((out->n_chunks ? out->n_chunks : 1) * sizeof(v2p_chunk)));
    out->hap_out_begin = static_cast<uint64_t*>(malloc((out->n_haps + 1) * 8));
    if (!out->seg || !out->patch || !out->chunks || !out->hap_out_begin) { v2p_patch_image_free(out); return -2; }
    if (!im.seg.empty()) memcpy(out->seg, im.seg.data(), im.seg.size() * 8);
    if (!im.patch.empty()) memcpy(out->patch, im.patch.data(), im.patch.size() * 4);
    if (out->n_chunks) memcpy(out->chunks, im.chunks.data(), out->n_chunks * sizeof(v2p_chunk));
    memcpy(out->hap_out_begin, im.hap_out_begin.data(), (out->n_haps + 1) * 8);
    return 0;
}

void v2p_patch_image_free(v2p_patch_image* im)
{
    if (!im) return;
    free(im->seg); free(im->patch); free(im->chunks); free(im->hap_out_begin);
    memset(im, 0, sizeof *im);
}

// a PATCH image executed on the host (what stitch_patch_kernel does, cell by cell; refuses a malformed chunk): 0 = done
int v2p_patch_interpret(const uint64_t* seg, const uint32_t* patch, const v2p_chunk* chunks, uint64_t n_chunks, const uint8_t* src0, uint64_t src0_len,
                        const uint8_t* src1, uint64_t src1_len, uint8_t* out, uint64_t out_len)
{
    static_assert(sizeof(v2p_chunk) == sizeof(v2p::Chunk), "chunk records");
    return v2p::patch_interpret(seg, patch, reinterpret_cast<const v2p::Chunk*>(chunks), n_chunks, src0, src0_len, src1, src1_len, out, out_len) ? 0 : -1;
}

uint64_t v2p_cohort_fasta_headers(const v2p_cohort* c, uint8_t* out, uint64_t cap)
{
    if (!c) return 0;
    const uint64_t need = 1 + 2ull * c->p.n_transcripts * HEADER_BYTES;
    if (!out || cap < need) return need;
    *out++ = '\n';                                  // every header is preceded by a line feed (see interleave_fasta)
    for (uint32_t t = 0; t < c->p.n_transcripts; ++t)
        for (int h = 0; h < 2; ++h) {
            char buf[32];
            snprintf(buf, sizeof buf, ">ENST%011u_%d\n", t, h + 1);          // personalized_genome.rs:92,97
            memcpy(out + (2ull * t + h) * HEADER_BYTES, buf, HEADER_BYTES);
        }
    return need;
}

int v2p_cohort_launch_bits(const v2p_chunk* chunks, uint64_t n_chunks)
{
    if (n_chunks && !chunks) return -1;
    return v2p::stitch_launch_bits(reinterpret_cast<const v2p::Chunk*>(chunks), n_chunks);
}

void v2p_packed_free(v2p_packed_image* img)
{
    if (!img) return;
    free(img->desc); free(img->chunks); free(img->payload); free(img->hap_out_begin);
    memset(img, 0, sizeof *img);
}

}  // extern "C"
