// rows_image.hpp -- ROWS images: the device image of a batch built in ONE pass over the transcript stream (round 4).
//
// What v2p_batch_build_on_device produced until round 3 -- descriptors cut on a fixed grid of result windows, counted in one walk
// over the Task arrays and written in a second one, one lane per transcript -- cost the north star's cohort 12.3 ms for an execute
// of 7.6.  A rows image drops the grid:
//   1. PARSE (build_rows.hip: rows_parse_kernel; here: rows_reference / rows_emulate): lane = ITEM of the stream -- every Task
//      (task.rs:2-9); a transcript's first item also opens it (FASTA header, personalized_genome.rs:90-113), its last item also
//      closes it ('.' fill of the cells no task covers, haplotype_instruction.rs:78; FASTA line feed); a transcript without tasks
//      is one item of its own.  Step 5 (haplotype_instruction.rs:94-133) is what it always was here: ref_counter
//      disappears (reference tasks read the resident proteome), alt_counter is tx_alt_begin, res_counter a prefix sum.  The
//      descriptors are the packer's (sir_pack.hpp) -- plain, '.' fill, immediate, fused substitution -- but WHOLE: nothing is cut.
//      The greedy fusion of the packer's stage() -- a three-state machine over the tasks of a transcript -- is solved for 64 tasks
//      at once on ballot masks (rows_parse below): which copies stay held is a recurrence h[i] = f_i(h[i-2]) over one-bit functions,
//      composed by shifting the masks (five steps of a dozen scalar instructions per wave).
//   2. CUT (rows_cut_kernel; here: rows_cut): chunks are cut afterwards on 1 KiB rows of the arena, greedily as the host packer
//      does -- as many rows as fit one wave (ten) while the descriptors fit its lanes (64) -- from a per-row map {descriptor that
//      covers the row's first byte, offset inside it} written during the parse.  A descriptor lying across a cut belongs to both
//      chunks (head skip / row clip in the chunk record, sir_pack.hpp).
// This header is the HOST side: the shared mask solver, a sequential restatement of the parse (ground truth for the tests: the
// packer's state machine without cuts), an emulation of the device kernel's tiles / windows / masks lane by lane (so that its
// logic is checked on the CPU against the restatement before any GPU sees it), and the cutter.
#pragma once
#include "sir_pack.hpp"

namespace v2p {

constexpr uint32_t ROWS_SEG = 640;             // rows per segment of the cutter: chunks never cross a multiple of 640 KiB (one wave walks a segment; 64 ten-row chunks tile it exactly)
constexpr uint32_t ROWS_MAX_WAVE = CHUNK_BYTES_WAVE / ROW_BYTES;      // 10
constexpr uint32_t ROWS_MAX_DENSE = 12;        // the dense kernel's LDS image is 12 KiB

enum : int { ROWS_WAVE = 1, ROWS_DENSE = 2, ROWS_TILES = 3 };      // 3 (round 6): a TILE image -- pieces straight from the parse, a tile of transcripts = the executor's work item (dense_pieces.h)

// h[i] = h[i-2] ? g1[i] : g0[i] for the 64 lanes of a wave, lanes 0 and 1 taking 0 as their input: every lane's bit is a one-bit
// function of the bit two lanes down (even and odd lanes are two independent chains); composing f_i with f_(i-s) for s = 2, 4, ..,
// 32 turns every lane's function into a constant.
V2P_HOST_DEVICE inline uint64_t solve_stride2(uint64_t g0, uint64_t g1)
{
    uint64_t f0 = g0, f1 = (g1 & ~3ull) | (g0 & 3ull);               // lanes 0, 1: input 0 -> constants
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int s = 2; s < 64; s <<= 1) {
        if (f0 == f1) break;                                           // every lane's function is a constant already (the usual case after a step or two)
        const uint64_t sh0 = f0 << s, sh1 = f1 << s;
        const uint64_t n0 = (sh0 & f1) | (~sh0 & f0);
        const uint64_t n1 = (sh1 & f1) | (~sh1 & f0);
        f0 = n0; f1 = n1;
    }
    return f0;
}

// The packer's fusion (ImageBuilder::stage, wave images) over 64 consecutive items.  Masks, one bit per lane:
//   A    a reference copy that may be held as the first part of a fused substitution (len <= 4095, source in range)
//   B    a one-byte literal
//   PS   a reference copy of <= 4095 residues (may close a substitution)
//   C0   ... that goes on one residue behind a literal that opened the run itself (or an empty held copy)
//   C1   ... that goes on one residue behind the copy two lanes down
//   rst  the state machine starts afresh at this lane: a HEAD item, the first task of a transcript, a task behind a gap
// known: lanes 0 and 1 are context of the window before (their h is `carry`); otherwise they are the window's own first lanes.
// Result: h = copies left held (emitted on their own unless absorbed), F = lanes that CLOSE a fused substitution, real = the
// substitution closing here starts with the copy two lanes down (else with the literal one lane down).
struct RowsParse { uint64_t h, F, real; };
V2P_HOST_DEVICE inline RowsParse rows_parse(uint64_t A, uint64_t B, uint64_t PS, uint64_t C0, uint64_t C1, uint64_t rst, bool known, uint64_t carry)
{
    const uint64_t pend = (B << 1) & ~rst;                           // a literal is pending one lane down and nothing reset the machine since
    const uint64_t P = pend & PS;
    const uint64_t rprev = rst << 1;                                 // the machine was reset AT the literal: no held copy before it
    const uint64_t X = P & ((~rprev & C1) | (rprev & C0));           // closes if the copy two down is held
    const uint64_t Y = P & C0;                                       // closes if it is not
    uint64_t g1 = A & ~X, g0 = A & ~Y;
    if (known) { g0 = (g0 & ~3ull) | (carry & 3ull); g1 = (g1 & ~3ull) | (carry & 3ull); }
    RowsParse r;
    r.h = solve_stride2(g0, g1);
    const uint64_t hp2 = r.h << 2;
    r.F = (hp2 & X) | (~hp2 & Y);
    r.real = hp2 & ~rprev;
    return r;
}
// Dense images pair fused substitutions that follow each other directly: second[i] = L[i] & ~second[i-2], L = both short, the second
// one opened by its literal right behind the first one's last copy, its copy going on one residue behind it.
V2P_HOST_DEVICE inline uint64_t rows_pair(uint64_t L, bool known, uint64_t carry /* second of lanes 2, 3 */)
{
    uint64_t g0 = L, g1 = 0;
    if (known) { g0 = (g0 & ~15ull) | (carry & 12ull); g1 = (g1 & ~15ull) | (carry & 12ull); }      // lanes 0..3: context
    return solve_stride2(g0, g1);
}

#if !defined(__HIP_DEVICE_COMPILE__)
// ---- host side -----------------------------------------------------------------------------------------------------------
struct TxStreamView {
    uint64_t n_haps, n_tx, n_tasks, n_alt;
    const uint64_t* hap_tx_begin; const uint64_t* tx_proteome_off; const uint32_t* tx_ref_len; const uint32_t* tx_res_len;
    const uint64_t* tx_task_begin; const uint64_t* tx_alt_begin;
    const uint8_t* code; const uint32_t* start_pos; const uint32_t* length; const uint32_t* start_pos_res; const uint8_t* alt;
    const uint64_t* tx_header_off; const uint32_t* tx_header_len;     // nullptr: plain result tapes
};

struct RowsImage {
    std::vector<uint64_t> desc;
    std::vector<Chunk> chunks;                 // arena order
    std::vector<uint64_t> hap_out_begin;
    uint64_t out_bytes = 0;
    uint64_t status = ~0ull;                   // min over offending tasks of (task << 8 | reason), as the device reports it
};

inline void rows_report(RowsImage& im, uint64_t index, uint32_t reason) { const uint64_t v = (index << 8) | reason; if (v < im.status) im.status = v; }
constexpr uint32_t ROWS_BAD_CODE = 1, ROWS_RES_OOB = 2, ROWS_SRC_OOB = 3, ROWS_NOT_CONTIGUOUS = 4, ROWS_TOO_MANY = 5;

inline void rows_push(std::vector<uint64_t>& d, unsigned space, uint64_t src, uint64_t len)
{
    while (len) {
        const uint32_t piece = uint32_t(len < PIECE_MAX ? len : PIECE_MAX);
        d.push_back(pack_desc(src, piece, space));
        src = ImageBuilder::advance(space, src, piece);
        len -= piece;
    }
}

inline uint64_t rows_arena_len(const TxStreamView& s, uint64_t t)
{
    const uint32_t hl = s.tx_header_len ? s.tx_header_len[t] : 0u;
    return uint64_t(s.tx_res_len[t]) + (hl ? hl + 1u : 0u);
}

// The parse, sequentially: ImageBuilder::stage()'s state machine (wave images) with nothing cut.  mode ROWS_DENSE pairs fused
// substitutions that follow each other directly (rows_pair).
inline void rows_reference(const TxStreamView& s, uint64_t proteome_len, int mode, RowsImage& im)
{
    im = RowsImage();
    std::vector<uint64_t>& d = im.desc;
    struct Fused { bool any = false; uint64_t end_task = 0, run = 0; uint32_t len1 = 0, byte = 0, len2 = 0; } pend;   // dense: a fused substitution that may still pair up
    auto flush_pend = [&]() { if (pend.any) { d.push_back(pack_snv3(pend.run, pend.len1, pend.len2, uint8_t(pend.byte))); pend.any = false; } };
    auto out = [&](unsigned space, uint64_t src, uint64_t len) { if (len) { flush_pend(); rows_push(d, space, src, len); } };
    im.hap_out_begin.assign(1, 0);
    uint64_t res_base = 0, h_next = 0;
    while (h_next < s.n_haps && s.hap_tx_begin[h_next + 1] == 0) { im.hap_out_begin.push_back(0); ++h_next; }
    for (uint64_t t = 0; t < s.n_tx; ++t) {
        const uint32_t hl = s.tx_header_len ? s.tx_header_len[t] : 0u;
        const uint64_t hsrc = hl ? proteome_len + s.tx_header_off[t] : 0ull;
        const uint64_t poff = s.tx_proteome_off[t], alt0 = s.tx_alt_begin[t], n_alt = s.tx_alt_begin[t + 1] - alt0;
        const uint32_t ref_len = s.tx_ref_len[t], res_len = s.tx_res_len[t];
        const uint64_t i0 = s.tx_task_begin[t], i1 = s.tx_task_begin[t + 1];
        if (poff + ref_len > proteome_len) rows_report(im, i0, ROWS_SRC_OOB);
        out(SPACE_PROTEOME, hsrc, hl);
        // state machine: st 0 nothing held; 1 a copy held (s0); 2 copy (or nothing: s0_len = 0) + literal held
        int st = 0;
        uint64_t s0_src = 0, s0_len = 0, s1_byte = 0;
        auto flush = [&]() {
            if (st >= 1) out(SPACE_PROTEOME, s0_src, s0_len);
            if (st == 2) out(SPACE_IMM, s1_byte, 1);
            st = 0;
        };
        uint64_t cur = 0;
        for (uint64_t i = i0; i < i1; ++i) {
            const uint32_t code = s.code[i];
            const uint64_t sp = s.start_pos[i], ln = s.length[i], sr = s.start_pos_res[i];
            uint32_t why = 0;
            if (code > 1u) why = ROWS_BAD_CODE;
            else if (sr + ln > res_len) why = ROWS_RES_OOB;
            else if (sp + ln > (code == 0 ? uint64_t(ref_len) : n_alt)) why = ROWS_SRC_OOB;
            else if (sr < cur) why = ROWS_NOT_CONTIGUOUS;
            if (why) { rows_report(im, i, why); return; }
            if (sr > cur) { flush(); out(SPACE_FILL, 0, sr - cur); }
            unsigned space; uint64_t src;
            if (code == 0) { space = SPACE_PROTEOME; src = poff + sp; }
            else if (ln >= 1 && ln <= IMM_MAX_BYTES) { space = SPACE_IMM; src = 0; for (uint64_t k = 0; k < ln; ++k) src |= uint64_t(s.alt[alt0 + sp + k]) << (8 * k); }
            else { space = SPACE_PAYLOAD; src = alt0 + sp; }
            cur = sr + ln;
            // ImageBuilder::stage (kernel_choice 4)
            if (st == 2) {
                const bool fits = s0_len == 0 ? (ln > 0 && src >= 1 && src - 1 + 1 + ln <= SNV3_MAX_SRC) : (ln == 0 || src == s0_src + s0_len + 1);
                if (space == SPACE_PROTEOME && ln <= ROWS_FUSE_LEN && fits) {
                    const uint64_t run = s0_len == 0 ? src - 1 : s0_src;
                    st = 0;
                    const Fused f{true, i, run, uint32_t(s0_len), uint32_t(s1_byte & 0xFF), uint32_t(ln)};
                    if (mode == ROWS_DENSE && pend.any && pend.end_task + 2 == i && f.len1 == 0 && pend.len1 <= SNV5_MAX_LEN && pend.len2 <= SNV5_MAX_LEN &&
                        f.len2 <= SNV5_MAX_LEN && f.run == pend.run + pend.len1 + 1 + pend.len2) {
                        d.push_back(pack_snv5(pend.run, pend.len1, uint8_t(pend.byte), pend.len2, uint8_t(f.byte), f.len2));
                        pend.any = false;
                    } else {
                        flush_pend();
                        if (mode == ROWS_DENSE) pend = f; else d.push_back(pack_snv3(f.run, f.len1, f.len2, uint8_t(f.byte)));
                    }
                    continue;
                }
                flush();
            }
            if (st == 1) {
                if (space == SPACE_IMM && ln == 1) { s1_byte = src; st = 2; continue; }
                flush();
            }
            if (space == SPACE_PROTEOME && ln <= ROWS_FUSE_LEN && src + ln + 1 + ROWS_FUSE_LEN <= SNV3_MAX_SRC) { s0_src = src; s0_len = ln; st = 1; continue; }
            if (space == SPACE_IMM && ln == 1) { s0_src = 0; s0_len = 0; s1_byte = src; st = 2; continue; }
            out(space, src, ln);
        }
        flush();
        if (cur < res_len) out(SPACE_FILL, 0, res_len - cur);
        if (hl) out(SPACE_PROTEOME, hsrc + hl - 1u, 1);
        res_base += rows_arena_len(s, t);
        while (h_next < s.n_haps && s.hap_tx_begin[h_next + 1] == t + 1) { im.hap_out_begin.push_back(res_base); ++h_next; }
    }
    flush_pend();
    im.out_bytes = res_base;
}

// ---- the device kernel's structure, lane by lane (build_rows.hip: rows_parse_kernel) ----
// Items: every task is one, and a transcript without tasks contributes one item of its own.  A transcript's first item also opens it
// (FASTA header), its last item also closes it ('.' tail, FASTA line feed).  A tile is K consecutive transcripts (one wave); a window
// is 64 consecutive items of it; a window's first CTX lanes are context (emitted by the window before, re-read so that every lane
// sees its true neighbours), its last two are look-ahead.
struct RowsEmuStats { uint64_t tiles = 0, windows = 0, max_tile_desc = 0; };
inline void rows_emulate(const TxStreamView& s, uint64_t proteome_len, int mode, uint32_t K, RowsImage& im, std::vector<uint64_t>* cover_out = nullptr, RowsEmuStats* stats = nullptr)
{
    im = RowsImage();
    const uint32_t CTX = mode == ROWS_DENSE ? 4u : 2u;
    const uint64_t n_tiles = s.n_tx ? (s.n_tx + K - 1) / K : 1;
    // P0: arena bytes per tile -> exclusive prefix
    std::vector<uint64_t> tile_res_base(n_tiles + 1, 0);
    for (uint64_t t = 0; t < s.n_tx; ++t) tile_res_base[t / K + 1] += rows_arena_len(s, t);
    for (uint64_t k = 0; k < n_tiles; ++k) tile_res_base[k + 1] += tile_res_base[k];
    im.out_bytes = tile_res_base[n_tiles];
    const uint64_t n_rows = (im.out_bytes + ROW_BYTES - 1) / ROW_BYTES;
    std::vector<uint64_t> cover(n_rows ? n_rows : 1, ~0ull);         // [r] = global descriptor index << 22 | offset inside it
    // hap_out_begin (hap_begin kernel): res_base of the haplotype's first transcript
    {
        im.hap_out_begin.assign(s.n_haps + 1, 0);
        for (uint64_t h = 0; h <= s.n_haps; ++h) {
            const uint64_t t = h < s.n_haps ? s.hap_tx_begin[h] : s.n_tx;
            const uint64_t tile = t / K;
            uint64_t b = tile_res_base[tile < n_tiles ? tile : n_tiles];
            if (tile < n_tiles) for (uint64_t u = tile * K; u < t; ++u) b += rows_arena_len(s, u);
            im.hap_out_begin[h] = b;
        }
    }
    std::vector<uint64_t>& d = im.desc;
    for (uint64_t tile = 0; tile < n_tiles; ++tile) {
        const uint64_t t0 = tile * K;
        const uint32_t nh = uint32_t(s.n_tx - t0 < K ? s.n_tx - t0 : K);
        // per-transcript table and the tile's items
        struct Tx { uint64_t poff = 0, alt0 = 0, res_base = 0, hsrc = 0; uint32_t ref_len = 0, res_len = 0, n_alt = 0, hl = 0; };
        struct Item { uint32_t j; bool first, lastp, empty; uint64_t task; };
        std::vector<Tx> tx(nh ? nh : 1);
        std::vector<Item> items;
        {
            uint64_t rb = tile_res_base[tile];
            for (uint32_t j = 0; j < nh; ++j) {
                const uint64_t u = t0 + j;
                Tx& x = tx[j];
                x.poff = s.tx_proteome_off[u]; x.alt0 = s.tx_alt_begin[u]; x.n_alt = uint32_t(s.tx_alt_begin[u + 1] - x.alt0);
                x.ref_len = s.tx_ref_len[u]; x.res_len = s.tx_res_len[u]; x.hl = s.tx_header_len ? s.tx_header_len[u] : 0u;
                x.hsrc = x.hl ? proteome_len + s.tx_header_off[u] : 0ull;
                x.res_base = rb; rb += rows_arena_len(s, u);
                if (x.poff + x.ref_len > proteome_len) rows_report(im, s.tx_task_begin[u], ROWS_SRC_OOB);
                const uint64_t i0 = s.tx_task_begin[u], i1 = s.tx_task_begin[u + 1];
                if (i0 == i1) items.push_back(Item{j, true, true, true, 0});
                else for (uint64_t i = i0; i < i1; ++i) items.push_back(Item{j, i == i0, i + 1 == i1, false, i});
            }
        }
        const uint64_t I1 = items.size();
        uint64_t carry_e = 0, carry_h = 0, carry_second = 0;
        const uint64_t tile_desc0 = d.size();
        bool first = true;
        for (uint64_t R0 = 0; ; ) {
            const uint32_t nvalid = uint32_t(I1 - R0 < 64 ? I1 - R0 : 64);
            const bool last = R0 + 64 >= I1;
            const uint32_t e_lo = first ? 0u : CTX, e_hi = last ? nvalid : 62u;
            // ---- per lane ----
            bool isFirst[64] = {}, isLast[64] = {}, isEmpty[64] = {}, isTask[64] = {};
            uint32_t jj[64] = {};                                     // the lane's transcript inside the tile
            uint64_t ti[64] = {}, sp[64] = {}, ln[64] = {}, sr[64] = {}, e[64] = {}, pe[64] = {}, src[64] = {}, lit[64] = {};
            uint32_t code[64] = {};
            bool isRef[64] = {}, imm[64] = {}, bad[64] = {};
            for (uint32_t l = 0; l < nvalid; ++l) {
                const Item& it = items[R0 + l];
                jj[l] = it.j; isFirst[l] = it.first; isLast[l] = it.lastp; isEmpty[l] = it.empty; isTask[l] = !it.empty;
                if (isTask[l]) {
                    const uint64_t i = it.task;
                    ti[l] = i; code[l] = s.code[i]; sp[l] = s.start_pos[i]; ln[l] = s.length[i]; sr[l] = s.start_pos_res[i];
                    if (sr[l] + ln[l] <= tx[it.j].res_len) e[l] = sr[l] + ln[l];
                }
            }
            for (uint32_t l = 0; l < nvalid; ++l) pe[l] = isFirst[l] ? 0 : (l ? e[l - 1] : carry_e);
            uint64_t mA = 0, mB = 0, mPS = 0, mC0 = 0, mC1 = 0, mRst = 0;
            uint64_t gapfill[64] = {};
            for (uint32_t l = 0; l < 64; ++l) {
                if (l >= nvalid || !isTask[l]) { mRst |= 1ull << l; continue; }
                const Tx& x = tx[jj[l]];
                uint32_t why = 0;
                if (code[l] > 1u) why = ROWS_BAD_CODE;
                else if (sr[l] + ln[l] > x.res_len) why = ROWS_RES_OOB;
                else if (sp[l] + ln[l] > (code[l] == 0 ? uint64_t(x.ref_len) : uint64_t(x.n_alt))) why = ROWS_SRC_OOB;
                else if (sr[l] < pe[l]) why = ROWS_NOT_CONTIGUOUS;
                if (why) { if (l >= e_lo && l < e_hi) rows_report(im, ti[l], why); bad[l] = true; mRst |= 1ull << l; continue; }
                const bool gap = sr[l] > pe[l];
                if (gap) gapfill[l] = sr[l] - pe[l];
                if (isFirst[l] || gap) mRst |= 1ull << l;             // (a later window's lane 0 is context: its reset bit is never used)
                isRef[l] = code[l] == 0;
                imm[l] = code[l] == 1 && ln[l] >= 1 && ln[l] <= IMM_MAX_BYTES;
                src[l] = isRef[l] ? x.poff + sp[l] : x.alt0 + sp[l];
                if (imm[l]) { uint64_t v = 0; for (uint64_t k = 0; k < ln[l]; ++k) v |= uint64_t(s.alt[x.alt0 + sp[l] + k]) << (8 * k); lit[l] = v; }
                const bool ps = isRef[l] && ln[l] <= ROWS_FUSE_LEN;
                if (isRef[l] && ln[l] <= ROWS_FUSE_LEN && src[l] + ln[l] + 1 + ROWS_FUSE_LEN <= SNV3_MAX_SRC) mA |= 1ull << l;
                if (imm[l] && ln[l] == 1) mB |= 1ull << l;
                if (ps) mPS |= 1ull << l;
                const bool c0 = ps && ln[l] > 0 && src[l] >= 1 && src[l] + ln[l] <= SNV3_MAX_SRC;
                if (c0) mC0 |= 1ull << l;
                if (l >= 2) {
                    const bool c1 = ln[l - 2] == 0 ? c0 : (ps && (ln[l] == 0 || src[l] == src[l - 2] + ln[l - 2] + 1));
                    if (c1) mC1 |= 1ull << l;
                }
            }
            const RowsParse p = rows_parse(mA, mB, mPS, mC0, mC1, mRst, !first, carry_h);
            // fused substitutions: every closing lane has its own (run, len1, byte, len2)
            uint64_t f_run[64] = {}; uint32_t f_len1[64] = {}, f_byte[64] = {}, f_len2[64] = {};
            for (uint32_t l = 1; l < 64; ++l) if ((p.F >> l) & 1) {
                const bool real = (p.real >> l) & 1;
                f_len1[l] = real ? uint32_t(ln[l - 2]) : 0u;
                f_byte[l] = uint32_t(lit[l - 1] & 0xFF);
                f_len2[l] = uint32_t(ln[l]);
                f_run[l] = f_len1[l] == 0 ? src[l] - 1 : src[l - 2];
            }
            uint64_t second = 0;
            if (mode == ROWS_DENSE) {
                uint64_t L = 0;
                for (uint32_t l = 2; l < 64; ++l)
                    if (((p.F >> l) & 1) && ((p.F >> (l - 2)) & 1) && !((mRst >> (l - 1)) & 1) && f_len1[l] == 0 && f_len1[l - 2] <= SNV5_MAX_LEN && f_len2[l - 2] <= SNV5_MAX_LEN &&
                        f_len2[l] <= SNV5_MAX_LEN && f_run[l] == f_run[l - 2] + f_len1[l - 2] + 1 + f_len2[l - 2]) L |= 1ull << l;
                second = rows_pair(L, !first, carry_second);
            }
            const uint64_t absorbed = (p.F >> 1) | ((p.F & p.real) >> 2) | (mode == ROWS_DENSE ? (second >> 2) & p.F : 0ull);
            // ---- emission, lane by lane (the device: one scan of the lanes' descriptor counts) ----
            for (uint32_t l = e_lo; l < e_hi; ++l) {
                const Tx& x = tx[jj[l]];
                auto put = [&](uint64_t pos, unsigned space, uint64_t sr_, uint64_t len) {       // plain descriptor(s) at arena position pos
                    uint64_t at = pos;
                    while (len) {
                        const uint32_t piece = uint32_t(len < PIECE_MAX ? len : PIECE_MAX);
                        const uint64_t k = d.size();
                        d.push_back(pack_desc(sr_, piece, space));
                        for (uint64_t r = (at + ROW_BYTES - 1) / ROW_BYTES; r * ROW_BYTES < at + piece; ++r) if (r >= 1) cover[r] = (k << 22) | (r * ROW_BYTES - at);
                        sr_ = ImageBuilder::advance(space, sr_, piece); len -= piece; at += piece;
                    }
                };
                auto put1 = [&](uint64_t pos, uint64_t word, uint64_t len) {                       // one ready descriptor
                    const uint64_t k = d.size();
                    d.push_back(word);
                    for (uint64_t r = (pos + ROW_BYTES - 1) / ROW_BYTES; r * ROW_BYTES < pos + len; ++r) if (r >= 1) cover[r] = (k << 22) | (r * ROW_BYTES - pos);
                };
                const uint64_t posbase = x.res_base + x.hl;
                if (isFirst[l] && x.hl) put(x.res_base, SPACE_PROTEOME, x.hsrc, x.hl);
                auto close = [&]() {                                  // the transcript's last item: '.' tail, line feed
                    if (!isLast[l]) return;
                    if (x.res_len > e[l]) put(posbase + e[l], SPACE_FILL, 0, x.res_len - e[l]);
                    if (x.hl) put(posbase + x.res_len, SPACE_PROTEOME, x.hsrc + x.hl - 1u, 1);
                };
                if (isEmpty[l]) { close(); continue; }
                if (bad[l]) continue;
                if (gapfill[l]) put(posbase + pe[l], SPACE_FILL, 0, gapfill[l]);
                if ((p.F >> l) & 1) {
                    if (!((absorbed >> l) & 1)) {                       // (dense: the first of a pair is absorbed)
                        if ((second >> l) & 1) {
                            const uint32_t a1 = f_len1[l - 2], a2 = f_len2[l - 2];
                            put1(posbase + sr[l] - 1 - a2 - 1 - a1, pack_snv5(f_run[l - 2], a1, uint8_t(f_byte[l - 2]), a2, uint8_t(f_byte[l]), f_len2[l]), uint64_t(a1) + 1 + a2 + 1 + f_len2[l]);
                        } else put1(posbase + sr[l] - 1 - f_len1[l], pack_snv3(f_run[l], f_len1[l], f_len2[l], uint8_t(f_byte[l])), uint64_t(f_len1[l]) + 1 + f_len2[l]);
                    }
                } else if (!((absorbed >> l) & 1) && ln[l] != 0) {
                    if (imm[l]) put(posbase + sr[l], SPACE_IMM, lit[l], ln[l]);
                    else put(posbase + sr[l], isRef[l] ? SPACE_PROTEOME : SPACE_PAYLOAD, src[l], ln[l]);
                }
                close();
            }
            if (stats) ++stats->windows;
            if (last) break;
            // next window: starts CTX lanes before this window's first look-ahead lane
            const uint32_t adv = 62u - CTX;
            carry_h = p.h >> adv; carry_second = second >> adv;
            carry_e = e[adv - 1];
            R0 += adv;
            first = false;
        }
        if (stats) { ++stats->tiles; if (d.size() - tile_desc0 > stats->max_tile_desc) stats->max_tile_desc = d.size() - tile_desc0; }
    }
    if (cover_out) *cover_out = cover;
}

// The cutter.  `cover`: per row r >= 1 the descriptor covering byte r * 1024 (index << 22 | offset), as the parse records it -- or
// empty: derived here from the descriptors' lengths.  Greedy from each segment's first row: the most rows (<= max_rows) whose
// descriptors fit max_desc.  Returns false when a single row holds more descriptors than that.
inline bool rows_cut(RowsImage& im, int mode, const std::vector<uint64_t>* cover_in = nullptr)
{
    const uint32_t max_rows = mode == ROWS_DENSE ? ROWS_MAX_DENSE : ROWS_MAX_WAVE, max_desc = mode == ROWS_DENSE ? CHUNK_TASKS_DEEP : CHUNK_TASKS_WAVE;
    const uint64_t flag = mode == ROWS_DENSE ? CHUNK_DENSE : CHUNK_WAVE;
    const uint64_t n_desc = im.desc.size(), n_rows = (im.out_bytes + ROW_BYTES - 1) / ROW_BYTES;
    std::vector<uint64_t> cover;
    if (cover_in) cover = *cover_in;
    else {
        cover.assign(n_rows ? n_rows : 1, ~0ull);
        uint64_t at = 0;
        for (uint64_t k = 0; k < n_desc; ++k) {
            const uint64_t len = desc_len(im.desc[k]);
            for (uint64_t r = (at + ROW_BYTES - 1) / ROW_BYTES; r * ROW_BYTES < at + len; ++r) if (r >= 1) cover[r] = (k << 22) | (r * ROW_BYTES - at);
            at += len;
        }
    }
    im.chunks.clear();
    auto first_of = [&](uint64_t r, uint32_t& hs) { if (r == 0) { hs = 0; return uint64_t(0); } hs = uint32_t(cover[r] & 0x3FFFFFu); return cover[r] >> 22; };
    auto last_of = [&](uint64_t r) { if (r >= n_rows) return n_desc - 1; return (cover[r] & 0x3FFFFFu) ? cover[r] >> 22 : (cover[r] >> 22) - 1; };
    for (uint64_t seg = 0; seg < n_rows; seg += ROWS_SEG) {
        const uint64_t seg_end = seg + ROWS_SEG < n_rows ? seg + ROWS_SEG : n_rows;
        for (uint64_t r0 = seg; r0 < seg_end; ) {
            uint32_t hs;
            const uint64_t f = first_of(r0, hs);
            uint64_t r1 = r0 + max_rows < seg_end ? r0 + max_rows : seg_end;
            while (r1 > r0 + 1 && last_of(r1) - f + 1 > max_desc) --r1;
            const uint64_t lastd = last_of(r1), n = lastd - f + 1;
            if (n > max_desc) { rows_report(im, f, ROWS_TOO_MANY); return false; }
            // what the chunk's last descriptor has behind the cut
            uint32_t tc = 0;
            if (r1 < n_rows && (cover[r1] & 0x3FFFFFu)) tc = desc_len(im.desc[lastd]) - uint32_t(cover[r1] & 0x3FFFFFu);
            im.chunks.push_back(Chunk{f | (uint64_t(hs) << TB_IDX_BITS) | (uint64_t(tc) << (TB_IDX_BITS + TB_SKIP_BITS)), (r0 * ROW_BYTES) | (n << 48) | CHUNK_CLIP | flag});
            r0 = r1;
        }
    }
    return true;
}
#endif  // host side

}  // namespace v2p
