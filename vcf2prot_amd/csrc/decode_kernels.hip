// decode_kernels.hip -- BCSQ bitmask decode on gfx950.
//
// What the reference does per proband (vcf_ds.rs:192-329): split every record line at tabs, take the proband's
// column, keep the text after its last ':' (text_parser.rs:163-197), read it as one or more 32-bit words
// (MaskDecoder.rs:33-51), turn bit pairs into consequence indices (MaskDecoder.rs:95-153), index the record's
// consequence list (vcf_ds.rs:311-327) and keep supported types (vcf_ds.rs:262-292).  Result: per haplotype the list of
// consequences in record order.  Here the whole table is done in four passes, no MFMA:
//
//   parse   one workgroup per record.  The record's sample columns stream through a two-tile LDS ring, 16 B per lane
//           per step; tab positions are found with byte SWAR, ranked with a wave64 DPP scan, and lane j then owns the
//           j-th column that ENDS in the tile: it walks back to the last ':' and parses the words.  Output: the record's
//           CARRIERS -- (sample, filtered first word) of every column whose mask is not empty, compacted per wave into
//           the record's row of the carrier table.  Real cohorts are sparse (a few percent of the columns carry anything),
//           so the table costs a tenth of the text to write and to read back, where a dense matrix costs two thirds of it.
//           Multi-word masks (records with more than 15 consequences) go to a side list.
//   count   one workgroup per 64-record block: LDS histogram of the block's carriers per haplotype.
//   scan    exclusive prefix down the record blocks per haplotype, then over haplotypes: every (block, haplotype)
//           knows where its ids go.
//   emit    one workgroup per 64-record block with a write cursor per haplotype in LDS.  Records are taken in order, one
//           step each; inside a record every sample appears once, so thread k places carrier k with no conflict and an
//           LDS-only barrier separates the records.
#include "decode_kernels.h"
#include <cstdlib>

namespace v2p {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t dec_wave_incl_scan(uint32_t x)
{
    x += __builtin_amdgcn_update_dpp(0u, x, 0x111, 0xf, 0xf, true);   // row_shr:1 (lanes shifted in from outside the row read 0)
    x += __builtin_amdgcn_update_dpp(0u, x, 0x112, 0xf, 0xf, true);   // row_shr:2
    x += __builtin_amdgcn_update_dpp(0u, x, 0x114, 0xf, 0xf, true);   // row_shr:4
    x += __builtin_amdgcn_update_dpp(0u, x, 0x118, 0xf, 0xf, true);   // row_shr:8
    x += __builtin_amdgcn_update_dpp(0u, x, 0x142, 0xa, 0xf, false);  // row_bcast:15
    x += __builtin_amdgcn_update_dpp(0u, x, 0x143, 0xc, 0xf, false);  // row_bcast:31
    return x;
}

__device__ __forceinline__ void dec_report(unsigned long long* status, uint64_t field, uint32_t reason)
{
    atomicMin(status, (static_cast<unsigned long long>(field) << 8) | reason);
}

// Workgroup barrier that orders LDS only.  __syncthreads() also waits for vmcnt(0): in the parse kernel that is the NEXT tile's
// prefetch (a full HBM latency at every barrier) and the carrier stores; in the emit kernel every id store in flight.
__device__ __forceinline__ void dec_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// 8-bit mask (<< 7) of the bytes of the word pair (w0, w1) equal to '\t': exact 0x80 flags per byte, then one dot product per
// word weighs byte k with 2^k
__device__ __forceinline__ uint32_t tab_bits8(uint32_t w0, uint32_t w1)
{
    const uint32_t y0 = w0 ^ 0x09090909u, y1 = w1 ^ 0x09090909u;
    const uint32_t z0 = ~(((y0 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y0 | 0x7F7F7F7Fu);
    const uint32_t z1 = ~(((y1 & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y1 | 0x7F7F7F7Fu);
    return __builtin_amdgcn_udot4(z1, 0x80402010u, __builtin_amdgcn_udot4(z0, 0x08040201u, 0u, false), false);
}

// 0x80 in every byte of w that equals the byte replicated in pat, exact
__device__ __forceinline__ uint32_t eq_bytes(uint32_t w, uint32_t pat)
{
    const uint32_t y = w ^ pat;
    return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu);
}

struct Num { bool valid; bool neg; uint64_t val; };


// ring buffer of 2 tiles; RM = its size - 1
template <uint32_t RM>
__device__ __forceinline__ uint8_t rq(const uint8_t* ring, uint32_t q) { return ring[q & RM]; }

// Rust integer from_str on txt[b, e): optional sign, at least one digit, digits only (value saturates far above u32)
__device__ __forceinline__ Num parse_num(const uint8_t* txt, uint32_t b, uint32_t e)
{
    Num n{false, false, 0};
    if (b < e) {
        const uint8_t c = txt[b];
        if (c == '+' || c == '-') { n.neg = (c == '-'); ++b; }
    }
    if (b >= e) return n;
    uint64_t v = 0;
    for (uint32_t q = b; q < e; ++q) {
        const uint32_t d = uint32_t(txt[q]) - uint32_t('0');
        if (d > 9u) return n;
        v = v * 10u + d;
        if (v > (1ull << 40)) v = 1ull << 40;
    }
    n.valid = true;
    n.val = v;
    return n;
}

// text_parser::parse_fields + BitMask::from_string on one element: 0 = no consequences; aborts reported through err
__device__ __forceinline__ uint32_t single_word(const uint8_t* txt, uint32_t b, uint32_t e, uint32_t& err)
{
    const Num n = parse_num(txt, b, e);
    if (!n.valid) return 0u;                                   // parse::<i32>() Err -> DEF_CONSEQ (text_parser.rs:216)
    if (n.neg) {
        if (n.val > (1ull << 31)) return 0u;                   // below i32::MIN: Err as well
        err = n.val ? DEC_MASK_NEGATIVE : DEC_MASK_PARSE;      // "-5" panics at text_parser.rs:210, "-0" at MaskDecoder.rs:41
        return 0u;
    }
    return n.val <= 0x7FFFFFFFull ? uint32_t(n.val) : 0u;
}

__device__ __forceinline__ uint32_t top_pair(uint32_t w) { return (31u - uint32_t(__builtin_clz(w))) >> 1; }

// keep the pairs of w whose consequence (ids base .. base+15) is supported
__device__ __forceinline__ uint32_t filter_word(uint32_t w, uint32_t base, const uint32_t* __restrict__ sup_bits)
{
    uint32_t out = 0u, pairs = (w | (w >> 1)) & 0x55555555u;
    while (pairs) {
        const uint32_t b = uint32_t(__builtin_ctz(pairs));
        pairs &= pairs - 1u;
        const uint32_t id = base + (b >> 1);
        if ((sup_bits[id >> 5] >> (id & 31u)) & 1u) out |= w & (3u << b);
    }
    return out;
}

// ---------------------------------------------------------------------------------------------------------- parse
// BS threads per record: 256 for wide cohorts, fewer when a record's sample columns are shorter than a 4 KiB tile.
//
// Round 5: the columns that need the parser (5 % of a cohort's) are no longer parsed step by step.  Until now every 4 KiB step ended
// with ONE wave working a list of ~34 columns through the ~300-instruction parser while the step's other waves waited at the
// barrier -- half of the kernel's time.  Now a column that is not settled by its last two bytes is only NOTED when it is found -- sample
// and stream position, eight bytes of LDS -- in a list of the whole record, and the list is parsed ONCE, behind the record's last
// step, with every lane of the workgroup busy (a record of 2 504 samples: ~140 entries).  Two barriers per step instead of three, no
// parser between them, and what a step does for the rare columns is a dozen instructions (any code a wave runs for ONE lane costs it
// as much as for 64: capturing the column's last eight bytes from the ring there -- the first form of this -- cost more than it saved).
// The parser reads the column's last eight bytes from the text in memory (one unaligned load per entry; the record has just been
// read) and they settle what the fast path settles (a tail of at most seven digits, '.', no ':' in reach); anything else -- long
// numbers, comma lists, multi-word masks -- goes back through the text byte by byte, which therefore no longer depends on what the
// ring still holds: the tail of a column is searched for its ':' over exactly 4 KiB.
template <uint32_t BS>
__global__ __launch_bounds__(BS) void parse_rows_kernel(DecodeArgs a)
{
    constexpr uint32_t TILE = BS * 16u, RING = 2u * TILE, RM = RING - 1u, NW = BS / 64u;
    constexpr uint32_t CAP = 8u * BS;                           // entries the record's list holds (flushed early when it fills)
    constexpr uint32_t TAIL_MAX = 4096u;                        // V2P_ERR_FIELD_TOO_LONG: no ':' / tab within this many bytes before the column's end
    __shared__ __align__(16) uint8_t ring[RING];
    __shared__ uint32_t e_f[CAP], e_pos[CAP];                         // sample, stream position one past the column
    __shared__ uint32_t s_cnt[2];                                     // entries found in this step (double-buffered by step parity)
    __shared__ uint32_t wave_tot[NW];
    __shared__ uint32_t s_nnz;

    const uint32_t row = blockIdx.x;
    const uint32_t tid = threadIdx.x, wave = tid >> 6;
    const uint64_t rb = a.row_begin[row], re = a.row_end[row];
    const uintptr_t first = reinterpret_cast<uintptr_t>(a.text) + rb;
    const uintptr_t base = first & ~uintptr_t(15);
    const uint8_t* const gtext = reinterpret_cast<const uint8_t*>(base);     // stream position q <-> gtext[q], q0 <= q < Lq
    const uint32_t q0 = uint32_t(first - base);                 // stream position of the first row byte
    const uint32_t Lq = uint32_t(re - rb) + q0;                 // stream position one past the last row byte
    const uint32_t n_tiles = Lq ? (Lq + TILE - 1u) / TILE : 1u;
    const uint32_t c0 = a.csq_begin[row], n_csq = a.csq_begin[row + 1] - c0;
    const uint32_t sup = a.sup_pairs[row];
    const uint64_t field0 = uint64_t(row) * a.n_samples;
    uint32_t fields_before = 0;
    DecCarrier* const carriers = a.carriers + uint64_t(row) * a.n_samples;
    if (tid == 0) { s_nnz = 0u; s_cnt[0] = 0u; s_cnt[1] = 0u; }      // (a barrier before the first append)

    auto load_tile = [&](uint32_t t) -> u32x4 {
        const uint32_t qs = t * TILE + tid * 16u;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (qs < Lq) v = *reinterpret_cast<const u32x4 __attribute__((address_space(1)))*>(base + qs);
        return v;
    };

    // ---- the parser: entries [0, n) of the record's list, lane = entry; carriers appended to the record's row ----
    auto flush = [&](uint32_t n) {
        for (uint32_t j = tid; j < n; j += BS) {
            const uint32_t f = e_f[j], p = e_pos[j];
            uint32_t q = p, err = 0u, entry = 0u;
            bool colon = false, slow = true;
            if (p >= q0 + 8u) {
                // fast path: the last eight bytes of the column, one unaligned load from the text (the record has just been read: L2).
                // Settles every column whose text after the last ':' is at most seven digits (or '.'), and every column without a
                // ':' that starts inside the window.
                struct __attribute__((packed, aligned(1))) U64 { uint64_t v; };
                const uint64_t w8 = reinterpret_cast<const U64 __attribute__((address_space(1)))*>(base + (p - 8u))->v;
                const uint32_t lo32 = uint32_t(w8), hi32 = uint32_t(w8 >> 32);
                // Straight-line: the parser is the heaviest part of the kernel (a third of its VALU work before this form).
                const uint32_t c_hi = eq_bytes(hi32, 0x3A3A3A3Au), c_lo = eq_bytes(lo32, 0x3A3A3A3Au);
                const uint32_t d_hi = c_hi | eq_bytes(hi32, 0x09090909u), d_lo = c_lo | eq_bytes(lo32, 0x09090909u);   // ':' or tab, 0x80 per byte
                const bool in_hi = d_hi != 0u;
                const uint32_t dsel = in_hi ? d_hi : d_lo, csel = in_hi ? c_hi : c_lo;   // the word that holds the window's last delimiter
                const bool any = dsel != 0u;
                const uint32_t top = 31u - uint32_t(__builtin_clz(dsel | 1u));         // bit of its flag (7, 15, 23, 31)
                const bool is_colon = any && ((csel >> top) & 1u) != 0u;
                const uint32_t bi = (top >> 3) + (in_hi ? 4u : 0u);                     // its byte in the window, 0..7
                // the bytes behind it as digit values, everything up to the delimiter as leading zeros
                const uint64_t X = (uint64_t(hi32) << 32) | lo32;
                const uint64_t D = (X ^ 0x3030303030303030ull) & ((~0ull << (8u * bi)) << 8);
                const uint32_t Dl = uint32_t(D), Dh = uint32_t(D >> 32);
                const bool digits = (((((Dl & 0x7F7F7F7Fu) + 0x76767676u) | Dl) | (((Dh & 0x7F7F7F7Fu) + 0x76767676u) | Dh)) & 0x80808080u) == 0u;   // every byte <= 9
                // eight decimal digits, the first character the most significant: pairs, then fours
                const uint32_t tl = ((Dl << 3) + (Dl << 1)) + (Dl >> 8), th = ((Dh << 3) + (Dh << 1)) + (Dh >> 8);
                const uint32_t vl = (tl & 0xFFu) * 100u + ((tl >> 16) & 0xFFu), vh = (th & 0xFFu) * 100u + ((th >> 16) & 0xFFu);
                const uint32_t value = vl * 10000u + vh;                                // < 10^7: a valid i32
                const bool no_colon = any && !is_colon;                                 // the column starts after the window's last ':': nothing
                const bool dot = is_colon && bi == 6u && (hi32 >> 24) == uint32_t('.');
                const bool number = is_colon && digits;                                 // (no digits at all: "" -> nothing)
                entry = number ? value : 0u;
                slow = !(no_colon || dot || number);
            }
            if (slow) {
                // the column's text in memory: back to its last ':' (or to the tab / the row's first byte: no ':' at all)
                const uint32_t lo = max(q0, p > TAIL_MAX ? p - TAIL_MAX : 0u);
                while (q > lo) {
                    const uint8_t c = gtext[q - 1u];
                    if (c == ':') { colon = true; break; }
                    if (c == '\t') break;
                    --q;
                }
                if (!colon && q == lo && lo > q0) err = DEC_FIELD_TOO_LONG;
            }
            if (colon) {
                const uint32_t s = q;                                          // tail = gtext[s, p)
                const uint32_t len = p - s;
                if (len == 0u || (len == 1u && gtext[s] == '.')) {
                    entry = 0u;                                                // "" parses to Err, "." is the missing value
                } else {
                    // elements, how many survive remove_leading_zeros (it strips trailing "0" elements), any '-'
                    uint32_t n_el = 1u, kept = 0u, es = s, first_end = p;
                    bool minus = false;
                    for (uint32_t k = s; k < p; ++k) {
                        const uint8_t c = gtext[k];
                        if (c == ',') {
                            if (!(k - es == 1u && gtext[es] == '0')) kept = n_el;
                            if (n_el == 1u) first_end = k;
                            ++n_el;
                            es = k + 1u;
                        } else if (c == '-') {
                            minus = true;
                        }
                    }
                    if (!(p - es == 1u && gtext[es] == '0')) kept = n_el;
                    if (n_el == 1u) {
                        entry = single_word(gtext, s, p, err);                     // text_parser.rs:179-182
                    } else if (kept == 0u) {
                        entry = 0u;                                            // "0,0" (text_parser.rs:240-243)
                    } else if (minus) {
                        err = DEC_MASK_NEGATIVE;                               // text_parser.rs:244
                    } else if (kept == 1u) {
                        entry = single_word(gtext, s, first_end, err);             // "x,0" falls back to parse_fields (text_parser.rs:189-192)
                    } else {
                        // MaskDecoder.rs:45-50: every kept element must be a u32; word k covers indices 15k .. 15k+15
                        uint32_t any = 0u;
                        bool bad_index = false;                                // every word is parsed before any is used as an index
                        for (int pass = 0; pass < 2 && !err; ++pass) {
                            uint32_t off = 0u;
                            if (pass == 1) {
                                if (!any) break;
                                const unsigned long long o = atomicAdd(a.ovf_used, static_cast<unsigned long long>(kept + 1u));
                                if (o + kept + 1u > a.ovf_capacity) { err = DEC_CAPACITY; break; }
                                off = uint32_t(o);
                                a.ovf[off] = kept;
                                entry = DEC_MULTI | off;
                            }
                            uint32_t eb = s, k = 0u;
                            for (uint32_t x = s; x <= p && k < kept; ++x) {
                                if (x == p || gtext[x] == ',') {
                                    const Num n = parse_num(gtext, eb, x);
                                    if (!n.valid || n.neg || n.val > 0xFFFFFFFFull) { err = DEC_MASK_PARSE; break; }
                                    const uint32_t w = uint32_t(n.val);
                                    const bool oob = w && 15u * k + top_pair(w) >= n_csq;
                                    bad_index |= oob;
                                    const uint32_t fw = (w && !oob) ? filter_word(w, c0 + 15u * k, a.sup_bits) : 0u;
                                    if (pass == 0) any |= fw; else a.ovf[off + 1u + k] = fw;
                                    ++k;
                                    eb = x + 1u;
                                }
                            }
                            if (!err && bad_index) err = DEC_MASK_INDEX;
                        }
                    }
                }
            }
            if (!err && entry && !(entry & DEC_MULTI)) {
                if (top_pair(entry) >= n_csq) err = DEC_MASK_INDEX;            // vcf_ds.rs:321: splitted_csq[idx] out of range
                entry &= sup;
            }
            if (f >= a.n_samples) err = err ? err : DEC_COLUMNS;
            if (err) dec_report(a.status, field0 + min(f, a.n_samples - 1u), err);
            // append the carriers to the record's row: one LDS atomic per wave, order inside a record is free
            const bool keep = !err && entry != 0u;
            const uint64_t kb = __builtin_amdgcn_ballot_w64(keep);
            if (kb) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi(uint32_t(kb >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(kb), 0u));
                uint32_t at = 0u;
                if (keep && rank == 0u) at = atomicAdd(&s_nnz, uint32_t(__builtin_popcountll(kb)));
                at = uint32_t(__builtin_amdgcn_readlane(int(at), int(__builtin_ctzll(kb))));
                if (keep) carriers[at + rank] = DecCarrier{f, entry};
            }
        }
    };

    uint32_t pending = 0;                                       // entries in the record's list (uniform)
    u32x4 cur = load_tile(0);
    for (uint32_t t = 0; t < n_tiles; ++t) {
        const uint32_t tile0 = t * TILE;
        *reinterpret_cast<u32x4*>(&ring[(tile0 & (RING - 1u)) + tid * 16u]) = cur;
        u32x4 nxt = {0u, 0u, 0u, 0u};
        if (t + 1u < n_tiles) nxt = load_tile(t + 1u);

        // tabs among this lane's 16 bytes that belong to the row
        const uint32_t qs = tile0 + tid * 16u;
        uint32_t tm = (tab_bits8(cur.x, cur.y) >> 7) | ((tab_bits8(cur.z, cur.w) >> 7) << 8);
        {
            const uint32_t lo = q0 > qs ? min(q0 - qs, 16u) : 0u;
            const uint32_t hi = Lq > qs ? min(Lq - qs, 16u) : 0u;
            tm &= ((1u << hi) - 1u) & ~((1u << lo) - 1u);
        }
        const bool last = (t + 1u == n_tiles);
        if (last) {
            // the end of the line closes the last column: one more "tab", at Lq -- bit Lq - qs of the lane whose sixteen bytes hold
            // that position (bit 16 of the last lane when the line ends with the tile)
            const uint32_t te = min((Lq - tile0) >> 4, BS - 1u);
            if (tid == te) tm |= 1u << (Lq - tile0 - te * 16u);
        }
        const uint32_t cnt = uint32_t(__builtin_popcount(tm));
        const uint32_t incl = dec_wave_incl_scan(cnt);
        if ((tid & 63u) == 63u) wave_tot[wave] = incl;
        dec_lds_barrier();
        if (tid == 0) s_cnt[(t + 1u) & 1u] = 0u;                               // (the next step's counter: its last readers are a barrier behind)
        uint32_t wbase = 0, tile_ends = 0;
#pragma unroll
        for (uint32_t w = 0; w < NW; ++w) {
            const uint32_t x = wave_tot[w];
            if (w < wave) wbase += x;
            tile_ends += x;
        }
        const uint32_t lo = max(q0, t ? tile0 - TILE : 0u);               // oldest stream position still in the ring
        // Almost every column of a real cohort carries nothing: its text ends in ":0" or ":.".  Those are settled here with two byte
        // reads per tab, in a loop without ballots or atomics; only the others are captured for the parser, so its cost scales with
        // the carriers and not with the columns.
        auto tab_pos = [&](uint32_t b) -> uint32_t { return tile0 + tid * 16u + b; };
        const uint32_t tm0 = tm, slot0 = wbase + incl - cnt;
        uint32_t wm = 0u;                                                       // the lane's column ends that need the parser
        {
            uint32_t tt = tm;
            while (tt) {
                const uint32_t b = uint32_t(__builtin_ctz(tt));
                tt &= tt - 1u;
                const uint32_t p = tab_pos(b);
                const uint8_t c1 = rq<RM>(ring, p - 1u), c2 = rq<RM>(ring, p - 2u);
                const bool empty = p >= lo + 2u && c2 == ':' && (c1 == '0' || c1 == '.');
                wm |= empty ? 0u : 1u << b;
            }
        }
        const uint32_t wm0 = wm;
        const uint32_t nw = uint32_t(__builtin_popcount(wm));
        uint32_t k0 = 0u;                                                       // the lane's first entry among the step's
        if (__builtin_amdgcn_ballot_w64(nw != 0u)) {
            const uint32_t wincl = dec_wave_incl_scan(nw);
            uint32_t at = 0u;
            if ((tid & 63u) == 63u) at = atomicAdd(&s_cnt[t & 1u], wincl);
            at = uint32_t(__builtin_amdgcn_readlane(int(at), 63));
            k0 = at + wincl - nw;
        }
        // noted: sample and position.  A step that finds more entries than the list has room for flushes and goes round again.
        uint32_t done = 0u;                                                     // entries of this step already in the list or parsed (uniform)
        for (;;) {
            uint32_t w = wm0, k = k0;
            while (w) {
                const uint32_t b = uint32_t(__builtin_ctz(w));
                w &= w - 1u;
                const uint32_t idx = pending + (k - done);
                if (k >= done && idx < CAP) {
                    e_pos[idx] = tab_pos(b);
                    e_f[idx] = fields_before + slot0 + uint32_t(__builtin_popcount(tm0 & ((1u << b) - 1u)));   // (>= n_samples: V2P_ERR_COLUMNS)
                }
                ++k;
            }
            dec_lds_barrier();
            const uint32_t total = s_cnt[t & 1u];
            const uint32_t fit = min(total - done, CAP - pending);
            pending += fit; done += fit;
            if (done == total) break;
            flush(pending);                                                     // (rare: more than CAP non-empty columns between two flushes)
            pending = 0u;
            dec_lds_barrier();
        }
        fields_before += tile_ends;
        cur = nxt;
    }
    flush(pending);
    dec_lds_barrier();
    if (tid == 0) {
        a.row_nnz[row] = s_nnz;
        if (fields_before != a.n_samples) dec_report(a.status, field0 + min(fields_before, a.n_samples - 1u), DEC_COLUMNS);
    }
}

// counts of one matrix entry for haplotype bit h (0/1)
__device__ __forceinline__ uint32_t entry_count(uint32_t m, uint32_t h, const uint32_t* __restrict__ ovf)
{
    if (!(m & DEC_MULTI)) return uint32_t(__builtin_popcount((m >> h) & 0x55555555u));
    const uint32_t off = m & ~DEC_MULTI, n = ovf[off];
    uint32_t c = 0;
    for (uint32_t k = 0; k < n; ++k) c += uint32_t(__builtin_popcount((ovf[off + 1u + k] >> h) & 0x55555555u));
    return c;
}

// ---------------------------------------------------------------------------------------------------------- count
// One workgroup per (64-record block, range of DEC_RANGE_HAPS haplotypes): histogram of the block's carriers in LDS.
__global__ __launch_bounds__(256) void count_kernel(DecodeArgs a, uint32_t n_ranges)
{
    extern __shared__ uint32_t hist[];
    if (a.status[0] != ~0ull) return;                                           // a failed parse leaves the table incomplete
    const uint32_t rbk = blockIdx.x / n_ranges, rg = blockIdx.x % n_ranges;
    const uint32_t n_haps = 2u * a.n_samples, h0 = rg * DEC_RANGE_HAPS, hn = min(DEC_RANGE_HAPS, n_haps - h0);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    __shared__ uint32_t s_pre[DEC_ROWBLOCK + 1];
    constexpr uint32_t G = 4u;                                                  // carriers a lane has in flight together
    const uint32_t r0 = rbk * DEC_ROWBLOCK, r1 = min(r0 + DEC_ROWBLOCK, a.n_rows);
    for (uint32_t i = tid; i < hn; i += 256u) hist[i] = 0u;
    if (tid < 64u) {                                                            // exclusive prefix of the block's 64 record sizes (one wave)
        static_assert(DEC_ROWBLOCK == 64u, "one wave scans the block's record sizes");
        const uint32_t n = r0 + tid < r1 ? a.row_nnz[r0 + tid] : 0u, incl = dec_wave_incl_scan(n);
        s_pre[tid] = incl - n;
        if (tid == 63u) s_pre[64] = incl;
    }
    __syncthreads();
    // the block's carriers as ONE list (a binary search per carrier finds its record): every lane has work whatever the records'
    // sizes, and a lane's G loads are in flight together
    const uint32_t total = s_pre[DEC_ROWBLOCK];
    (void)wave;
    for (uint32_t base = 0; base < total; base += 256u * G) {
        DecCarrier v[G];
#pragma unroll
        for (uint32_t u = 0; u < G; ++u) {
            const uint32_t idx = base + 256u * u + tid;
            uint32_t lo = 0u;                                                   // largest i with s_pre[i] <= idx
#pragma unroll
            for (uint32_t step = DEC_ROWBLOCK / 2u; step; step >>= 1) if (s_pre[lo + step] <= idx) lo += step;
            v[u] = DecCarrier{~0u, 0u};
            if (idx < total) v[u] = a.carriers[uint64_t(r0 + lo) * a.n_samples + (idx - s_pre[lo])];
        }
#pragma unroll
        for (uint32_t u = 0; u < G; ++u) {
            const uint32_t h = 2u * v[u].sample - h0;
            if (v[u].sample != ~0u && h < hn) {                                 // (h0 is even: both haplotypes of a sample share a range)
                const uint32_t c0 = entry_count(v[u].entry, 0, a.ovf), c1 = entry_count(v[u].entry, 1, a.ovf);
                if (c0) atomicAdd(&hist[h], c0);
                if (c1) atomicAdd(&hist[h + 1u], c1);
            }
        }
    }
    __syncthreads();
    uint32_t* out = a.cnt + uint64_t(rbk) * n_haps + h0;
    uint32_t mine = 0u;
    for (uint32_t i = tid; i < hn; i += 256u) { const uint32_t v = hist[i]; out[i] = v; mine += v; }
    // the block's ids in this range: the staged emit kernel takes it when they fit its LDS stage
    __shared__ uint32_t s_tot;
    if (tid == 0) s_tot = 0u;
    __syncthreads();
    const uint32_t wsum = dec_wave_incl_scan(mine);
    if (lane == 63u) atomicAdd(&s_tot, wsum);
    __syncthreads();
    if (tid == 0) a.blk_total[blockIdx.x] = s_tot;                              // (index = record block * n_ranges + range)
}

// ---------------------------------------------------------------------------------------------------------- scan
// Exclusive prefix down the record blocks per haplotype, in DEC_SCAN_GROUPS independent groups of blocks so that the
// chain a thread walks stays short: (1) every (group, haplotype) sums its blocks, (2) every (group, haplotype) adds the
// groups above and rewrites its blocks as prefixes; the last group leaves the haplotype total in hap_begin[h + 1].
__global__ __launch_bounds__(256) void scan_groups_kernel(DecodeArgs a, uint32_t n_rowblocks, uint32_t per_group, uint32_t hap_blocks)
{
    if (a.status[0] != ~0ull) return;                       // after a failed parse `cnt` was never written: nothing to scan, and nothing that may
                                                            // overwrite the error the parse reported
    const uint32_t g = blockIdx.x / hap_blocks, h = (blockIdx.x % hap_blocks) * 256u + threadIdx.x;
    const uint32_t n_haps = 2u * a.n_samples;
    if (h >= n_haps) return;
    const uint32_t b0 = g * per_group, b1 = min(b0 + per_group, n_rowblocks);
    const uint32_t* c = a.cnt + uint64_t(b0) * n_haps + h;
    uint32_t sum = 0;
#pragma unroll 8
    for (uint32_t b = b0; b < b1; ++b, c += n_haps) sum += *c;
    a.group_tot[uint64_t(g) * n_haps + h] = sum;
}

__global__ __launch_bounds__(256) void scan_blocks_kernel(DecodeArgs a, uint32_t n_rowblocks, uint32_t per_group, uint32_t hap_blocks, uint32_t n_groups)
{
    if (a.status[0] != ~0ull) return;
    const uint32_t g = blockIdx.x / hap_blocks, h = (blockIdx.x % hap_blocks) * 256u + threadIdx.x;
    const uint32_t n_haps = 2u * a.n_samples;
    if (h >= n_haps) return;
    uint64_t run = 0;
    for (uint32_t k = 0; k < g; ++k) run += a.group_tot[uint64_t(k) * n_haps + h];
    const uint32_t b0 = g * per_group, b1 = min(b0 + per_group, n_rowblocks);
    uint32_t* c = a.cnt + uint64_t(b0) * n_haps + h;
#pragma unroll 8
    for (uint32_t b = b0; b < b1; ++b, c += n_haps) {
        const uint32_t v = *c;
        *c = uint32_t(run);
        run += v;
    }
    if (g + 1u == n_groups) {
        a.hap_begin[h + 1u] = run;
        if (run > 0xFFFFFFFFull) dec_report(a.status, uint64_t(h >> 1), DEC_CAPACITY);   // a haplotype list beyond 2^32 ids
    }
}

// one workgroup: exclusive prefix over the haplotype totals (hap_begin[1..] holds the totals on entry)
__global__ __launch_bounds__(1024) void scan_haps_kernel(DecodeArgs a)
{
    if (a.status[0] != ~0ull) {                             // (the host still wants the overflow words the parse asked for: it may retry)
        if (threadIdx.x == 0) a.status[1] = *a.ovf_used;
        return;
    }
    __shared__ uint64_t part[16];
    __shared__ uint64_t carry;
    const uint32_t n_haps = 2u * a.n_samples, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) { carry = 0; a.hap_begin[0] = 0; }
    __syncthreads();
    for (uint32_t b = 0; b < n_haps; b += 1024u) {
        const uint32_t h = b + tid;
        const uint64_t v = h < n_haps ? a.hap_begin[h + 1u] : 0ull;
        unsigned long long x = v;                                 // wave inclusive scan, 64-bit, via shuffles
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const unsigned long long y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63u) part[wave] = x;
        __syncthreads();
        uint64_t wb = carry;
        for (uint32_t w = 0; w < wave; ++w) wb += part[w];
        if (h < n_haps) a.hap_begin[h + 1u] = wb + x;
        __syncthreads();
        if (tid == 1023u) carry = wb + x;
        __syncthreads();
    }
    if (tid == 0) {
        a.status[1] = *a.ovf_used;
        if (carry > a.ids_capacity) dec_report(a.status, 0, DEC_CAPACITY);
    }
}

// ---------------------------------------------------------------------------------------------------------- emit
// One workgroup per (64-record block, haplotype range), a write cursor per haplotype in LDS.  The block's records are
// taken in order, one step each: inside a record every sample appears once, so thread k places carrier k (cursor
// read-modify-write, then the id stores) with no conflict, and an LDS-only barrier separates the records.  The first 256
// carriers of the next records are fetched a group ahead.  CUR = uint32_t while all ids of the call fit 2^32 positions
// (half the LDS), uint64_t otherwise; both are launched and the one that does not apply returns at once.
// The kernel's time is the scattered 4-byte id stores (16.7 M of them in the bench: 0.27 ms with stores, 0.08 ms without,
// wherever they point); neither longer record blocks nor haplotype-range phases change what a store costs.
template <typename CUR>
__global__ __launch_bounds__(256) void emit_kernel(DecodeArgs a, uint32_t n_ranges)
{
    extern __shared__ __align__(16) uint8_t emit_lds[];
    CUR* const cursor = reinterpret_cast<CUR*>(emit_lds);
    typedef CUR cur2 __attribute__((ext_vector_type(2)));
    constexpr uint32_t G = DEC_EMIT_GROUP;                                      // records per group
    __shared__ uint32_t s_n[DEC_ROWBLOCK], s_id0[DEC_ROWBLOCK];
    if (a.status[0] != ~0ull) return;
    // Workgroup b runs on XCD b % 8: every XCD takes a contiguous eighth of the record blocks, so the partial lines that
    // neighbouring blocks write into a haplotype's list meet in one L2 (measured: 0.27 ms against 0.31 ms in launch order).
    const uint32_t n_rowblocks = (a.n_rows + DEC_ROWBLOCK - 1u) / DEC_ROWBLOCK, per_xcd = (n_rowblocks + 7u) / 8u;
    const bool force64 = (n_ranges >> 31) != 0u;                                // (tests: the 64-bit cursors for a small call)
    n_ranges &= 0x7FFFFFFFu;
    const uint32_t unit = blockIdx.x / n_ranges, rg = blockIdx.x % n_ranges;
    const uint32_t rbk = (unit & 7u) * per_xcd + (unit >> 3);
    if (rbk >= n_rowblocks) return;
    const uint32_t n_haps = 2u * a.n_samples, h0 = rg * DEC_RANGE_HAPS, hn = min(DEC_RANGE_HAPS, n_haps - h0);
    const uint64_t total = a.hap_begin[n_haps];
    if (total > a.ids_capacity || ((total >> 32) != 0ull || force64) != (sizeof(CUR) == 8)) return;
    if (!force64 && a.blk_total[rbk * n_ranges + rg] <= DEC_STAGE_IDS &&
        a.csq_begin[min(rbk * DEC_ROWBLOCK + DEC_ROWBLOCK, a.n_rows)] - a.csq_begin[rbk * DEC_ROWBLOCK] <= 0xFFFFu) return;   // emit_staged_kernel takes this block
    const uint32_t tid = threadIdx.x;
    const uint32_t r0 = rbk * DEC_ROWBLOCK, r1 = min(r0 + DEC_ROWBLOCK, a.n_rows);
    {
        for (uint32_t i = tid; i < DEC_ROWBLOCK; i += 256u) {
            s_n[i] = r0 + i < r1 ? a.row_nnz[r0 + i] : 0u;
            s_id0[i] = r0 + i < r1 ? a.csq_begin[r0 + i] : 0u;
        }
        constexpr uint32_t U = 8u;                                              // independent loads in flight per thread
        const uint32_t* c = a.cnt + uint64_t(rbk) * n_haps + h0;
        const uint64_t* hb = a.hap_begin + h0;
        for (uint32_t i0 = tid; i0 < hn; i0 += 256u * U) {
            uint32_t cv[U];
            uint64_t hv[U];
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t i = i0 + 256u * u;
                cv[u] = i < hn ? c[i] : 0u;
                hv[u] = i < hn ? hb[i] : 0ull;
            }
#pragma unroll
            for (uint32_t u = 0; u < U; ++u) {
                const uint32_t i = i0 + 256u * u;
                if (i < hn) cursor[i] = CUR(hv[u] + cv[u]);
            }
        }
    }
    __syncthreads();

    auto place = [&](const DecCarrier v, bool valid, uint32_t id0) {
        const uint32_t h = 2u * v.sample - h0;                                  // (h0 is even: a sample's two cursors are one aligned pair)
        if (!valid || h >= hn) return;
        const uint32_t c0 = entry_count(v.entry, 0, a.ovf), c1 = entry_count(v.entry, 1, a.ovf);
        cur2* cp = reinterpret_cast<cur2*>(cursor + h);
        const cur2 at = *cp;
        cur2 nx = at;
        nx.x += c0; nx.y += c1;
        *cp = nx;
#pragma unroll
        for (uint32_t hb = 0; hb < 2u; ++hb) {
            uint32_t* o = a.ids + (hb ? at.y : at.x);
            if (!(v.entry & DEC_MULTI)) {
                uint32_t bits = (v.entry >> hb) & 0x55555555u;
                while (bits) { const uint32_t b = uint32_t(__builtin_ctz(bits)); bits &= bits - 1u; *o++ = id0 + (b >> 1); }
            } else {
                const uint32_t off = v.entry & ~DEC_MULTI, nw = a.ovf[off];
                for (uint32_t w = 0; w < nw; ++w) {
                    uint32_t bits = (a.ovf[off + 1u + w] >> hb) & 0x55555555u;
                    while (bits) { const uint32_t b = uint32_t(__builtin_ctz(bits)); bits &= bits - 1u; *o++ = id0 + 15u * w + (b >> 1); }
                }
            }
        }
    };

    const DecCarrier none{0u, 0u};
    DecCarrier cur[G], nxt[G];
    auto fetch = [&](uint32_t g0, DecCarrier (&x)[G]) {
#pragma unroll
        for (uint32_t i = 0; i < G; ++i)                                        // (slots past row_nnz hold anything: masked when placed)
            x[i] = g0 + i < r1 && tid < a.n_samples ? a.carriers[uint64_t(g0 + i) * a.n_samples + tid] : none;
    };
    fetch(r0, cur);
    for (uint32_t g0 = r0; g0 < r1; g0 += G) {
        fetch(g0 + G, nxt);                                                     // (past the block's end: nothing is loaded)
#pragma unroll
        for (uint32_t i = 0; i < G; ++i) {
            if (g0 + i >= r1) break;
            const uint32_t ni = uint32_t(__builtin_amdgcn_readfirstlane(int(s_n[g0 + i - r0])));
            const uint32_t id0 = uint32_t(__builtin_amdgcn_readfirstlane(int(s_id0[g0 + i - r0])));
            place(cur[i], tid < ni, id0);
            if (ni > 256u) {                                                    // dense records: the rest on demand
                const DecCarrier* e = a.carriers + uint64_t(g0 + i) * a.n_samples;
                for (uint32_t k0 = 256u; k0 < ni; k0 += 256u) place(k0 + tid < ni ? e[k0 + tid] : none, k0 + tid < ni, id0);
            }
            dec_lds_barrier();
        }
#pragma unroll
        for (uint32_t i = 0; i < G; ++i) cur[i] = nxt[i];
    }
}

// The same walk with the block's ids STAGED in LDS, haplotype after haplotype, and written out at the end with lane = haplotype:
// a haplotype's two or three ids of this block then leave within a few instructions of each other and the L2 merges them into one
// write, where the direct kernel above writes each id when its record comes by (tens of microseconds apart: 0.67 GB of HBM
// writes for 67 MB of ids).  Local cursors are 16-bit positions in the stage (a block here has at most DEC_STAGE_IDS ids; blocks
// with more, blocks whose consequence ids span more than 2^16 -- the stage holds 16-bit offsets from the block's first -- and calls
// that force the 64-bit cursors keep the direct kernel).  34 KB of LDS: four workgroups per CU.
__global__ __launch_bounds__(256) void emit_staged_kernel(DecodeArgs a, uint32_t n_ranges)
{
    extern __shared__ __align__(16) uint8_t emit_lds[];
    constexpr uint32_t G = DEC_EMIT_GROUP;
    __shared__ uint32_t s_n[DEC_ROWBLOCK], s_id0[DEC_ROWBLOCK], s_ws[4];
    if (a.status[0] != ~0ull) return;
    const uint32_t n_rowblocks = (a.n_rows + DEC_ROWBLOCK - 1u) / DEC_ROWBLOCK, per_xcd = (n_rowblocks + 7u) / 8u;
    const uint32_t unit = blockIdx.x / n_ranges, rg = blockIdx.x % n_ranges;
    const uint32_t rbk = (unit & 7u) * per_xcd + (unit >> 3);
    if (rbk >= n_rowblocks) return;
    const uint32_t n_haps = 2u * a.n_samples, h0 = rg * DEC_RANGE_HAPS, hn = min(DEC_RANGE_HAPS, n_haps - h0);
    if (a.hap_begin[n_haps] > a.ids_capacity) return;
    const uint32_t bt = a.blk_total[rbk * n_ranges + rg];
    if (bt == 0u || bt > DEC_STAGE_IDS) return;
    const uint32_t r0 = rbk * DEC_ROWBLOCK, r1 = min(r0 + DEC_ROWBLOCK, a.n_rows);
    const uint32_t id_base = a.csq_begin[r0];
    if (a.csq_begin[r1] - id_base > 0xFFFFu) return;                           // the stage keeps ids as 16-bit offsets from the block's first consequence
    uint16_t* const lcur = reinterpret_cast<uint16_t*>(emit_lds);              // [hn + 1] local cursors
    uint16_t* const st = reinterpret_cast<uint16_t*>(emit_lds + ((2u * (hn + 2u) + 15u) & ~15u));   // [bt] the block's ids, haplotype-major
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t* const c = a.cnt + uint64_t(rbk) * n_haps + h0;
    const bool last_block = rbk + 1u == n_rowblocks;
    for (uint32_t i = tid; i < DEC_ROWBLOCK; i += 256u) {
        s_n[i] = r0 + i < r1 ? a.row_nnz[r0 + i] : 0u;
        s_id0[i] = r0 + i < r1 ? a.csq_begin[r0 + i] - id_base : 0u;
    }
    // the block's count per haplotype: the next block's prefix minus this one's (the last block: the haplotype's total)
    for (uint32_t i = tid; i < hn; i += 256u) {
        const uint32_t nxt = last_block ? uint32_t(a.hap_begin[h0 + i + 1u] - a.hap_begin[h0 + i]) : c[n_haps + i];
        lcur[i] = uint16_t(nxt - c[i]);
    }
    __syncthreads();
    {   // exclusive prefix over the haplotypes: a segment per thread, a DPP scan of the segment sums
        const uint32_t seg = (hn + 255u) / 256u, b = min(tid * seg, hn), e = min(b + seg, hn);
        uint32_t sum = 0u;
        for (uint32_t i = b; i < e; ++i) sum += lcur[i];
        const uint32_t incl = dec_wave_incl_scan(sum);
        if (lane == 63u) s_ws[wid] = incl;
        __syncthreads();
        uint32_t run = incl - sum;
        for (uint32_t w = 0; w < wid; ++w) run += s_ws[w];
        for (uint32_t i = b; i < e; ++i) { const uint32_t v = lcur[i]; lcur[i] = uint16_t(run); run += v; }
    }
    __syncthreads();

    struct Placed { uint32_t sample, entry; };
    auto place = [&](const Placed v, bool valid, uint32_t id0) {
        const uint32_t h = 2u * v.sample - h0;                                  // (h0 is even: a sample's two cursors are one aligned 32-bit pair)
        if (!valid || h >= hn) return;
        const uint32_t c0 = entry_count(v.entry, 0, a.ovf), c1 = entry_count(v.entry, 1, a.ovf);
        uint32_t* cp = reinterpret_cast<uint32_t*>(lcur + h);
        const uint32_t at = *cp;
        *cp = at + c0 + (c1 << 16);                                             // (no carry between the halves: positions stay below 2^16)
#pragma unroll
        for (uint32_t hb = 0; hb < 2u; ++hb) {
            uint16_t* o = st + (hb ? at >> 16 : at & 0xFFFFu);
            if (!(v.entry & DEC_MULTI)) {
                uint32_t bits = (v.entry >> hb) & 0x55555555u;
                while (bits) { const uint32_t b = uint32_t(__builtin_ctz(bits)); bits &= bits - 1u; *o++ = uint16_t(id0 + (b >> 1)); }
            } else {
                const uint32_t off = v.entry & ~DEC_MULTI, nw = a.ovf[off];
                for (uint32_t w = 0; w < nw; ++w) {
                    uint32_t bits = (a.ovf[off + 1u + w] >> hb) & 0x55555555u;
                    while (bits) { const uint32_t b = uint32_t(__builtin_ctz(bits)); bits &= bits - 1u; *o++ = uint16_t(id0 + 15u * w + (b >> 1)); }
                }
            }
        }
    };
    const Placed none{0u, 0u};
    Placed cur[G], nxt[G];
    auto fetch = [&](uint32_t g0, Placed (&x)[G]) {
#pragma unroll
        for (uint32_t i = 0; i < G; ++i) {
            x[i] = none;
            if (g0 + i < r1 && tid < a.n_samples) { const DecCarrier w = a.carriers[uint64_t(g0 + i) * a.n_samples + tid]; x[i] = Placed{w.sample, w.entry}; }
        }
    };
    fetch(r0, cur);
    for (uint32_t g0 = r0; g0 < r1; g0 += G) {
        fetch(g0 + G, nxt);
#pragma unroll
        for (uint32_t i = 0; i < G; ++i) {
            if (g0 + i >= r1) break;
            const uint32_t ni = uint32_t(__builtin_amdgcn_readfirstlane(int(s_n[g0 + i - r0])));
            const uint32_t id0 = uint32_t(__builtin_amdgcn_readfirstlane(int(s_id0[g0 + i - r0])));
            place(cur[i], tid < ni, id0);
            if (ni > 256u) {
                const DecCarrier* e = a.carriers + uint64_t(g0 + i) * a.n_samples;
                for (uint32_t k0 = 256u; k0 < ni; k0 += 256u) {
                    Placed w = none;
                    if (k0 + tid < ni) { const DecCarrier x = e[k0 + tid]; w = Placed{x.sample, x.entry}; }
                    place(w, k0 + tid < ni, id0);
                }
            }
            dec_lds_barrier();
        }
#pragma unroll
        for (uint32_t i = 0; i < G; ++i) cur[i] = nxt[i];
    }
    // flush: lane = haplotype; its run is st[end of the previous haplotype's run, its own end)
    for (uint32_t h = tid; h < hn; h += 256u) {
        const uint32_t b = h ? lcur[h - 1u] : 0u, e = lcur[h];
        if (b == e) continue;
        uint32_t* o = a.ids + (a.hap_begin[h0 + h] + c[h]);
        for (uint32_t k = b; k < e; ++k) *o++ = id_base + st[k];
    }
}

}  // namespace

DecodeLayout decode_layout(uint64_t n_rows, uint64_t n_samples, uint64_t ovf_words)
{
    auto up = [](uint64_t x) { return (x + 255ull) & ~255ull; };
    DecodeLayout L{};
    L.n_rowblocks = uint32_t((n_rows + DEC_ROWBLOCK - 1) / DEC_ROWBLOCK);
    uint64_t o = 0;
    L.carriers_off = o; o += up(n_rows * n_samples * sizeof(DecCarrier));
    L.nnz_off = o; o += up(n_rows * 4ull);
    L.cnt_off = o; o += up(uint64_t(L.n_rowblocks) * 2ull * n_samples * 4ull);
    L.group_off = o; o += up(uint64_t(DEC_SCAN_GROUPS) * 2ull * n_samples * 4ull);
    L.blk_off = o; o += up(uint64_t(L.n_rowblocks) * ((2ull * n_samples + DEC_RANGE_HAPS - 1) / DEC_RANGE_HAPS) * 4ull);
    L.ovf_off = o; o += up(ovf_words * 4ull);
    L.ovf_used_off = o; o += 256;
    L.total = o;
    return L;
}

hipError_t launch_decode(const DecodeArgs& a, hipStream_t stream, unsigned phases)
{
    if (a.n_rows == 0 || a.n_samples == 0) return hipErrorInvalidValue;
    const uint32_t n_rowblocks = (a.n_rows + DEC_ROWBLOCK - 1u) / DEC_ROWBLOCK;
    const uint32_t n_haps = 2u * a.n_samples;
    if (phases & 1u) {
        hipError_t e = hipMemsetAsync(a.ovf_used, 0, sizeof(unsigned long long), stream);
        if (e != hipSuccess) return e;
        // a record of S samples is about 6..30 S bytes of text; a tile is 16 bytes per thread
        const uint32_t bs = a.parse_threads ? a.parse_threads : (a.n_samples <= 96u ? 64u : (a.n_samples <= 320u ? 128u : 256u));
        if (bs <= 64u) hipLaunchKernelGGL(parse_rows_kernel<64>, dim3(a.n_rows), dim3(64), 0, stream, a);
        else if (bs <= 128u) hipLaunchKernelGGL(parse_rows_kernel<128>, dim3(a.n_rows), dim3(128), 0, stream, a);
        else hipLaunchKernelGGL(parse_rows_kernel<256>, dim3(a.n_rows), dim3(256), 0, stream, a);
    }
    const uint32_t n_ranges = (n_haps + DEC_RANGE_HAPS - 1u) / DEC_RANGE_HAPS, range_haps = min(n_haps, DEC_RANGE_HAPS);
    if (phases & 2u) hipLaunchKernelGGL(count_kernel, dim3(n_rowblocks * n_ranges), dim3(256), range_haps * 4u, stream, a, n_ranges);
    if (phases & 4u) {
        const uint32_t n_groups = min(DEC_SCAN_GROUPS, n_rowblocks), per_group = (n_rowblocks + n_groups - 1u) / n_groups;
        const uint32_t groups = (n_rowblocks + per_group - 1u) / per_group, hbk = (n_haps + 255u) / 256u;
        hipLaunchKernelGGL(scan_groups_kernel, dim3(groups * hbk), dim3(256), 0, stream, a, n_rowblocks, per_group, hbk);
        hipLaunchKernelGGL(scan_blocks_kernel, dim3(groups * hbk), dim3(256), 0, stream, a, n_rowblocks, per_group, hbk, groups);
        hipLaunchKernelGGL(scan_haps_kernel, dim3(1), dim3(1024), 0, stream, a);
    }
    if (phases & 8u) {
        const uint32_t units = 8u * ((n_rowblocks + 7u) / 8u);
        const bool force64 = getenv("V2P_DECODE_CURSOR64") != nullptr;            // test hook: exercise the 64-bit cursor kernel on small inputs
        const uint32_t nr = n_ranges | (force64 ? 0x80000000u : 0u);
        if (!force64) hipLaunchKernelGGL(emit_staged_kernel, dim3(units * n_ranges), dim3(256), ((2u * (range_haps + 2u) + 15u) & ~15u) + DEC_STAGE_IDS * 2u,
                                         stream, a, n_ranges);
        hipLaunchKernelGGL(emit_kernel<uint32_t>, dim3(units * n_ranges), dim3(256), range_haps * 4u, stream, a, nr);
        hipLaunchKernelGGL(emit_kernel<uint64_t>, dim3(units * n_ranges), dim3(256), range_haps * 8u, stream, a, nr);
    }
    return hipGetLastError();
}

}  // namespace v2p
