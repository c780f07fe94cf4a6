// build_rows.hip -- the device image of a batch built in ONE pass over the transcript stream (ROWS images: rows_image.hpp).
//
// Replaces, for wave and dense images, round 3's build_kernels.hip (one lane per transcript walking its tasks twice -- a counting
// pass and an emitting pass, 12.3 ms for the north star's cohort whose execute takes 7.6).  Here
//   tile_bytes   arena bytes per tile of K transcripts, scanned: res_counter of haplotype_instruction.rs:90,132 per tile
//   parse        lane = ITEM of the stream (a Task, task.rs:2-9, or the HEAD of a transcript).  Every stream array is read once,
//                coalesced; update_task's checks (haplotype_instruction.rs:140-158) and Task::execute's bounds (task.rs:43,47) are
//                what the device reports instead of panicking; the packer's fusion state machine is solved for 64 items at once on
//                ballot masks (rows_parse); descriptors -- whole, nothing is cut -- are staged in LDS and leave coalesced once a
//                decoupled look-back over the tiles' descriptor counts has told the wave where they go; every 1 KiB row of the
//                arena learns which descriptor covers its first byte
//   cut          one wave per segment of 640 rows walks the row map greedily: as many rows as one wave takes (ten) while the
//                descriptors fit its lanes; counted, scanned, emitted
//   keys         proteome slice and window of every chunk for the XCD / window order (build_kernels.hip: launch_order_blocks)
// Integer / index work only (no MFMA); the parse is bound by vector-instruction issue and by the stream's bytes, see DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "build_rows.h"
#include "build_kernels.h"
#include "stitch_device.hpp"

namespace v2p {

namespace {

__device__ __forceinline__ void rreport(unsigned long long* status, uint64_t index, uint32_t reason) { atomicMin(status, (unsigned long long)((index << 8) | reason)); }
__device__ __forceinline__ uint32_t mbcnt(uint64_t m) { return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u)); }
// lane i receives lane i - 1's value, lane 0 `first` (DPP wave_shr:1)
__device__ __forceinline__ uint32_t up1(uint32_t x, uint32_t first) { return uint32_t(__builtin_amdgcn_update_dpp(int(first), int(x), 0x138, 0xf, 0xf, false)); }
__device__ __forceinline__ uint32_t up2(uint32_t x) { return up1(up1(x, 0u), 0u); }
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ uint64_t wave_incl_scan64(uint64_t v, uint32_t lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint64_t y = __shfl_up(v, o); if (lane >= uint32_t(o)) v += y; }
    return v;
}

// ---- arena bytes per tile ----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_tile_bytes_kernel(RowsArgs a)
{
    const uint64_t t = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    uint64_t len = 0;
    if (t < a.n_tx) {
        const uint32_t hl = a.tx_header_len ? a.tx_header_len[t] : 0u;
        len = uint64_t(a.tx_res_len[t]) + (hl ? hl + 1u : 0u);
    }
    // K consecutive lanes are one tile (K a power of two <= 64, tiles aligned with the wave)
    for (uint32_t o = 1; o < a.K; o <<= 1) len += __shfl_xor(len, int(o));
    if ((t & (a.K - 1u)) == 0u && (t >> a.log2K) < a.n_tiles) a.tile_bytes[t >> a.log2K] = len;
}

// exclusive scan of u64 values, three passes over 1024-element tiles (as build_kernels.hip's launch_scan_u32)
constexpr uint32_t RS_TILE = 1024;
__global__ __launch_bounds__(256) void rows_scan_sums(const uint64_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ tile_sum)
{
    __shared__ uint64_t s[4];
    const uint64_t base = uint64_t(blockIdx.x) * RS_TILE;
    uint64_t v = 0;
    for (uint32_t k = 0; k < 4; ++k) { const uint64_t i = base + threadIdx.x * 4u + k; if (i < n) v += in[i]; }
    v = wave_sum64(v);
    if ((threadIdx.x & 63u) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(1024) void rows_scan_tiles(uint64_t* __restrict__ tile_sum, uint64_t n_tiles)
{
    __shared__ uint64_t s_w[16];
    __shared__ uint64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t b = 0; b < n_tiles; b += 1024) {
        const uint64_t i = b + threadIdx.x;
        const uint64_t x = i < n_tiles ? tile_sum[i] : 0;
        const uint64_t v = wave_incl_scan64(x, threadIdx.x & 63u);
        if ((threadIdx.x & 63u) == 63u) s_w[threadIdx.x >> 6] = v;
        __syncthreads();
        uint64_t before = s_carry;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += s_w[w];
        if (i < n_tiles) tile_sum[i] = before + v - x;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_sum[n_tiles] = s_carry;
}
__global__ __launch_bounds__(256) void rows_scan_apply(const uint64_t* __restrict__ in, uint64_t n, const uint64_t* __restrict__ tile_sum, uint64_t* __restrict__ out)
{
    __shared__ uint64_t s[4];
    const uint64_t base = uint64_t(blockIdx.x) * RS_TILE;
    uint64_t x[4], v = 0;
    for (uint32_t k = 0; k < 4; ++k) { const uint64_t i = base + threadIdx.x * 4u + k; x[k] = i < n ? in[i] : 0ull; v += x[k]; }
    const uint64_t incl = wave_incl_scan64(v, threadIdx.x & 63u);
    if ((threadIdx.x & 63u) == 63u) s[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t before = tile_sum[blockIdx.x];
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += s[w];
    uint64_t run = before + incl - v;
    for (uint32_t k = 0; k < 4; ++k) { const uint64_t i = base + threadIdx.x * 4u + k; if (i < n) out[i] = run; run += x[k]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) out[n] = tile_sum[gridDim.x];
}

// ---- the parse ------------------------------------------------------------------------------------------------------------------
// What bounds it (measured, profiles/r04_build_*): vector-instruction issue.  A window of 64 items costs a few hundred VALU
// instructions however its loads are arranged (a burst of the tile's Task arrays into LDS with global_load_lds changed nothing; a
// decoupled look-back for the descriptors' final place cost a third of the kernel: persistent waves run in lockstep, so every
// generation of tiles waits for the prefix to ripple through it).  So: one tile per 64-lane workgroup, plain grid; descriptors go
// to a PADDED array -- ROWS_PAD slots per tile -- and a copy kernel compacts them once a scan of the tiles' counts has told every
// tile where it starts; positions are 32-bit offsets from the tile's first emitted byte; per-transcript values sit in LDS as
// arrays of words; everything rare (runs of more than 1 KiB, which may cross two rows, or of more than a descriptor's length
// field) is behind one wave-uniform branch.
constexpr uint32_t ROWS_PAD = 256;             // descriptor slots per tile in the padded array (a tile with more: the two-pass form)
enum : int { PH_PAD = 0, PH_COUNT = 1, PH_DIRECT = 2 };

struct __attribute__((aligned(16))) WaveLds {
    uint64_t base[2][66];                      // [0]: proteome offset of the slot's transcript, [1]: its alt tape's offset (slot j = transcript t0 - 1 + j)
    uint32_t bound[2][66];                     // [0]: its reference length, [1]: its alt tape's length
    uint32_t res_len[66];
    uint32_t pos[66];                          // arena offset of its first result byte (behind its FASTA header) minus the tile's first emitted byte
    uint32_t hl[66];                           // FASTA: length of its record header (0: none)
    uint64_t hsrc[66];                         // ... and where the header sits in the resident reference
    uint32_t flag[16];                         // one byte per lane: the item is a HEAD
};

// PHASE: PH_PAD descriptors to the tile's slots of the padded array and its count to tile_count (a tile that does not fit is
// reported); PH_COUNT only the count; PH_DIRECT descriptors to their final place (tile_desc_base: the scan of the counts).
template <int MODE, bool FASTA, int PHASE>
__global__ __launch_bounds__(64) void rows_parse_kernel(RowsArgs a)
{
    constexpr uint32_t CTX = MODE == ROWS_DENSE ? 4u : 2u, ADV = 62u - CTX;
    __shared__ WaveLds L;
    const uint32_t lane = threadIdx.x;
    const uint64_t tile = blockIdx.x;
    const uint64_t n_heads = a.n_tx + 1u;
    const uint64_t t0 = tile << a.log2K;
    const uint32_t nh = uint32_t(n_heads - t0 < a.K ? n_heads - t0 : a.K);
    const uint64_t t1 = t0 + nh;
    const uint64_t task_lo = a.tx_task_begin[t0], task_end = t1 <= a.n_tx ? a.tx_task_begin[t1] : a.n_tasks;
    const uint64_t tile_base = a.tile_res_base[tile];
    // end of the last task before the tile: HEAD(t0) fills the rest of the transcript before it with '.'
    uint32_t carry_e = 0, prev_res_len = 0, prev_hl = 0;
    uint64_t prev_hsrc = 0;
    if (t0 > 0) {
        prev_res_len = a.tx_res_len[t0 - 1u];
        if (FASTA) { prev_hl = a.tx_header_len[t0 - 1u]; prev_hsrc = prev_hl ? a.proteome_len + a.tx_header_off[t0 - 1u] : 0ull; }
        if (task_lo > a.tx_task_begin[t0 - 1u]) { const uint64_t i = task_lo - 1u; carry_e = a.start_pos_res[i] + a.length[i]; }
    }
    // the tile's descriptors start where the transcript BEFORE it left off (HEAD(t0) writes that one's '.' fill and line feed): positions
    // are 32-bit offsets from there (a tile of more than 2 GiB of result takes the two-pass form with K = 1 ... refused below)
    const uint32_t prev_tail = (prev_res_len > carry_e ? prev_res_len - carry_e : 0u) + (prev_hl ? 1u : 0u);
    const uint64_t ebase = tile_base - prev_tail;
    const uint64_t tile_end = a.tile_res_base[tile + 1u];
    if (tile_end - ebase > 0x7FFFFFFFull) { if (lane == 0) rreport(a.status, task_lo, STATUS_ROWS_SPAN); if (PHASE != PH_DIRECT && lane == 0) a.tile_count[tile] = 0u; return; }
    const uint32_t eoff = uint32_t(ebase) & (ROW_BYTES - 1u);        // the tile's first emitted byte inside its row
    const uint64_t erow = ebase / ROW_BYTES;
    // ---- the tile's transcripts: slot lane + 1 = transcript t0 + lane, slot 0 = the one before the tile ----
    uint32_t hp = 0xFFFFFFFFu;                                       // item (relative to the tile's first) of the lane's HEAD
    {
        const uint64_t u = t0 + lane;
        const bool valid = lane < nh && u < a.n_tx;
        uint64_t poff = 0, alt0 = 0, hsrc = 0;
        uint32_t ref_len = 0, res_len = 0, n_alt = 0, hl = 0, alen = 0;
        if (lane < nh) hp = uint32_t(a.tx_task_begin[u] - task_lo) + lane;
        if (valid) {
            poff = a.tx_proteome_off[u]; alt0 = a.tx_alt_begin[u]; n_alt = uint32_t(a.tx_alt_begin[u + 1] - alt0);
            ref_len = a.tx_ref_len[u]; res_len = a.tx_res_len[u];
            if (FASTA) { hl = a.tx_header_len[u]; hsrc = hl ? a.proteome_len + a.tx_header_off[u] : 0ull; }
            alen = res_len + (hl ? hl + 1u : 0u);                    // (< 2^31 in total: checked above)
            if (PHASE != PH_DIRECT && poff + ref_len > a.proteome_len) rreport(a.status, a.tx_task_begin[u], STATUS_SRC_OOB);   // transcript outside the resident proteome
        }
        const uint32_t rb = prev_tail + wave_incl_scan(alen) - alen;  // the transcript's first arena byte, from ebase
        if (lane < nh) {
            L.base[0][lane + 1u] = poff; L.base[1][lane + 1u] = alt0; L.bound[0][lane + 1u] = ref_len; L.bound[1][lane + 1u] = n_alt;
            L.res_len[lane + 1u] = res_len; L.pos[lane + 1u] = rb + hl;
            if (FASTA) { L.hl[lane + 1u] = hl; L.hsrc[lane + 1u] = hsrc; }
        }
        if (lane == 0) {
            L.res_len[0] = prev_res_len; L.pos[0] = 0u - carry_e;    // (slot 0's tasks ended at carry_e: its '.' fill starts at offset 0)
            if (FASTA) { L.hl[0] = prev_hl; L.hsrc[0] = prev_hsrc; }
        }
    }
    asm volatile("" ::: "memory");
    const uint32_t n_items = uint32_t(task_end - task_lo) + nh;
    const uint8_t* const g_code = a.code + task_lo;
    const uint32_t* const g_sp = a.start_pos + task_lo;
    const uint32_t* const g_ln = a.length + task_lo;
    const uint32_t* const g_sr = a.start_pos_res + task_lo;
    uint64_t* const out = PHASE == PH_PAD ? a.desc_pad + tile * ROWS_PAD : (PHASE == PH_DIRECT ? a.desc + a.tile_desc_base[tile] : nullptr);
    (void)out;
    const uint32_t out_cap = PHASE == PH_PAD ? ROWS_PAD : 0xFFFFFFFFu;
    // cover entry: tile : 25 | descriptor inside the tile : 16 | offset inside it : 22 -- the two-pass form: 1 << 63 | descriptor << 22 | offset
    const uint64_t dbase = PHASE == PH_DIRECT ? a.tile_desc_base[tile] : 0ull;
    auto cover_word = [&](uint32_t dk, uint32_t off) -> uint64_t { return PHASE == PH_DIRECT ? (1ull << 63) | ((dbase + dk) << 22) | off : (tile << 38) | (uint64_t(dk) << 22) | off; };

    uint32_t tile_cnt = 0;                     // descriptors of the tile so far (wave-uniform)
    uint64_t carry_h = 0, carry_second = 0;
    bool first = true;
    for (uint32_t R0 = 0; ; R0 += ADV) {
        const uint32_t nvalid = n_items - R0 < 64u ? n_items - R0 : 64u;
        const bool last = R0 + 64u >= n_items;
        const uint32_t e_lo = first ? 0u : CTX, e_hi = last ? nvalid : 62u;
        // ---- which items are HEADs, and every item's transcript ----
        if (lane < 16u) L.flag[lane] = 0u;
        asm volatile("" ::: "memory");
        if (hp - R0 < 64u) reinterpret_cast<uint8_t*>(L.flag)[hp - R0] = 1u;
        asm volatile("" ::: "memory");
        const bool active = lane < nvalid;
        const bool isHead = reinterpret_cast<const uint8_t*>(L.flag)[lane] != 0u;      // (lanes >= nvalid: no HEAD of this tile lies there)
        const uint64_t headmask = __ballot(isHead);
        const uint32_t heads_before = uint32_t(__popcll(__ballot(hp < R0)));
        const uint32_t slot = heads_before + mbcnt(headmask) + (isHead ? 1u : 0u);       // slot of the item's transcript (a HEAD: the one it opens)
        const bool isTask = active && !isHead;
        const uint32_t ti = R0 + lane - slot;                                            // task, relative to the tile's first
        uint32_t code = 0, sp = 0, ln = 0, sr = 0;
        if (isTask) { code = g_code[ti]; sp = g_sp[ti]; ln = g_ln[ti]; sr = g_sr[ti]; }
        const uint32_t res_len = L.res_len[slot], pos0 = L.pos[slot];
        // ---- update_task / Task::execute checks; result positions ----
        const bool res_oob = isTask && (ln > res_len || sr > res_len - ln);
        const uint32_t e = isTask && !res_oob ? sr + ln : 0u;                             // end of the task inside its transcript's result
        const uint32_t pe = up1(e, carry_e);                                              // ... of the item before
        const uint32_t csel = code == 1u ? 1u : 0u;
        const uint32_t bound = L.bound[csel][slot];
        uint32_t why = 0;
        if (isTask) {
            if (code > 1u) why = STATUS_BAD_CODE;
            else if (res_oob) why = STATUS_RES_OOB;
            else if (ln > bound || sp > bound - ln) why = STATUS_SRC_OOB;
            else if (sr < pe) why = STATUS_NOT_CONTIGUOUS;
        }
        const bool in_emit = lane >= e_lo && lane < e_hi;
        if (PHASE != PH_DIRECT && __ballot(why != 0u && in_emit)) { if (why != 0u && in_emit) rreport(a.status, task_lo + ti, why); }
        const bool good = isTask && why == 0u;
        // ---- classes of the fusion state machine ----
        const bool isRef = good && code == 0u;
        const bool imm = good && code == 1u && ln - 1u < IMM_MAX_BYTES;
        const uint64_t src = L.base[csel][slot] + sp;
        uint64_t lit = 0;
        if (imm) {                                                                        // short alt payloads travel inside their descriptor
            struct __attribute__((packed, aligned(1))) U64 { uint64_t v; };
            lit = reinterpret_cast<const U64*>(a.alt + src)->v & (~0ull >> (64u - 8u * ln));
        }
        const bool ps = isRef && ln <= ROWS_FUSE_LEN;
        const bool cA = ps && src + ln <= SNV3_MAX_SRC - 1u - ROWS_FUSE_LEN;
        const bool cB = imm && ln == 1u;
        const bool c0 = ps && ln > 0u && src >= 1u && src + ln <= SNV3_MAX_SRC;
        const uint32_t src32 = uint32_t(src);
        const uint32_t src2 = up2(src32), ln2 = up2(ln);
        const bool c1 = ln2 == 0u ? c0 : (ps && (ln == 0u || src == uint64_t(src2) + ln2 + 1u));
        const uint64_t prev_head = (headmask << 1) | (first ? 1ull : 0ull);
        const bool gap = good && sr > pe;
        const uint64_t mRst = ~__ballot(good) | prev_head | __ballot(gap);
        const RowsParse p = rows_parse(__ballot(cA), __ballot(cB), __ballot(ps), __ballot(c0), __ballot(c1), mRst, !first, carry_h);
        const bool isF = (p.F >> lane) & 1ull, isReal = (p.real >> lane) & 1ull;
        // a closing lane's fused substitution: run, len1, byte, len2
        const uint32_t f_len1 = isF && isReal ? ln2 : 0u;
        const uint32_t f_byte = up1(uint32_t(lit), 0u) & 0xFFu;
        const uint32_t f_run = f_len1 == 0u ? src32 - 1u : src2;
        uint64_t second = 0;
        uint32_t p_len1 = 0, p_len2 = 0, p_run = 0, p_byte = 0;
        if (MODE == ROWS_DENSE) {
            p_len1 = up2(f_len1); p_len2 = up2(ln); p_run = up2(f_run); p_byte = up2(f_byte);
            const bool Lc = isF && ((p.F >> (lane >= 2u ? lane - 2u : 63u)) & 1ull) && lane >= 2u && !((mRst >> (lane - 1u)) & 1ull) && f_len1 == 0u &&
                            p_len1 <= SNV5_MAX_LEN && p_len2 <= SNV5_MAX_LEN && ln <= SNV5_MAX_LEN && f_run == p_run + p_len1 + 1u + p_len2;
            second = rows_pair(__ballot(Lc), !first, carry_second);
        }
        const uint64_t absorbed = (p.F >> 1) | ((p.F & p.real) >> 2) | (MODE == ROWS_DENSE ? (second >> 2) & p.F : 0ull);
        const bool isAbs = (absorbed >> lane) & 1ull, isSecond = (second >> lane) & 1ull;
        // ---- what the lane emits: up to three runs of result bytes, in order, back to back from offset q0 ----
        //   HEAD: '.' fill of the transcript it closes, [FASTA: that transcript's line feed, its own header];  task: '.' fill of a gap, itself
        uint32_t w0l = 0, w0h = 0, w1l = 0, w1h = 0, w2l = 0, w2h = 0;      // descriptor words (the length field of a plain one is filled in below)
        uint32_t l0 = 0, l1 = 0, l2 = 0, q0 = 0;
        bool fusedw = false;                                                 // w1 is a complete fused word
        if (isHead) {
            const uint32_t prl = L.res_len[slot - 1u];
            q0 = L.pos[slot - 1u] + pe;
            l0 = prl > pe ? prl - pe : 0u; w0h = SPACE_FILL << 30;
            if (FASTA) {
                const uint32_t phl = L.hl[slot - 1u], xhl = L.hl[slot];
                if (phl) { const uint64_t lf = L.hsrc[slot - 1u] + phl - 1u; l1 = 1u; w1l = uint32_t(lf); w1h = uint32_t(lf >> 32) & 0xFFu; }
                if (xhl) { const uint64_t hs = L.hsrc[slot]; l2 = xhl; w2l = uint32_t(hs); w2h = uint32_t(hs >> 32) & 0xFFu; }
            }
        } else if (good) {
            q0 = pos0 + pe;
            if (gap) { l0 = sr - pe; w0h = SPACE_FILL << 30; }
            if (isF) {
                if (!isAbs) {
                    fusedw = true;
                    if (MODE == ROWS_DENSE && isSecond) {
                        l1 = p_len1 + 1u + p_len2 + 1u + ln; q0 -= p_len1 + p_len2 + 2u;
                        const uint64_t w = SNV5_MARK | (uint64_t(f_byte) << 52) | (uint64_t(p_byte) << 44) | (uint64_t(ln & 31u) << 39) | (uint64_t(p_len2 & 31u) << 34) | (uint64_t(p_len1 & 31u) << 29) | (uint64_t(p_run) & SNV3_MAX_SRC);
                        w1l = uint32_t(w); w1h = uint32_t(w >> 32);
                    } else {
                        l1 = f_len1 + 1u + ln; q0 -= f_len1 + 1u;
                        w1l = (f_run & 0x1FFFFFFFu) | (f_len1 << 29);
                        w1h = ((f_len1 & 0xFFFu) >> 3) | ((ln & 0xFFFu) << 9) | (f_byte << 21) | (7u << 29);
                    }
                }
            } else if (!isAbs && ln != 0u) {
                l1 = ln;
                if (imm) { w1l = uint32_t(lit); w1h = uint32_t(lit >> 32) | (SPACE_IMM << 30); }
                else { w1l = src32; w1h = (uint32_t(src >> 32) & 0xFFu) | ((isRef ? SPACE_PROTEOME : SPACE_PAYLOAD) << 30); }
            }
        }
        if (!in_emit) { l0 = 0; l1 = 0; l2 = 0; }
        // Everything rare behind ONE uniform branch: a run of more than 1 KiB (it may cross two rows of the arena, or exceed a
        // descriptor's length field and become several descriptors)
        const bool slow = __ballot(l0 > ROW_BYTES || l1 > ROW_BYTES || (FASTA && l2 > ROW_BYTES)) != 0ull;
        if (!slow) {
            const uint32_t c0n = l0 ? 1u : 0u, c1n = l1 ? 1u : 0u, c2n = FASTA && l2 ? 1u : 0u;
            const uint32_t cnt = c0n + c1n + c2n;
            const uint32_t incl = wave_incl_scan(cnt);
            const uint32_t round_total = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
            if (PHASE != PH_COUNT && round_total != 0u) {
                const uint32_t k = tile_cnt + incl - cnt;                                 // the lane's first slot inside the tile
                if (!fusedw) w1h |= l1 << 8;
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                if (l0 && k < out_cap) *reinterpret_cast<u32x2*>(out + k) = u32x2{w0l, w0h | (l0 << 8)};
                if (l1 && k + c0n < out_cap) *reinterpret_cast<u32x2*>(out + k + c0n) = u32x2{w1l, w1h};
                if (FASTA && l2 && k + c0n + c1n < out_cap) *reinterpret_cast<u32x2*>(out + k + c0n + c1n) = u32x2{w2l, w2h | (l2 << 8)};
                // the rows whose first byte the lane's runs cover: its span is at most 2 (FASTA: 3) KiB + ...; one boundary per run at most
                const uint32_t s = eoff + q0, t1e = s + l0, t2e = t1e + l1, t3e = t2e + (FASTA ? l2 : 0u);
                const uint32_t rfirst = (s + ROW_BYTES - 1u) >> 10;                       // first boundary at or behind the span's start
                if ((rfirst << 10) < t3e && cnt != 0u) {
                    for (uint32_t r = rfirst; (r << 10) < t3e; ++r) {                      // (one round, rarely two)
                        const uint32_t rb = r << 10;
                        const uint32_t which = rb < t1e ? 0u : (rb < t2e ? 1u : 2u);
                        const uint32_t dk = which == 0u ? k : (which == 1u ? k + c0n : k + c0n + c1n);
                        const uint32_t off = rb - (which == 0u ? s : (which == 1u ? t1e : t2e));
                        const uint64_t row = erow + r;
                        if (row >= 1u && row < a.n_rows) a.cover[row] = cover_word(dk, off);
                    }
                }
            }
            tile_cnt += round_total;
        } else {
            // the general form: any length, any number of pieces and rows
            const uint64_t W0 = (uint64_t(w0h) << 32) | w0l, W1 = (uint64_t(w1h) << 32) | w1l, W2 = (uint64_t(w2h) << 32) | w2l;
            auto pieces = [](uint32_t l) -> uint32_t { return l == 0u ? 0u : (l + PIECE_MAX - 1u) / PIECE_MAX; };
            const uint32_t n0 = pieces(l0), n1 = fusedw ? (l1 ? 1u : 0u) : pieces(l1), n2 = pieces(l2);
            const uint32_t cnt = n0 + n1 + n2;
            const uint32_t incl = wave_incl_scan(cnt);
            const uint32_t round_total = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
            if (PHASE != PH_COUNT && round_total != 0u) {
                uint32_t k = tile_cnt + incl - cnt;
                uint32_t pos = eoff + q0;
                auto put = [&](uint64_t word, uint32_t len, bool whole) {
                    const unsigned space = unsigned(word >> 62);
                    uint64_t sfield = word & SRC_MASK;
                    while (len) {
                        const uint32_t piece = whole ? len : (len < PIECE_MAX ? len : PIECE_MAX);
                        const uint64_t dword = whole ? word : ((word & ~SRC_MASK) | (sfield & SRC_MASK) | (uint64_t(piece) << 40));
                        if (k < out_cap) out[k] = dword;
                        for (uint32_t r = (pos + ROW_BYTES - 1u) >> 10; (uint64_t(r) << 10) < uint64_t(pos) + piece; ++r) {
                            const uint64_t row = erow + r;
                            if (row >= 1u && row < a.n_rows) a.cover[row] = cover_word(k, (r << 10) - pos);
                        }
                        if (space == SPACE_PROTEOME || space == SPACE_PAYLOAD) sfield += piece;      // (an immediate is never cut: <= 5 bytes)
                        len -= piece; pos += piece; ++k;
                        if (whole) break;
                    }
                };
                if (l0) put(W0, l0, false);
                if (l1) put(W1, l1, fusedw);
                if (l2) put(W2, l2, false);
            }
            tile_cnt += round_total;
        }
        if (last) break;
        carry_h = p.h >> ADV; carry_second = second >> ADV;
        carry_e = uint32_t(__builtin_amdgcn_readlane(int(e), int(ADV - 1u)));
        first = false;
    }
    if (PHASE != PH_DIRECT && lane == 0) {
        a.tile_count[tile] = tile_cnt;
        if (PHASE == PH_PAD && (tile_cnt > ROWS_PAD || tile_cnt > 0xFFFFu)) rreport(a.status, tile, STATUS_ROWS_STAGE);
    }
}

// descriptors out of the padded array into their final, dense place (tile_desc_base: the scan of the tiles' counts): one wave per tile
__global__ __launch_bounds__(256) void rows_compact_kernel(RowsArgs a)
{
    const uint64_t tile = uint64_t(blockIdx.x) * 4u + (threadIdx.x >> 6);
    if (tile >= a.n_tiles) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t b0 = a.tile_desc_base[tile], n = a.tile_desc_base[tile + 1u] - b0;
    if (n > ROWS_PAD || b0 + n > a.desc_cap) return;                  // (reported by the parse / by the host)
    const uint64_t* src = a.desc_pad + tile * ROWS_PAD;
    for (uint32_t k = lane; k < n; k += 64u) a.desc[b0 + k] = src[k];
}

__global__ __launch_bounds__(256) void rows_hap_begin_kernel(RowsArgs a)
{
    const uint64_t h = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (h > a.n_haps) return;
    const uint64_t t = h < a.n_haps ? a.hap_tx_begin[h] : a.n_tx;
    const uint64_t tile = t >> a.log2K;
    uint64_t b = a.tile_res_base[tile < a.n_tiles ? tile : a.n_tiles];
    if (tile < a.n_tiles)
        for (uint64_t u = tile << a.log2K; u < t; ++u) { const uint32_t hl = a.tx_header_len ? a.tx_header_len[u] : 0u; b += uint64_t(a.tx_res_len[u]) + (hl ? hl + 1u : 0u); }
    a.hap_out_begin[h] = b;
}

// ---- the cutter: one wave per segment of ROWS_SEG rows ----------------------------------------------------------------------------
template <bool EMIT>
__global__ __launch_bounds__(64) void rows_cut_kernel(RowsArgs a, uint32_t max_rows, uint32_t max_desc, uint64_t flag)
{
    const uint32_t lane = threadIdx.x;
    const uint64_t n_desc = a.tile_desc_base[a.n_tiles];
    if (*a.status != STATUS_CLEAN) {                                                      // the parse failed (or asks for its two-phase form): the row map is not to be walked
        if (!EMIT && lane == 0) a.seg_count[blockIdx.x] = 0u;
        return;
    }
    const uint64_t seg = blockIdx.x;
    const uint64_t s0 = seg * ROWS_SEG, s1 = s0 + ROWS_SEG < a.n_rows ? s0 + ROWS_SEG : a.n_rows;
    uint32_t count = 0;
    uint64_t out_k = EMIT ? a.seg_base[seg] : 0;
    uint64_t last_dst = 0;
    uint64_t r0 = s0;
    while (r0 < s1) {
        // rows r0 .. r0 + 63 in registers: the descriptor covering each row's first byte
        const uint64_t b = r0, r = b + lane;
        uint64_t c = 0, idx = 0;
        if (r >= 1u && r < a.n_rows) { c = a.cover[r]; idx = (c >> 63) ? (c >> 22) & ((1ull << 41) - 1ull) : a.tile_desc_base[c >> 38] + ((c >> 22) & 0xFFFFu); }
        const uint32_t off = uint32_t(c) & 0x3FFFFFu;
        const uint64_t lastd = r >= a.n_rows ? n_desc - 1u : (off ? idx : idx - 1u);      // last descriptor of a chunk that ends at row r
        uint32_t cur = 0;
        for (;;) {
            const uint64_t f = uint64_t(uint32_t(__builtin_amdgcn_readlane(int(uint32_t(idx)), int(cur)))) | (uint64_t(uint32_t(__builtin_amdgcn_readlane(int(uint32_t(idx >> 32)), int(cur)))) << 32);
            const uint32_t hs = uint32_t(__builtin_amdgcn_readlane(int(off), int(cur)));
            const bool ok = lane > cur && lane <= cur + max_rows && r <= s1 && lastd - f + 1u <= max_desc;
            const uint64_t m = __ballot(ok);
            if (!m) { if (lane == 0) rreport(a.status, f, STATUS_ROWS_TOO_MANY); return; }
            const uint32_t hb = 63u - uint32_t(__builtin_clzll(m));
            if (EMIT && lane == hb) {
                const uint64_t n = lastd - f + 1u;
                uint32_t tc = 0;                                                          // what the chunk's last descriptor has behind the cut
                if (r < a.n_rows && off != 0u) {
                    const uint64_t d = a.desc[lastd];
                    const uint32_t dl = (d >> 60) == 0xDull ? uint32_t((d >> 29) & 31u) + uint32_t((d >> 34) & 31u) + uint32_t((d >> 39) & 31u) + 2u
                                      : ((d & SNV3_MARK) == SNV3_MARK ? uint32_t((d >> 29) & 0xFFFu) + 1u + uint32_t((d >> 41) & 0xFFFu) : uint32_t(d >> 40) & LEN_MASK);
                    tc = dl - off;
                }
                a.chunks_tmp[out_k] = Chunk{f | (uint64_t(hs) << TB_IDX_BITS) | (uint64_t(tc) << (TB_IDX_BITS + TB_SKIP_BITS)), ((b + cur) * ROW_BYTES) | (n << 48) | CHUNK_CLIP | flag};
            }
            last_dst = (b + cur) * ROW_BYTES;
            ++count; ++out_k;
            cur = hb;
            if (b + cur >= s1) break;
            if (cur + max_rows > 63u) break;                                              // the next chunk's candidates leave the registers: reload from its first row
        }
        r0 = b + cur;
    }
    if (!EMIT && lane == 0) {
        a.seg_count[seg] = count;
        if (seg + 1u == a.n_segs) a.totals[2] = last_dst;
    }
}

// proteome slice and window of every chunk (order_chunks_for_xcds: the first reference read among its first six descriptors)
__global__ __launch_bounds__(256) void rows_keys_kernel(RowsArgs a, uint64_t n_chunks, uint64_t n_desc)
{
    const uint64_t k = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (k >= n_chunks) return;
    const Chunk ch = a.chunks_tmp[k];
    const uint64_t tb = ch.task_begin & TB_IDX_MASK;
    const uint32_t n = uint32_t(ch.dst_n >> 48) & CHUNK_N_MASK;
    uint64_t key = 0;
    for (uint32_t q = 0; q < n && q < 6u && tb + q < n_desc; ++q) {
        const uint64_t d = a.desc[tb + q];
        const bool snv = (d & SNV3_MARK) == SNV3_MARK || (d >> 60) == 0xDull;
        const uint64_t src = snv ? (d & SNV3_MAX_SRC) : (d & SRC_MASK);
        if ((snv || (d >> 62) == SPACE_PROTEOME) && src < a.proteome_len) { key = src; break; }
    }
    const uint64_t per = (a.proteome_len + 7) / 8;
    const uint64_t bk = per ? key / per : 0;
    const uint8_t bucket = uint8_t(bk < 8 ? bk : 7);
    a.bucket[k] = bucket;
    a.sub[k] = xcd_sub_window(key, bucket, per);
}

}  // namespace

uint64_t rows_scan_scratch_entries(uint64_t n) { return (n + RS_TILE - 1) / RS_TILE + 2; }

hipError_t launch_rows_tile_bytes(const RowsArgs& a, uint64_t* scan_scratch, hipStream_t stream)
{
    const uint64_t n_lanes = a.n_tiles << a.log2K;
    hipLaunchKernelGGL(rows_tile_bytes_kernel, dim3(uint32_t((n_lanes + 255) / 256)), dim3(256), 0, stream, a);
    const uint64_t n = a.n_tiles, n_t = (n + RS_TILE - 1) / RS_TILE;
    hipLaunchKernelGGL(rows_scan_sums, dim3(uint32_t(n_t)), dim3(256), 0, stream, a.tile_bytes, n, scan_scratch);
    hipLaunchKernelGGL(rows_scan_tiles, dim3(1), dim3(1024), 0, stream, scan_scratch, n_t);
    hipLaunchKernelGGL(rows_scan_apply, dim3(uint32_t(n_t)), dim3(256), 0, stream, a.tile_bytes, n, scan_scratch, a.tile_res_base);
    return hipGetLastError();
}

static_assert(ROWS_PAD == ROWS_PAD_SLOTS, "build_rows.h");

template <int MODE, bool FASTA>
static hipError_t launch_parse_t(const RowsArgs& a, int phase, hipStream_t stream)
{
    const dim3 grid{uint32_t(a.n_tiles)};
    if (phase == PH_PAD) hipLaunchKernelGGL((rows_parse_kernel<MODE, FASTA, PH_PAD>), grid, dim3(64), 0, stream, a);
    else if (phase == PH_COUNT) hipLaunchKernelGGL((rows_parse_kernel<MODE, FASTA, PH_COUNT>), grid, dim3(64), 0, stream, a);
    else hipLaunchKernelGGL((rows_parse_kernel<MODE, FASTA, PH_DIRECT>), grid, dim3(64), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_rows_parse(const RowsArgs& a, int mode, bool fasta, int phase, hipStream_t stream)
{
    if (a.n_tiles == 0) return hipSuccess;
    if (a.n_tiles > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (mode == ROWS_DENSE) return fasta ? launch_parse_t<ROWS_DENSE, true>(a, phase, stream) : launch_parse_t<ROWS_DENSE, false>(a, phase, stream);
    return fasta ? launch_parse_t<ROWS_WAVE, true>(a, phase, stream) : launch_parse_t<ROWS_WAVE, false>(a, phase, stream);
}

hipError_t launch_rows_compact(const RowsArgs& a, hipStream_t stream)
{
    if (a.n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(rows_compact_kernel, dim3(uint32_t((a.n_tiles + 3) / 4)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_rows_hap_begin(const RowsArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(rows_hap_begin_kernel, dim3(uint32_t((a.n_haps + 1 + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_rows_cut(const RowsArgs& a, int mode, bool emit, hipStream_t stream)
{
    if (a.n_segs == 0) return hipSuccess;
    const uint32_t max_rows = mode == ROWS_DENSE ? ROWS_MAX_DENSE : ROWS_MAX_WAVE, max_desc = mode == ROWS_DENSE ? CHUNK_TASKS_DEEP : CHUNK_TASKS_WAVE;
    const uint64_t flag = mode == ROWS_DENSE ? CHUNK_DENSE : CHUNK_WAVE;
    if (emit) hipLaunchKernelGGL(rows_cut_kernel<true>, dim3(uint32_t(a.n_segs)), dim3(64), 0, stream, a, max_rows, max_desc, flag);
    else hipLaunchKernelGGL(rows_cut_kernel<false>, dim3(uint32_t(a.n_segs)), dim3(64), 0, stream, a, max_rows, max_desc, flag);
    return hipGetLastError();
}

hipError_t launch_rows_keys(const RowsArgs& a, uint64_t n_chunks, uint64_t n_desc, hipStream_t stream)
{
    if (n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(rows_keys_kernel, dim3(uint32_t((n_chunks + 255) / 256)), dim3(256), 0, stream, a, n_chunks, n_desc);
    return hipGetLastError();
}

}  // namespace v2p
