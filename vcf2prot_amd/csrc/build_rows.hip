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
constexpr uint32_t ROWS_CAP = 384;             // descriptors a wave stages in LDS per tile
constexpr uint32_t ROWS_CAPR = 128;            // rows of the arena a tile may touch (their cover entries are staged too)
enum : int { PH_STAGE = 0, PH_COUNT = 1, PH_DIRECT = 2 };
constexpr uint64_t LB_AGG = 1ull << 62, LB_PREFIX = 2ull << 62, LB_VALUE = (1ull << 62) - 1;

struct __attribute__((aligned(16))) TxRec { uint64_t poff, alt0, res_base, hsrc; uint32_t ref_len, res_len, n_alt, hl; };
static_assert(sizeof(TxRec) == 48, "three 16-byte LDS reads");

struct WaveLds {
    TxRec tx[65];                              // slot j = transcript t0 - 1 + j
    uint64_t stage[ROWS_CAP];
    uint64_t cover[ROWS_CAPR];
    uint32_t flag[16];                         // one byte per lane: the item is a HEAD
};

// One tile's windows.  PHASE: PH_STAGE descriptors and cover entries into LDS (the caller writes them out once it knows where the
// tile's descriptors start), PH_COUNT nothing (count only), PH_DIRECT straight to the arrays from descriptor `base` on.
// Returns the number of descriptors of the tile; *over: something did not fit the stage.
template <int MODE, int PHASE>
__device__ __forceinline__ uint32_t rows_tile(const RowsArgs& a, WaveLds& L, uint32_t lane, uint64_t t0, uint32_t nh, uint64_t I0, uint64_t I1, uint64_t hp,
                                              uint32_t carry_e0, uint64_t row_first, uint64_t base, bool& over)
{
    constexpr uint32_t CTX = MODE == ROWS_DENSE ? 4u : 2u, ADV = 62u - CTX;
    uint32_t tile_cnt = 0;                     // descriptors of the tile so far (wave-uniform)
    uint32_t carry_e = carry_e0;
    uint64_t carry_h = 0, carry_second = 0;
    bool first = true;
    for (uint64_t R0 = I0; ; R0 += ADV) {
        const uint32_t nvalid = uint32_t(I1 - R0 < 64u ? I1 - R0 : 64u);
        const bool last = R0 + 64u >= I1;
        const uint32_t e_lo = first ? 0u : CTX, e_hi = last ? nvalid : 62u;
        // ---- which items are HEADs, and every item's transcript ----
        if (lane < 16u) L.flag[lane] = 0u;
        asm volatile("" ::: "memory");
        const bool my_head_in = lane < nh && hp >= R0 && hp < R0 + 64u;
        if (my_head_in) reinterpret_cast<uint8_t*>(L.flag)[uint32_t(hp - R0)] = 1u;
        asm volatile("" ::: "memory");
        const bool active = lane < nvalid;
        const bool isHead = reinterpret_cast<const uint8_t*>(L.flag)[lane] != 0u;      // (lanes >= nvalid: no HEAD of this tile lies there)
        const uint64_t headmask = __ballot(isHead);
        const uint32_t heads_before = uint32_t(__popcll(__ballot(lane < nh && hp < R0)));
        const uint32_t slot = heads_before + mbcnt(headmask) + (isHead ? 1u : 0u);       // slot of the item's transcript (a HEAD: the one it opens)
        const bool isTask = active && !isHead;
        const uint64_t item = R0 + lane;
        const uint64_t ti = item - (t0 + slot - 1u) - 1u;                                // task index
        // ---- the stream: one coalesced load per array ----
        uint32_t code = 0, sp = 0, ln = 0, sr = 0;
        if (isTask) { code = a.code[ti]; sp = a.start_pos[ti]; ln = a.length[ti]; sr = a.start_pos_res[ti]; }
        const TxRec x = L.tx[slot < 65u ? slot : 0u];
        // ---- update_task / Task::execute checks; result positions ----
        const bool res_oob = isTask && uint64_t(sr) + ln > x.res_len;
        const uint32_t e = isTask && !res_oob ? sr + ln : 0u;                             // end of the task inside its transcript's result
        const uint32_t pe = up1(e, carry_e);                                              // ... of the item before
        uint32_t why = 0;
        if (isTask) {
            if (code > 1u) why = STATUS_BAD_CODE;
            else if (res_oob) why = STATUS_RES_OOB;
            else if (uint64_t(sp) + ln > (code == 0u ? x.ref_len : x.n_alt)) why = STATUS_SRC_OOB;
            else if (sr < pe) why = STATUS_NOT_CONTIGUOUS;
        }
        const bool in_emit = lane >= e_lo && lane < e_hi;
        if (why != 0u && in_emit && PHASE != PH_DIRECT) rreport(a.status, ti, why);
        const bool good = isTask && why == 0u;
        // ---- classes of the fusion state machine ----
        const bool isRef = good && code == 0u;
        const bool imm = good && code == 1u && ln - 1u < IMM_MAX_BYTES;
        const uint64_t src = (isRef ? x.poff : x.alt0) + sp;
        uint64_t lit = 0;
        if (imm) {                                                                        // short alt payloads travel inside their descriptor
            struct __attribute__((packed, aligned(1))) U64 { uint64_t v; };
            lit = reinterpret_cast<const U64*>(a.alt + src)->v & (~0ull >> (64u - 8u * ln));
        }
        const bool ps = isRef && ln <= SNV3_MAX_LEN;
        const bool cA = ps && src + ln + 1u + SNV3_MAX_LEN <= SNV3_MAX_SRC;
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
        // ---- what the lane emits: up to three runs of result bytes, in order ----
        //   HEAD: '.' fill of the transcript it closes, that transcript's line feed, its own header;  task: '.' fill of a gap, itself
        const TxRec pv = L.tx[isHead ? slot - 1u : 0u];
        uint64_t w0 = 0, w1 = 0, w2 = 0;                               // descriptor words (w0, w1: plain words whose length may exceed a piece)
        uint64_t l0 = 0, l1 = 0, l2 = 0, q0 = 0, q1 = 0, q2 = 0;      // lengths, arena positions
        if (isHead && active) {
            if (pv.res_len > pe) { l0 = pv.res_len - pe; q0 = pv.res_base + pv.hl + pe; w0 = uint64_t(SPACE_FILL) << 62; }
            if (pv.hl) { l1 = 1; q1 = pv.res_base + pv.hl + pv.res_len; w1 = ((pv.hsrc + pv.hl - 1u) & SRC_MASK) | (uint64_t(SPACE_PROTEOME) << 62); }
            if (x.hl) { l2 = x.hl; q2 = x.res_base; w2 = (x.hsrc & SRC_MASK) | (uint64_t(SPACE_PROTEOME) << 62); }
        } else if (good) {
            const uint64_t posbase = x.res_base + x.hl;
            if (gap) { l0 = sr - pe; q0 = posbase + pe; w0 = uint64_t(SPACE_FILL) << 62; }
            if (isF) {
                if (!isAbs) {
                    if (MODE == ROWS_DENSE && isSecond) {
                        l1 = uint64_t(p_len1) + 1u + p_len2 + 1u + ln; q1 = posbase + sr - 1u - p_len2 - 1u - p_len1;
                        w1 = SNV5_MARK | (uint64_t(f_byte) << 52) | (uint64_t(p_byte) << 44) | (uint64_t(ln & 31u) << 39) | (uint64_t(p_len2 & 31u) << 34) | (uint64_t(p_len1 & 31u) << 29) | (uint64_t(p_run) & SNV3_MAX_SRC);
                    } else {
                        l1 = uint64_t(f_len1) + 1u + ln; q1 = posbase + sr - 1u - f_len1;
                        w1 = SNV3_MARK | (uint64_t(f_byte) << 53) | (uint64_t(ln & 0xFFFu) << 41) | (uint64_t(f_len1 & 0xFFFu) << 29) | (uint64_t(f_run) & SNV3_MAX_SRC);
                    }
                }
            } else if (!isAbs && ln != 0u) {
                l1 = ln; q1 = posbase + sr;
                w1 = imm ? (lit | (uint64_t(SPACE_IMM) << 62)) : ((src & SRC_MASK) | (uint64_t(isRef ? SPACE_PROTEOME : SPACE_PAYLOAD) << 62));
            }
        }
        if (!in_emit) { l0 = 0; l1 = 0; l2 = 0; }
        // pieces: a run longer than a descriptor's length field (a '.' fill or copy of more than 4 MiB) is several descriptors
        const bool fusedw = good && isF;                                                   // (its word is complete: the length is not a field of it)
        const uint32_t n0 = l0 ? uint32_t((l0 + PIECE_MAX - 1u) / PIECE_MAX) : 0u;
        const uint32_t n1 = l1 ? (fusedw ? 1u : uint32_t((l1 + PIECE_MAX - 1u) / PIECE_MAX)) : 0u;
        const uint32_t n2 = l2 ? uint32_t((l2 + PIECE_MAX - 1u) / PIECE_MAX) : 0u;
        const uint32_t cnt = n0 + n1 + n2;
        const uint32_t incl = wave_incl_scan(cnt);
        const uint32_t round_total = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
        if (PHASE != PH_COUNT && round_total != 0u) {
            uint32_t k = tile_cnt + incl - cnt;                                           // the lane's first slot, tile-relative
            auto put = [&](uint64_t word, uint64_t len, uint64_t pos, bool whole) {
                // one run: pieces of at most PIECE_MAX bytes (whole: a ready fused word)
                const unsigned space = unsigned(word >> 62);
                uint64_t sfield = word & SRC_MASK;
                while (len) {
                    const uint32_t piece = whole ? uint32_t(len) : uint32_t(len < PIECE_MAX ? len : PIECE_MAX);
                    const uint64_t dword = whole ? word : ((word & ~SRC_MASK) | (sfield & SRC_MASK) | (uint64_t(piece) << 40));
                    if (PHASE == PH_STAGE) { if (k < ROWS_CAP) L.stage[k] = dword; }
                    else if (base + k < a.desc_cap) a.desc[base + k] = dword;
                    // rows whose first byte lies inside the piece
                    for (uint64_t r = (pos + ROW_BYTES - 1u) / ROW_BYTES; r * ROW_BYTES < pos + piece; ++r) {
                        if (r == 0u) continue;
                        const uint64_t off = r * ROW_BYTES - pos;
                        if (PHASE == PH_STAGE) { const uint64_t rr = r - row_first; if (rr < ROWS_CAPR) L.cover[rr] = (uint64_t(k) << 22) | off; }
                        else if (r < a.n_rows) a.cover[r] = ((base + k) << 22) | off;
                    }
                    if (space == SPACE_PROTEOME || space == SPACE_PAYLOAD) sfield += piece;      // (an immediate is never cut: <= 5 bytes)
                    len -= piece; pos += piece; ++k;
                    if (whole) break;
                }
            };
            if (l0) put(w0, l0, q0, false);
            if (l1) put(w1, l1, q1, fusedw);
            if (l2) put(w2, l2, q2, false);
        }
        tile_cnt += round_total;
        if (last) break;
        carry_h = p.h >> ADV; carry_second = second >> ADV;
        carry_e = uint32_t(__builtin_amdgcn_readlane(int(e), int(ADV - 1u)));
        first = false;
    }
    if (PHASE == PH_STAGE && tile_cnt > ROWS_CAP) over = true;
    return tile_cnt;
}

// decoupled look-back over the tiles' descriptor counts: returns the number of descriptors before `tile`
__device__ __forceinline__ uint64_t rows_lookback(uint64_t* state, uint64_t tile, uint64_t count, uint32_t lane)
{
    if (tile == 0) {
        if (lane == 0) __hip_atomic_store(&state[0], LB_PREFIX | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0;
    }
    if (lane == 0) __hip_atomic_store(&state[tile], LB_AGG | count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t sum = 0;
    int64_t j = int64_t(tile) - 1;
    for (;;) {
        const int64_t idx = j - int64_t(lane);
        const uint64_t v = idx >= 0 ? __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : LB_PREFIX;
        const uint64_t pm = __ballot((v >> 62) == 2ull), em = __ballot((v >> 62) == 0ull);
        const uint32_t firstP = pm ? uint32_t(__builtin_ctzll(pm)) : 64u;
        const uint64_t need = firstP >= 63u ? ~0ull : ((2ull << firstP) - 1ull);          // lanes 0 .. firstP must have published
        if (em & need) { __builtin_amdgcn_s_sleep(2); continue; }
        sum += wave_sum64(lane <= firstP ? (v & LB_VALUE) : 0ull);
        if (firstP < 64u) break;
        j -= 64;
    }
    if (lane == 0) __hip_atomic_store(&state[tile], LB_PREFIX | (sum + count), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return sum;
}

template <int MODE, bool TWO_PHASE>
__global__ __launch_bounds__(64) void rows_parse_kernel(RowsArgs a)
{
    __shared__ WaveLds L;
    const uint32_t lane = threadIdx.x;
    const uint64_t n_heads = a.n_tx + 1u;
    for (uint64_t tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
        const uint64_t t0 = tile << a.log2K;
        const uint32_t nh = uint32_t(n_heads - t0 < a.K ? n_heads - t0 : a.K);
        // ---- the tile's transcripts: slot lane + 1 = transcript t0 + lane, slot 0 = the one before the tile ----
        const uint64_t tile_base = a.tile_res_base[tile];
        uint64_t hp = ~0ull;
        {
            const uint64_t u = t0 + lane;
            const bool valid = lane < nh && u < a.n_tx;
            TxRec r{0, 0, 0, 0, 0, 0, 0, 0};
            uint64_t alen = 0;
            if (lane < nh) hp = (u <= a.n_tx ? a.tx_task_begin[u] : a.n_tasks) + u;
            if (valid) {
                r.poff = a.tx_proteome_off[u]; r.alt0 = a.tx_alt_begin[u]; r.n_alt = uint32_t(a.tx_alt_begin[u + 1] - r.alt0);
                r.ref_len = a.tx_ref_len[u]; r.res_len = a.tx_res_len[u];
                r.hl = a.tx_header_len ? a.tx_header_len[u] : 0u;
                r.hsrc = r.hl ? a.proteome_len + a.tx_header_off[u] : 0ull;
                alen = uint64_t(r.res_len) + (r.hl ? r.hl + 1u : 0u);
                if (r.poff + r.ref_len > a.proteome_len) rreport(a.status, a.tx_task_begin[u], STATUS_SRC_OOB);   // transcript outside the resident proteome
            }
            r.res_base = tile_base + wave_incl_scan64(alen, lane) - alen;
            if (lane < nh) L.tx[lane + 1u] = r;
            if (lane == 0) {
                TxRec q{0, 0, 0, 0, 0, 0, 0, 0};
                if (t0 > 0) {
                    const uint64_t v = t0 - 1u;
                    q.res_len = a.tx_res_len[v]; q.hl = a.tx_header_len ? a.tx_header_len[v] : 0u;
                    q.hsrc = q.hl ? a.proteome_len + a.tx_header_off[v] : 0ull;
                    q.res_base = tile_base - (uint64_t(q.res_len) + (q.hl ? q.hl + 1u : 0u));
                }
                L.tx[0] = q;
            }
        }
        asm volatile("" ::: "memory");
        const uint64_t I0 = t0 <= a.n_tx ? a.tx_task_begin[t0] + t0 : a.n_tasks + n_heads;
        const uint64_t t1 = t0 + nh;
        const uint64_t I1 = t1 <= a.n_tx ? a.tx_task_begin[t1] + t1 : a.n_tasks + n_heads;
        // end of the last task before the tile: HEAD(t0) fills the rest of the transcript before it with '.'
        uint32_t carry_e = 0;
        if (t0 > 0 && a.tx_task_begin[t0] > a.tx_task_begin[t0 - 1u]) { const uint64_t i = a.tx_task_begin[t0] - 1u; carry_e = a.start_pos_res[i] + a.length[i]; }
        // The tile's descriptors start where the transcript BEFORE it left off (HEAD(t0) writes that one's '.' fill and line feed) and
        // end where its own last transcript's tasks do (the HEAD of the next tile closes it): the rows they may cover
        const TxRec q0 = L.tx[0];
        const uint64_t emit_start = tile_base - ((q0.res_len > carry_e ? q0.res_len - carry_e : 0u) + (q0.hl ? 1u : 0u));
        const uint64_t row_first = (emit_start + ROW_BYTES - 1u) / ROW_BYTES;
        const uint64_t tile_end = a.tile_res_base[tile + 1u];
        bool over = false;
        if (!TWO_PHASE) {
            const uint64_t n_rows_tile = tile_end > row_first * ROW_BYTES ? (tile_end - row_first * ROW_BYTES + ROW_BYTES - 1u) / ROW_BYTES : 0u;
            for (uint32_t k = lane; k < ROWS_CAPR; k += 64u) L.cover[k] = ~0ull;
            asm volatile("" ::: "memory");
            const uint32_t n = rows_tile<MODE, PH_STAGE>(a, L, lane, t0, nh, I0, I1, hp, carry_e, row_first, 0, over);
            if (n_rows_tile > ROWS_CAPR) over = true;
            const uint64_t base = rows_lookback(a.tile_state, tile, n, lane);
            if (over) { if (lane == 0) rreport(a.status, tile, STATUS_ROWS_STAGE); }
            else if (base + n > a.desc_cap) { if (lane == 0) rreport(a.status, tile, STATUS_ROWS_CAP); }
            else {
                asm volatile("" ::: "memory");
                for (uint32_t k = lane; k < n; k += 64u) a.desc[base + k] = L.stage[k];
                for (uint32_t k = lane; k < n_rows_tile; k += 64u) {                  // (rows behind the tile's last task belong to the next tile)
                    const uint64_t r = row_first + k, c = L.cover[k];
                    if (c != ~0ull && r >= 1u && r < a.n_rows) a.cover[r] = c + (base << 22);
                }
            }
            if (tile + 1u == a.n_tiles && lane == 0) a.totals[0] = base + n;
        } else {
            const uint32_t n = rows_tile<MODE, PH_COUNT>(a, L, lane, t0, nh, I0, I1, hp, carry_e, row_first, 0, over);
            const uint64_t base = rows_lookback(a.tile_state, tile, n, lane);
            if (base + n > a.desc_cap) { if (lane == 0) rreport(a.status, tile, STATUS_ROWS_CAP); }
            else (void)rows_tile<MODE, PH_DIRECT>(a, L, lane, t0, nh, I0, I1, hp, carry_e, row_first, base, over);
            if (tile + 1u == a.n_tiles && lane == 0) a.totals[0] = base + n;
        }
        asm volatile("" ::: "memory");
    }
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
    const uint64_t n_desc = a.totals[0];                                                  // (written by the parse's last tile)
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
        uint64_t c = 0;
        if (r >= 1u && r < a.n_rows) c = a.cover[r];
        const uint64_t idx = c >> 22;
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
                const uint64_t n = lastd - f + 1u, r1 = b + hb;
                const uint64_t rows = r1 >= a.n_rows ? 0u : uint64_t(hb - cur);
                a.chunks_tmp[out_k] = Chunk{f | (uint64_t(hs) << TB_IDX_BITS), ((b + cur) * ROW_BYTES) | rows | (n << 48) | CHUNK_CLIP | flag};
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

template <int MODE, bool TP>
static hipError_t launch_parse_t(const RowsArgs& a, hipStream_t stream)
{
    int dev = 0, cus = 0, per_cu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rows_parse_kernel<MODE, TP>, 64, 0);
    if (e != hipSuccess) return e;
    if (per_cu < 1) per_cu = 1;
    // persistent waves, ALL co-resident: the look-back of a tile only ever waits for tiles of waves that are running
    uint64_t grid = uint64_t(cus) * uint64_t(per_cu);
    if (grid > a.n_tiles) grid = a.n_tiles;
    hipLaunchKernelGGL((rows_parse_kernel<MODE, TP>), dim3(uint32_t(grid)), dim3(64), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_rows_parse(const RowsArgs& a, int mode, bool two_phase, hipStream_t stream)
{
    if (a.n_tiles == 0) return hipSuccess;
    if (mode == ROWS_DENSE) return two_phase ? launch_parse_t<ROWS_DENSE, true>(a, stream) : launch_parse_t<ROWS_DENSE, false>(a, stream);
    return two_phase ? launch_parse_t<ROWS_WAVE, true>(a, stream) : launch_parse_t<ROWS_WAVE, false>(a, stream);
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
