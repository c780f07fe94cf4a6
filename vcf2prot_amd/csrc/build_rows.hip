// build_rows.hip -- the device image of a batch built in ONE pass over the transcript stream (ROWS images: rows_image.hpp).
//
// Replaces, for wave and dense images, round 3's build_kernels.hip (one lane per transcript walking its tasks twice -- a counting
// pass and an emitting pass, 12.3 ms for the north star's cohort whose execute takes 7.6).  Here
//   tile_bytes   arena bytes per tile of K transcripts, scanned: res_counter of haplotype_instruction.rs:90,132 per tile
//   parse        lane = ITEM of the stream (a Task, task.rs:2-9; a transcript's first item also opens it, its last also closes it).
//                Every stream array is read once, coalesced, a window ahead of its use; update_task's checks
//                (haplotype_instruction.rs:140-158) and Task::execute's bounds (task.rs:43,47) are what the device reports instead
//                of panicking; the packer's fusion state machine is solved for 64 items at once on ballot masks (rows_parse);
//                descriptors -- whole, nothing is cut -- go to the tile's slots of a padded array, compacted after a scan of the
//                tiles' counts; every 1 KiB row of the arena learns which descriptor covers its first byte
//   cut          one wave per segment of 640 rows walks the row map greedily: as many rows as one wave takes (ten) while the
//                descriptors fit its lanes; counted and emitted in one pass (padded chunk table, compacted after the scan)
//   keys         proteome slice and window of every chunk for the XCD / window order (build_kernels.hip: launch_order_blocks)
// Integer / index work only (no MFMA); the parse is bound by instruction issue -- vector and scalar alike -- see DESIGN.md 8.2a.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "build_rows.h"
#include "build_kernels.h"
#include "stitch_device.hpp"

namespace v2p {

namespace {

__device__ __forceinline__ void rreport(unsigned long long* status, uint64_t index, uint32_t reason) { atomicMin(status, (unsigned long long)((index << 8) | reason)); }
__device__ __forceinline__ uint32_t mbcnt(uint64_t m) { return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u)); }
// the lane's bit of a wave-uniform mask: one v_cndmask on the SGPR pair, no shifts
__device__ __forceinline__ bool lane_bit(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
// lane i receives lane i - 1's value, lane 0 `first` (DPP wave_shr:1)
__device__ __forceinline__ uint32_t up1(uint32_t x, uint32_t first) { return uint32_t(__builtin_amdgcn_update_dpp(int(first), int(x), 0x138, 0xf, 0xf, false)); }
// ... lane 0 receives 0: bound_ctrl writes the zero itself, no register to initialise in front of every shift
__device__ __forceinline__ uint32_t up1z(uint32_t x) { return uint32_t(__builtin_amdgcn_update_dpp(0, int(x), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ uint32_t up2(uint32_t x) { return up1z(up1z(x)); }
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
// lane = transcript (coalesced).  Tiles are K (1 .. 64, any number) consecutive transcripts, so a wave's 64 lanes are a few runs of
// equal tile: a segmented scan over the wave, then one global add per run (tile_bytes is zeroed by the launcher; two waves meet in
// a tile at most)
__global__ __launch_bounds__(256) void rows_tile_bytes_kernel(RowsArgs a)
{
    const uint64_t t = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const bool valid = t < a.n_tx;
    uint64_t v = 0;
    if (valid) {
        const uint32_t hl = a.tx_header_len ? a.tx_header_len[t] : 0u;
        v = uint64_t(a.tx_res_len[t]) + (hl ? hl + 1u : 0u);
    }
    const uint64_t t_wave = t - lane;
    const uint32_t seg = uint32_t((valid ? t : a.n_tx - 1u) / a.K - t_wave / a.K);     // the lane's tile, counted from the wave's first
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint64_t y = __shfl_up(v, o);
        const uint32_t so = uint32_t(__shfl_up(int(seg), o));
        if (lane >= uint32_t(o) && so == seg) v += y;
    }
    const uint32_t next = uint32_t(__shfl_down(int(seg), 1));
    if (valid && (lane == 63u || next != seg || t + 1u == a.n_tx)) atomicAdd(reinterpret_cast<unsigned long long*>(a.tile_bytes) + t / a.K, (unsigned long long)v);
}

// the same for tiles of at least half a wave: one wave per tile, four tiles per workgroup, the sum by two 32-bit DPP scans (64
// lengths below 2^32: their low 26 bits sum below 2^32, their high 6 bits below 2^12) instead of 64-bit shuffles through the LDS
__global__ __launch_bounds__(256) void rows_tile_bytes_wave_kernel(RowsArgs a)
{
    const uint64_t tile = uint64_t(blockIdx.x) * 4u + (threadIdx.x >> 6);
    if (tile >= a.n_tiles) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t t = tile * a.K + lane;
    uint32_t len = 0;
    if (lane < a.K && t < a.n_tx) {
        const uint32_t hl = a.tx_header_len ? a.tx_header_len[t] : 0u;
        const uint64_t l = uint64_t(a.tx_res_len[t]) + (hl ? hl + 1u : 0u);
        if ((l >> 31) && a.status) rreport(a.status, a.tx_task_begin[t], STATUS_ROWS_SPAN);   // (the parse refuses its tile: more than 2 GiB)
        len = (l >> 31) ? 0x80000000u : uint32_t(l);                           // (saturated: the tile's sum stays above 2 GiB whatever l was)
    }
    const uint32_t lo = wave_incl_scan(len & 0x3FFFFFFu), hi = wave_incl_scan(len >> 26);
    if (lane == 63u) a.tile_bytes[tile] = uint64_t(lo) + (uint64_t(hi) << 26);
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

// workgroup b of a grid of G -> the j-th item of the contiguous range XCD b % 8 owns (a bijection of [0, G) for any G)
__device__ __forceinline__ uint32_t xcd_contiguous(uint32_t b, uint32_t G)
{
    const uint32_t x = b & 7u, j = b >> 3, q = G >> 3, r = G & 7u;
    return x * q + (x < r ? x : r) + j;
}

// ---- the parse ------------------------------------------------------------------------------------------------------------------
// What bounds it (measured, profiles/r04_build_*, r04_sq_counters_parse_kernel_C3.txt): instruction issue, ~200 vector and ~190
// scalar instructions per window of 64 items in round 4 -- 142 and 158 at the end of round 6 (profiles/r06_parse_variants.txt,
// r06_sq_counters_parse_*_end.txt), where a fifth fewer instructions bought 4-9 % of time: what is left is as much the latency of a
// window's own chain (flags -> transcript -> records -> literal -> mask solver -> emit) as the issue of it -- once a tile's dependent
// loads are requested together (below).  A burst of the tile's
// Task arrays into LDS with global_load_lds changed nothing; a decoupled look-back for the descriptors' final place cost a third
// of the kernel (a tile waits for every tile before it).  So: one tile per 64-lane workgroup, plain grid; descriptors go
// to a PADDED array -- ROWS_PAD slots per tile -- and a copy kernel compacts them once a scan of the tiles' counts has told every
// tile where it starts; positions are 32-bit offsets from the tile's first emitted byte; per-transcript values sit in LDS as
// one 32-byte record each; everything rare (runs of more than 1 KiB, which may cross two rows, or of more than a descriptor's length
// field) is behind one wave-uniform branch.
constexpr uint32_t ROWS_PAD = ROWS_TILE_SLOTS;             // descriptor slots per tile in the padded array (a tile with more: the two-pass form)
enum : int { PH_PAD = 0, PH_COUNT = 1, PH_DIRECT = 2 };

constexpr uint32_t ROWS_ALT_LDS = 512;         // alt bytes of a tile kept in LDS (64 lanes x 8)
struct __attribute__((aligned(16))) WaveLds {
    // One 32-byte record per transcript of the tile -- { res_len, pos, bound[0], bound[1] } { base[0], base[1] } -- read by a lane as two
    // 16-byte loads the moment it knows its transcript: which of the two bounds / bases applies (the task's code) is a select on values that
    // have arrived, not an address two more LDS round trips wait for.  pos: arena offset of the transcript's first result byte (behind its
    // FASTA header) minus the tile's first byte; bound[0]: its reference length, [1]: its alt tape's length; base[0]: its proteome offset,
    // [1]: its alt tape's offset (64 bits each).
    uint32_t tx[64][8];
    uint32_t hl[64];                           // FASTA: length of its record header (0: none)
    uint64_t hsrc[64];                         // ... and where the header sits in the resident reference
    uint32_t flag[16];                         // one byte per lane: the item is its transcript's first
    uint32_t alt[ROWS_ALT_LDS / 4u + 2u];      // the tile's alt bytes (when they fit): literals are read from here, not from memory
};

// Items (round 4, second form): every Task is an item, and a transcript WITHOUT tasks contributes one item of its own; there are no
// HEAD items any more -- a transcript's first item also opens it (FASTA: its header), its last item also closes it ('.' fill of the
// cells behind the last task, haplotype_instruction.rs:78; FASTA: the line feed).  A lane therefore emits up to three (FASTA: five)
// runs of result bytes, back to back: [header] ['.' fill of the gap before the task] [the task] ['.' tail] [line feed].
// PHASE: PH_PAD descriptors to the tile's slots of the padded array and its count to tile_count (a tile that does not fit is
// reported); PH_COUNT only the count; PH_DIRECT descriptors to their final place (tile_desc_base: the scan of the counts).
// SRC32 (round 6): every source offset of the launch -- resident reference incl. the header table, the stream's alt bytes -- is below 2^32
// (the launcher checks): a Task's source and the fusion rules' comparisons on it are 32-bit arithmetic.  64-bit adds and compares issue
// at a fraction of the 32-bit rate, and this kernel is bound by instruction issue: eight of them per window were a tenth of its time.
template <int MODE, bool FASTA, int PHASE, bool SRC32>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_sgpr(96))) void rows_parse_kernel(RowsArgs a)
{
    using src_t = typename std::conditional<SRC32, uint32_t, uint64_t>::type;
    constexpr bool TILES = MODE == ROWS_TILES;                       // pieces of a TILE image (dense_pieces.h) instead of descriptors: no row map, no cutter behind this kernel
    constexpr uint32_t CTX = MODE == ROWS_DENSE ? 4u : 2u, ADV = 62u - CTX;
    constexpr int NR = FASTA ? 5 : 3;                                 // runs a lane may emit
    constexpr int RG = FASTA ? 1 : 0, RS = RG + 1, RT = RG + 2;       // run numbers: [0 header] RG gap, RS the task itself, RT tail [4 line feed]
    __shared__ WaveLds L;
    const uint32_t lane = threadIdx.x;
    // (a launch works on tiles [tile0, tile1): the whole stream, or one slice of it.)  Workgroup b runs on XCD b % 8 and every XCD has
    // its own L2: XCD x takes a CONTIGUOUS eighth of the tiles, so that the lines of the Task arrays, the row map and the padded array
    // that two neighbouring tiles share are fetched (and written back) by one L2, not by two
    const uint32_t rel = a.xcd_tiles ? xcd_contiguous(blockIdx.x, gridDim.x) : blockIdx.x;
    const uint64_t tile = a.tile0 + rel;
    const uint64_t t0 = tile * a.K;
    const uint32_t nh = uint32_t(a.n_tx - t0 < a.K ? a.n_tx - t0 : a.K);
    const uint64_t task_lo = a.tx_task_begin[t0], task_end = a.tx_task_begin[t0 + nh];
    const uint64_t alt_lo = a.tx_alt_begin[t0], alt_n = a.tx_alt_begin[t0 + nh] - alt_lo;
    const uint64_t tile_base = a.tile_res_base[tile], tile_end = a.tile_res_base[tile + 1u];
    // positions are 32-bit offsets from the tile's first byte (a tile of more than 2 GiB of result is refused)
    if (tile_end - tile_base > 0x7FFFFFFFull) { if (lane == 0) rreport(a.status, task_lo, STATUS_ROWS_SPAN); if (PHASE != PH_DIRECT && lane == 0) a.tile_count[tile] = 0u; return; }
    // a TILE image: the tile IS the executor's work item -- its result range must fit the executor's LDS image (the caller then builds a dense rows image)
    if (TILES && tile_end - tile_base > a.tile_span_max) { if (lane == 0) { rreport(a.status, tile, STATUS_ROWS_STAGE); a.tile_count[tile] = 0u; } return; }
    const uint32_t eoff = uint32_t(tile_base) & (ROW_BYTES - 1u);    // the tile's first byte inside its row
    const uint64_t erow = tile_base / ROW_BYTES;
    // (the row map through the tile's own base and a 32-bit row number: rows [row_lo, row_lo + row_span) of the tile are rows 1 .. n_rows - 1
    // of the arena -- one 32-bit compare and an SGPR-based store per entry instead of 64-bit adds and two 64-bit compares)
    uint64_t* const cover_tile = a.cover + erow;
    const uint32_t row_lo = erow == 0ull ? 1u : 0u;
    const uint32_t row_span = a.n_rows > erow + row_lo ? uint32_t(a.n_rows - erow - row_lo < 0xFFFFFFFFull ? a.n_rows - erow - row_lo : 0xFFFFFFFFull) : 0u;
    // Everything a tile reads before its first window is requested at once -- the transcripts' tables, the first window's tasks
    // (as if every transcript had tasks: item = task; a tile with an empty transcript reloads), the tile's alt bytes: what bounds
    // this kernel on shallow Task vectors is the chain of dependent loads per tile, not its instructions
    const uint32_t n_tile_tasks = uint32_t(task_end - task_lo);
    // (a tile without a single task -- transcripts that are all '.' -- points its task arrays at memory that is there whatever the stream
    // holds: the loop's request for the next window has no condition around it, see prefetch_clamped)
    const bool any_tasks = n_tile_tasks != 0u;
    const uint8_t* const g_code = any_tasks ? a.code + task_lo : reinterpret_cast<const uint8_t*>(a.tx_task_begin);
    const uint32_t* const g_sp = any_tasks ? a.start_pos + task_lo : reinterpret_cast<const uint32_t*>(a.tx_task_begin);
    const uint32_t* const g_ln = any_tasks ? a.length + task_lo : reinterpret_cast<const uint32_t*>(a.tx_task_begin);
    const uint32_t* const g_sr = any_tasks ? a.start_pos_res + task_lo : reinterpret_cast<const uint32_t*>(a.tx_task_begin);
    const uint32_t last_task = any_tasks ? n_tile_tasks - 1u : 0u;
    uint32_t pf_code = 0, pf_sp = 0, pf_ln = 0, pf_sr = 0;            // the tasks of the window about to be worked on (item = task)
    // (32-bit byte offsets on the wave-uniform bases: one shift per lane instead of three 64-bit address computations)
    auto at32 = [](const uint32_t* base, uint32_t i) -> uint32_t { return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(base) + (i << 2)); };
    auto prefetch = [&](uint32_t i) { pf_code = 0; pf_sp = 0; pf_ln = 0; pf_sr = 0; if (i < n_tile_tasks) { pf_code = g_code[i]; pf_sp = at32(g_sp, i); pf_ln = at32(g_ln, i); pf_sr = at32(g_sr, i); } };
    // (inside the loop: EVERY window requests the next one's tasks, the index clamped to the tile's last task instead of lanes masked off --
    // no zeroing, no exec-mask region, and no `if (!last)` around it, whose phi cost a second register set and eight moves per window; the
    // last window's request is never waited for, lanes beyond the tile's tasks hold some task's values and are masked as inactive lanes are)
    auto prefetch_clamped = [&](uint32_t i) {
        const uint32_t j = i < last_task ? i : last_task;
        pf_code = g_code[j]; pf_sp = at32(g_sp, j); pf_ln = at32(g_ln, j); pf_sr = at32(g_sr, j);
    };
    prefetch(lane);
    const bool alt_in_lds = alt_n <= ROWS_ALT_LDS;
    {
        struct __attribute__((packed, aligned(1))) U64 { uint64_t v; };
        uint64_t ab = 0;
        if (alt_in_lds && uint64_t(lane) * 8u < alt_n) ab = reinterpret_cast<const U64*>(a.alt + alt_lo + lane * 8u)->v;
        if (alt_in_lds) { L.alt[lane * 2u] = uint32_t(ab); L.alt[lane * 2u + 1u] = uint32_t(ab >> 32); }
        if (lane < 2u) L.alt[ROWS_ALT_LDS / 4u + lane] = 0u;
    }
    // ---- the tile's transcripts: lane j = transcript t0 + j ----
    uint32_t hp = 0xFFFFFFFFu;                                       // item (relative to the tile's first) of the lane's transcript's first
    uint64_t ne;                                                     // transcripts of the tile without tasks (they are one item each)
    {
        const uint64_t u = t0 + lane;
        const bool valid = lane < nh;
        uint64_t poff = 0, alt0 = 0, hsrc = 0, tb0 = 0, tb1 = 0;
        uint32_t ref_len = 0, res_len = 0, n_alt = 0, hl = 0, alen = 0;
        if (valid) {
            tb0 = a.tx_task_begin[u]; tb1 = a.tx_task_begin[u + 1];
            poff = a.tx_proteome_off[u]; alt0 = a.tx_alt_begin[u]; n_alt = uint32_t(a.tx_alt_begin[u + 1] - alt0);
            ref_len = a.tx_ref_len[u]; res_len = a.tx_res_len[u];
            if (FASTA) { hl = a.tx_header_len[u]; hsrc = hl ? a.proteome_len + a.tx_header_off[u] : 0ull; }
            alen = res_len + (hl ? hl + 1u : 0u);                    // (< 2^31 in total: checked above)
            if (PHASE != PH_DIRECT && poff + ref_len > a.proteome_len) rreport(a.status, tb0, STATUS_SRC_OOB);   // transcript outside the resident proteome
        }
        ne = __ballot(valid && tb1 == tb0);
        if (valid) hp = uint32_t(tb0 - task_lo) + mbcnt(ne);
        const uint32_t rb = wave_incl_scan(alen) - alen;            // the transcript's first arena byte, from the tile's
        if (valid) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            *reinterpret_cast<u32x4*>(&L.tx[lane][0]) = u32x4{res_len, rb + hl, ref_len, n_alt};
            *reinterpret_cast<u32x4*>(&L.tx[lane][4]) = u32x4{uint32_t(poff), uint32_t(poff >> 32), uint32_t(alt0), uint32_t(alt0 >> 32)};
            if (FASTA) { L.hl[lane] = hl; L.hsrc[lane] = hsrc; }
        }
    }
    asm volatile("" ::: "memory");
    const uint32_t n_items = n_tile_tasks + uint32_t(__popcll(ne));                      // >= nh
    // (a stream without a single transcript -- an empty slice of haplotypes -- is one tile of none: nothing to emit; every window below has
    // at least one item, which its lane masks rely on)
    if (n_items == 0u) { if (PHASE != PH_DIRECT && lane == 0) a.tile_count[tile] = 0u; return; }
    const uint32_t pad_slots = TILES ? a.tile_slots : ROWS_PAD;
    uint64_t* const out = PHASE == PH_PAD ? a.desc_pad + uint64_t(rel) * pad_slots : (PHASE == PH_DIRECT ? a.desc + a.tile_desc_base[tile] : nullptr);
    (void)out;
    const uint32_t out_cap = PHASE == PH_PAD ? pad_slots : 0xFFFFFFFFu;
    // cover entry: tile : 25 | descriptor inside the tile : 16 | what the descriptor has from the row's first byte on : 11 | offset of that byte
    // inside it : 11 (a descriptor is at most 2047 bytes) -- the two-pass form: 1 << 63 | descriptor << 22 | rest << 11 | offset.  With the
    // rest in the map the cutter never reads a descriptor (round 4 looked every row's up: 2.4 GB of scattered lines) and does not wait
    // for the compaction.
    const uint64_t dbase = PHASE == PH_DIRECT ? a.tile_desc_base[tile] : 0ull;
    auto cover_word = [&](uint32_t dk, uint32_t off, uint32_t rest) -> uint64_t {
        const uint32_t low = off | (rest << 11);
        return PHASE == PH_DIRECT ? (1ull << 63) | ((dbase + dk) << 22) | low : (tile << 38) | (uint64_t(dk) << 22) | low;
    };

    uint32_t tile_cnt = 0;                     // descriptors of the tile so far (wave-uniform)
    uint64_t carry_h = 0, carry_second = 0;
    uint32_t carry_e = 0;
    bool first = true;
    for (uint32_t R0 = 0; ; R0 += ADV) {
        const bool last = R0 + 64u >= n_items;
        // ---- which items open a transcript, and every item's transcript ----
        if (lane < 16u) L.flag[lane] = 0u;
        asm volatile("" ::: "memory");
        if (hp - R0 < 64u) reinterpret_cast<uint8_t*>(L.flag)[hp - R0] = 1u;
        asm volatile("" ::: "memory");
        // (Booleans live as 64-bit lane MASKS from here on, built from ballots of single comparisons and combined by scalar
        // instructions; a lane reads its bit with lane_bit() where a select or a branch needs it.  Written as per-lane bools combined
        // with && / || the compiler re-ballots every combination -- v_cndmask + v_cmp, thirteen times per window: a seventh of the
        // kernel's vector instructions, and the vector unit is what bounds it.)
        const uint32_t rem = n_items - R0;                                               // (>= 1: a tile has at least one item)
        const uint64_t m_active = ~0ull >> (rem >= 64u ? 0u : 64u - rem);
        const bool active = lane_bit(m_active);
        const uint64_t firstmask = __ballot(reinterpret_cast<const uint8_t*>(L.flag)[lane] != 0u);     // (inactive lanes: no transcript of this tile starts there)
        const bool isFirst = lane_bit(firstmask);
        const uint32_t heads_before = uint32_t(__popcll(__ballot(hp < R0)));
        const uint32_t slot = heads_before + mbcnt(firstmask) + (isFirst ? 1u : 0u) - 1u;   // the item's transcript (item 0 opens one: never negative for an active lane)
        // the transcript's last item: the next one opens another, or the tile ends
        const uint64_t lastmask = (firstmask >> 1) | (last ? m_active ^ (m_active >> 1) : 0ull);
        const bool isLast = lane_bit(lastmask);
        uint32_t ti = R0 + lane;                                                         // task, relative to the tile's first
        uint64_t m_empty = 0ull;
        // this window's tasks were requested a window ago; the next window's are requested now
        uint32_t code = pf_code, sp = pf_sp, ln = pf_ln, sr = pf_sr;
        prefetch_clamped(R0 + ADV + lane);
        if (ne) {                                                                        // (uniform, rare: transcripts without tasks in this tile)
            const uint64_t below = ne & ((1ull << (slot & 63u)) - 1ull);
            ti -= uint32_t(__popcll(below));
            const bool isEmpty0 = isFirst && ((ne >> (slot & 63u)) & 1ull);
            m_empty = __ballot(isEmpty0) & m_active;
            code = 0; sp = 0; ln = 0; sr = 0;
            if (active && !isEmpty0) { code = g_code[ti]; sp = g_sp[ti]; ln = g_ln[ti]; sr = g_sr[ti]; }
        }
        const uint64_t m_task = m_active & ~m_empty;
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 txa = *reinterpret_cast<const u32x4*>(&L.tx[slot & 63u][0]), txb = *reinterpret_cast<const u32x4*>(&L.tx[slot & 63u][4]);
        const uint32_t res_len = txa[0], pos0 = txa[1];
        // ---- update_task / Task::execute checks; result positions ----
        // (the comparison on a laundered copy: seeing `ln > res_len` next to `res_len - ln` the compiler folds them into one subtract-with-borrow
        // and then needs a select and a second compare to turn the borrow back into a lane mask)
        uint32_t ln_o = ln;
        asm("" : "+v"(ln_o));
        const uint64_t m_res_oob = (__ballot(ln_o > res_len) | __ballot(sr > res_len - ln)) & m_task;
        const uint32_t e = lane_bit(m_task & ~m_res_oob) ? sr + ln : 0u;                  // end of the task inside its transcript's result
        const uint32_t pe_raw = up1(e, carry_e);                                          // ... of the item before
        const uint32_t pe = isFirst ? 0u : pe_raw;
        const uint64_t m_code1 = __ballot(code == 1u);
        const bool code1 = lane_bit(m_code1);
        const uint32_t bound = code1 ? txa[3] : txa[2];
        const uint64_t m_src_oob = __ballot(ln_o > bound) | __ballot(sp > bound - ln);
        const uint64_t m_bad = m_task & (__ballot(code > 1u) | m_res_oob | m_src_oob | __ballot(sr < pe));
        const uint64_t m_in_emit = (last ? m_active : (1ull << 62) - 1ull) & (first ? ~0ull : ~((1ull << CTX) - 1ull));      // lanes [e_lo, e_hi)
        const bool in_emit = lane_bit(m_in_emit);
        if (PHASE != PH_DIRECT && (m_bad & m_in_emit)) {                                  // (rare: which rule, in update_task's order)
            const bool res_oob = lane_bit(m_res_oob), src_oob = lane_bit(m_src_oob);
            const uint32_t why = code > 1u ? STATUS_BAD_CODE : (res_oob ? STATUS_RES_OOB : (src_oob ? STATUS_SRC_OOB : STATUS_NOT_CONTIGUOUS));
            if (lane_bit(m_bad & m_in_emit)) rreport(a.status, task_lo + ti, why);
        }
        const uint64_t m_good = m_task & ~m_bad;
        // ---- classes of the fusion state machine ----
        const uint64_t m_isRef = m_good & __ballot(code == 0u);
        const uint64_t m_imm = m_good & m_code1 & __ballot(ln - 1u < IMM_MAX_BYTES);
        const bool isRef = lane_bit(m_isRef), imm = lane_bit(m_imm);
        const src_t src = (SRC32 ? src_t(code1 ? txb[2] : txb[0]) : src_t(code1 ? (uint64_t(txb[3]) << 32) | txb[2] : (uint64_t(txb[1]) << 32) | txb[0])) + sp;
        uint64_t lit = 0;
        if (alt_in_lds) {                                                                 // short alt payloads travel inside their descriptor
            const uint32_t rel = imm ? uint32_t(src - src_t(alt_lo)) : 0u;                // (< ROWS_ALT_LDS: inside the transcript's alt tape, checked above)
            const uint32_t d0 = L.alt[rel >> 2], d1 = L.alt[(rel >> 2) + 1u];
            const uint64_t v = ((uint64_t(d1) << 32) | d0) >> (8u * (rel & 3u));
            lit = imm ? v & (~0ull >> (64u - 8u * ln)) : 0ull;
        } else if (imm) {
            struct __attribute__((packed, aligned(1))) U64 { uint64_t v; };
            lit = reinterpret_cast<const U64*>(a.alt + src)->v & (~0ull >> (64u - 8u * ln));
        }
        const uint64_t m_ps = m_isRef & __ballot(ln <= ROWS_FUSE_LEN);
        const uint64_t m_cA = m_ps & __ballot(src + ln <= src_t(SNV3_MAX_SRC - 1u - ROWS_FUSE_LEN));
        const uint64_t m_cB = m_imm & __ballot(ln == 1u);
        const uint64_t m_ln0 = __ballot(ln == 0u);
        const uint64_t m_c0 = m_ps & ~m_ln0 & __ballot(src >= 1u) & __ballot(src + ln <= src_t(SNV3_MAX_SRC));
        const uint32_t src32 = uint32_t(src);
        const uint32_t src2 = up2(src32), ln2 = up2(ln);
        const uint64_t m_ln20 = __ballot(ln2 == 0u);
        const uint64_t m_c1 = (m_ln20 & m_c0) | (~m_ln20 & m_ps & (m_ln0 | __ballot(src == src_t(src2) + ln2 + 1u)));
        const uint64_t m_gap = m_good & __ballot(sr > pe);
        const uint64_t mRst = ~m_good | firstmask | m_gap;
        const RowsParse p = rows_parse(m_cA, m_cB, m_ps, m_c0, m_c1, mRst, !first, carry_h);
        // a closing lane's fused substitution: run, len1, byte, len2
        const uint32_t f_len1 = lane_bit(p.F & p.real) ? ln2 : 0u;
        const uint32_t f_byte = up1z(uint32_t(lit)) & 0xFFu;
        const uint32_t f_run = f_len1 == 0u ? src32 - 1u : src2;
        uint64_t second = 0;
        uint32_t p_len1 = 0, p_len2 = 0, p_run = 0, p_byte = 0;
        if (MODE == ROWS_DENSE) {
            p_len1 = up2(f_len1); p_len2 = up2(ln); p_run = up2(f_run); p_byte = up2(f_byte);
            const uint64_t m_Lc = p.F & (p.F << 2) & ~(mRst << 1) & __ballot(f_len1 == 0u) & __ballot(p_len1 <= SNV5_MAX_LEN) & __ballot(p_len2 <= SNV5_MAX_LEN) &
                                  __ballot(ln <= SNV5_MAX_LEN) & __ballot(f_run == p_run + p_len1 + 1u + p_len2);
            second = rows_pair(m_Lc, !first, carry_second);
        }
        const uint64_t absorbed = (p.F >> 1) | ((p.F & p.real) >> 2) | (MODE == ROWS_DENSE ? (second >> 2) & p.F : 0ull);
        const bool isSecond = lane_bit(second);
        // ---- what the lane emits: NR runs of result bytes, in order, back to back from offset q0 ----
        // (Round 6, last session: straight-line selects on the lane masks.  Written as nested ifs this block compiled to a dozen exec-mask
        // regions -- s_and_saveexec / s_cbranch_execz / s_or_b64 exec around two or three vector instructions each: 54 vector and 55 scalar
        // instructions per window, the largest block of a kernel that is bound by the issue of both.  Every candidate is computed by every
        // lane now and picked with v_cndmask on a mask that was a scalar value anyway.)
        uint32_t wl[NR], wh[NR], rl[NR];               // descriptor words (the length field of a plain one is filled in below), lengths
#pragma unroll
        for (int i = 0; i < NR; ++i) { wl[i] = 0u; wh[i] = 0u; rl[i] = 0u; }
        uint32_t q0 = pos0 + pe;
        if constexpr (FASTA) {
            if (active) {
                const uint32_t xhl = L.hl[slot & 63u];
                if (xhl) {
                    const uint64_t hs = L.hsrc[slot & 63u];
                    if (isFirst) { rl[0] = xhl; wl[0] = uint32_t(hs); wh[0] = uint32_t(hs >> 32) & 0xFFu; q0 -= xhl; }
                    if (isLast) { const uint64_t lf = hs + xhl - 1u; rl[4] = 1u; wl[4] = uint32_t(lf); wh[4] = uint32_t(lf >> 32) & 0xFFu; }
                }
            }
            if (!in_emit) { rl[0] = 0u; rl[NR - 1] = 0u; }
        }
        // the '.' tail behind a transcript's last item, the '.' gap in front of a task (both: masks inside the active, emitting lanes)
        rl[RT] = lane_bit(lastmask & (m_good | m_empty) & m_in_emit) ? res_len - e : 0u;  wh[RT] = SPACE_FILL << 30;
        wh[RG] = SPACE_FILL << 30;
        if ((m_gap & m_in_emit) != 0ull) rl[RG] = lane_bit(m_gap & m_in_emit) ? sr - pe : 0u;      // (wave-uniform: most windows have no gap)
        // run RS: a complete fused word (the closing lane of a fusion that nothing absorbed), or the task's own plain run
        const uint64_t m_fw = p.F & ~absorbed & m_good;
        const uint64_t m_pl = m_good & ~p.F & ~absorbed & ~m_ln0;
        const bool fusedw = lane_bit(m_fw);            // run RS is a complete fused word
        uint32_t f_rl = f_len1 + 1u + ln, f_back = f_len1 + 1u, f_wl, f_wh;
        if (TILES) {
            // (a piece image: the fused run is ONE contiguous source range -- the first copy, the residue the literal replaces, the
            // copy that goes on one residue later -- with the literal's position and byte beside it: source | has:1 << 29 | position:12 << 8 | byte)
            f_wl = f_run;
            f_wh = (1u << 29) | (f_len1 << 8) | f_byte;
        } else {
            f_wl = (f_run & 0x1FFFFFFFu) | (f_len1 << 29);
            f_wh = ((f_len1 & 0xFFFu) >> 3) | ((ln & 0xFFFu) << 9) | (f_byte << 21) | (7u << 29);
        }
        if (MODE == ROWS_DENSE) {
            const uint64_t w = SNV5_MARK | (uint64_t(f_byte) << 52) | (uint64_t(p_byte) << 44) | (uint64_t(ln & 31u) << 39) | (uint64_t(p_len2 & 31u) << 34) | (uint64_t(p_len1 & 31u) << 29) | (uint64_t(p_run) & SNV3_MAX_SRC);
            f_rl = isSecond ? p_len1 + 1u + p_len2 + 1u + ln : f_rl;
            f_back = isSecond ? p_len1 + p_len2 + 2u : f_back;
            f_wl = isSecond ? uint32_t(w) : f_wl;
            f_wh = isSecond ? uint32_t(w >> 32) : f_wh;
        }
        rl[RS] = lane_bit(m_fw & m_in_emit) ? f_rl : 0u;
        wl[RS] = f_wl; wh[RS] = f_wh;
        if ((m_pl & m_in_emit) != 0ull) {                                                            // (wave-uniform: streams of pure substitutions have no plain run)
            const uint32_t p_wl = imm ? uint32_t(lit) : src32;
            const uint32_t p_wh = imm ? uint32_t(lit >> 32) | (SPACE_IMM << 30)
                                      : (SRC32 ? 0u : uint32_t(uint64_t(src) >> 32) & 0xFFu) | ((isRef ? SPACE_PROTEOME : SPACE_PAYLOAD) << 30);
            rl[RS] = lane_bit(m_pl & m_in_emit) ? ln : rl[RS];
            wl[RS] = fusedw ? f_wl : p_wl;
            wh[RS] = fusedw ? f_wh : p_wh;
        }
        q0 -= fusedw ? f_back : 0u;
        if constexpr (TILES) {
            // ---- a TILE image: every run leaves as PIECES -- <= 16 result bytes of one source, their offset inside the tile's result and at
            // most one substituted residue (dense_pieces.h) -- into the tile's slots; the tile is the executor's work item as it stands
            uint32_t cn[NR], cnt = 0;
#pragma unroll
            for (int i = 0; i < NR; ++i) { cn[i] = (rl[i] + 15u) >> 4; cnt += cn[i]; }
            const uint32_t incl = wave_incl_scan(cnt);
            const uint32_t round_total = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
            if (PHASE != PH_COUNT && round_total != 0u) {
                uint32_t k = tile_cnt + incl - cnt;
                uint32_t pos = q0;                                                       // offset from the tile's first result byte (< 16 384: checked above)
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const uint32_t L = rl[i];
                    if (L != 0u) {
                        const uint32_t space = wh[i] >> 30;
                        const bool haslit = i == RS && space == SPACE_PROTEOME && ((wh[i] >> 29) & 1u) != 0u;
                        const uint32_t lp = (wh[i] >> 8) & 0xFFFu, lb = wh[i] & 0xFFu;
                        for (uint32_t at = 0; at < L; at += 16u) {
                            const uint32_t len = L - at < 16u ? L - at : 16u;
                            // (the piece as two 32-bit words: low = source : 31 | space bit 0; high = space bit 1 | offset : 14 << 1 | bytes - 1 : 4 << 15 |
                            // residue position : 4 << 19 | has one << 23 | its byte << 24 -- dense_pieces.h's layout without 64-bit shifts)
                            uint32_t w_lo, w_hi = (space >> 1) | ((pos + at) << 1) | ((len - 1u) << 15);
                            if (space == SPACE_IMM) {                                    // (<= 5 bytes: one piece; bits 0..30 and 51..63 hold them)
                                const uint32_t v_hi9 = (wl[i] >> 31) | ((wh[i] & 0xFFu) << 1);       // bits 31 .. 39 of the literal
                                w_lo = wl[i] | 0x80000000u;
                                w_hi |= v_hi9 << 19;
                            } else {
                                const bool has = haslit && lp - at < 16u;                // (unsigned: at <= lp < at + 16)
                                const uint32_t srcw = space == SPACE_FILL ? 0u : wl[i] + at;
                                w_lo = (srcw & 0x7FFFFFFFu) | (space << 31);
                                w_hi |= has ? (((lp - at) & 15u) << 19) | (1u << 23) | (lb << 24) : 0u;
                            }
                            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                            if (k < out_cap) *reinterpret_cast<u32x2*>(out + k) = u32x2{w_lo, w_hi};
                            ++k;
                        }
                        pos += L;
                    }
                }
            }
            tile_cnt += round_total;
            if (last) break;
            carry_h = p.h >> ADV; carry_second = second >> ADV;
            carry_e = uint32_t(__builtin_amdgcn_readlane(int(e), int(ADV - 1u)));
            first = false;
            continue;
        }
        // Everything rare behind ONE uniform branch: a run of more than 1 KiB (it may cross two rows of the arena, or exceed a
        // descriptor's length field and become several descriptors)
        uint64_t m_big = 0ull;
#pragma unroll
        for (int i = 0; i < NR; ++i) m_big |= __ballot(rl[i] > ROW_BYTES);
        const bool slow = m_big != 0ull;
        // ... and the usual window before the general one: no lane has a gap, a tail or FASTA text to emit -- every lane at most its
        // Task's own run -- so a descriptor's slot is a ballot and a popcount (no scan), and a run holds one row boundary at most
        uint64_t m_other = __ballot(rl[RG] != 0u) | __ballot(rl[RT] != 0u);
        if (FASTA) m_other |= __ballot(rl[0] != 0u) | __ballot(rl[NR - 1] != 0u);
        const bool plain_window = !slow && m_other == 0ull;
        if (plain_window) {
            const bool has = rl[RS] != 0u;
            const uint64_t hm = __ballot(has);
            const uint32_t round_total = uint32_t(__popcll(hm));
            if (PHASE != PH_COUNT && round_total != 0u) {
                const uint32_t k = tile_cnt + mbcnt(hm);
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                if (has && k < out_cap) *reinterpret_cast<u32x2*>(reinterpret_cast<char*>(out) + (k << 3)) = u32x2{wl[RS], fusedw ? wh[RS] : (wh[RS] | (rl[RS] << 8))};
                const uint32_t s0 = eoff + q0, rb = ((s0 + ROW_BYTES - 1u) >> 10) << 10;
                if (has && rb < s0 + rl[RS] && (rb >> 10) - row_lo < row_span)
                    *reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(cover_tile) + (rb >> 7)) = cover_word(k, rb - s0, s0 + rl[RS] - rb);       // (rb >> 10 rows of 8 bytes)
            }
            tile_cnt += round_total;
        } else if (!slow) {
            uint32_t cn[NR], cnt = 0;
#pragma unroll
            for (int i = 0; i < NR; ++i) { cn[i] = rl[i] ? 1u : 0u; cnt += cn[i]; }
            const uint32_t incl = wave_incl_scan(cnt);
            const uint32_t round_total = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
            if (PHASE != PH_COUNT && round_total != 0u) {
                const uint32_t k = tile_cnt + incl - cnt;                                 // the lane's first slot inside the tile
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                uint32_t kk = k;
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    if (rl[i] && kk < out_cap) *reinterpret_cast<u32x2*>(out + kk) = u32x2{wl[i], (i == RS && fusedw) ? wh[i] : (wh[i] | (rl[i] << 8))};
                    kk += cn[i];
                }
                // the rows whose first byte the lane's runs cover: every run is at most 1 KiB here, so a run holds at most one boundary
                const uint32_t s0 = eoff + q0;
                uint32_t end[NR];
                {
                    uint32_t t = s0;
#pragma unroll
                    for (int i = 0; i < NR; ++i) { t += rl[i]; end[i] = t; }
                }
                const uint32_t rfirst = (s0 + ROW_BYTES - 1u) >> 10;                      // first boundary at or behind the span's start
                if ((rfirst << 10) < end[NR - 1] && cnt != 0u) {
                    for (uint32_t r = rfirst; (r << 10) < end[NR - 1]; ++r) {              // (one round, rarely more)
                        const uint32_t rb = r << 10;
                        uint32_t dk = k, st = s0, en = end[0];
#pragma unroll
                        for (int i = 0; i + 1 < NR; ++i) if (rb >= end[i]) { dk += cn[i]; st = end[i]; en = end[i + 1]; }
                        if (r - row_lo < row_span) *reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(cover_tile) + (r << 3)) = cover_word(dk, rb - st, en - rb);
                    }
                }
            }
            tile_cnt += round_total;
        } else {
            // the general form: any length, any number of pieces and rows
            auto pieces = [](uint32_t l) -> uint32_t { return l == 0u ? 0u : (l + PIECE_MAX - 1u) / PIECE_MAX; };
            uint32_t cnt = 0;
#pragma unroll
            for (int i = 0; i < NR; ++i) cnt += (i == RS && fusedw) ? (rl[i] ? 1u : 0u) : pieces(rl[i]);
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
                            if (row >= 1u && row < a.n_rows) a.cover[row] = cover_word(k, (r << 10) - pos, pos + piece - (r << 10));
                        }
                        if (space == SPACE_PROTEOME || space == SPACE_PAYLOAD) sfield += piece;      // (an immediate is never cut: <= 5 bytes)
                        len -= piece; pos += piece; ++k;
                        if (whole) break;
                    }
                };
#pragma unroll
                for (int i = 0; i < NR; ++i) if (rl[i]) put((uint64_t(wh[i]) << 32) | wl[i], rl[i], i == RS && fusedw);
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
        if (PHASE == PH_PAD && (tile_cnt > pad_slots || tile_cnt > 0xFFFFu)) rreport(a.status, tile, STATUS_ROWS_STAGE);
    }
}

// descriptors out of the padded array into their final, dense place (tile_desc_base: the scan of the tiles' counts): one wave per tile
__global__ __launch_bounds__(256) void rows_compact_kernel(RowsArgs a)
{
    const uint64_t rel = uint64_t(blockIdx.x) * 4u + (threadIdx.x >> 6), tile = a.tile0 + rel;
    if (tile >= a.tile1) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t b0 = a.tile_desc_base[tile], n = a.tile_desc_base[tile + 1u] - b0;
    if (n > ROWS_PAD || b0 + n > a.desc_cap) return;                  // (reported by the parse / by the host)
    const uint64_t* src = a.desc_pad + rel * ROWS_PAD;
    for (uint32_t k = lane; k < n; k += 64u) a.desc[b0 + k] = src[k];
}

__global__ __launch_bounds__(256) void rows_hap_begin_kernel(RowsArgs a)
{
    const uint64_t h = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (h > a.n_haps) return;
    const uint64_t t = h < a.n_haps ? a.hap_tx_begin[h] : a.n_tx;
    const uint64_t tile = t / a.K;
    uint64_t b = a.tile_res_base[tile < a.n_tiles ? tile : a.n_tiles];
    if (tile < a.n_tiles)
        for (uint64_t u = tile * a.K; u < t; ++u) { const uint32_t hl = a.tx_header_len ? a.tx_header_len[u] : 0u; b += uint64_t(a.tx_res_len[u]) + (hl ? hl + 1u : 0u); }
    a.hap_out_begin[h] = b;
}

// ---- the cutter: one wave per segment of ROWS_SEG rows ----------------------------------------------------------------------------
// PASS 0: count the chunks of every segment (seg_count);  1: emit them at seg_base[seg] (after the scan of the counts);
//      2: both at once -- counted, and emitted to the segment's chunk_pad slots (rows_chunk_pad_for) of a padded table that a copy kernel compacts
//         after the scan (a segment with more chunks than slots raises totals[3]: the host then runs pass 1 instead of the copy)
template <int PASS, uint32_t MAX_ROWS>
__global__ __launch_bounds__(64) void rows_cut_kernel(RowsArgs a, uint32_t max_desc, uint64_t flag)
{
    constexpr uint32_t max_rows = MAX_ROWS;                 // (a compile-time count: the candidates' loop below is ten / twelve DPP shifts in a row)
    constexpr bool EMIT = PASS != 0;
    const uint32_t lane = threadIdx.x;
    const uint64_t n_desc = a.tile_desc_base[a.n_tiles];
    const uint64_t seg = a.seg0 + blockIdx.x;
    if (*a.status != STATUS_CLEAN) {                                                      // the parse failed (or asks for its two-phase form): the row map is not to be walked
        if (PASS != 1 && lane == 0) a.seg_count[seg] = 0u;
        return;
    }
    const uint64_t s0 = seg * ROWS_SEG, s1 = s0 + ROWS_SEG < a.n_rows ? s0 + ROWS_SEG : a.n_rows;
    uint32_t count = 0;
    uint64_t out_k = PASS == 1 ? a.seg_base[seg] : (PASS == 2 ? seg * a.chunk_pad : 0);
    Chunk* const table = PASS == 2 ? a.chunks_pad : a.chunks_tmp;
    uint64_t last_dst = 0;
    uint64_t r0 = s0;
    while (r0 < s1) {
        // rows r0 .. r0 + 63 in registers: the descriptor covering each row's first byte
        const uint64_t b = r0, r = b + lane;
        uint64_t c = 0, idx = 0;
        uint32_t pslot = 0, pd1 = 0;             // PADDED records: the row's descriptor as a slot of the padded array; descriptors from it to the end of its tile
        if (r < a.n_rows) {
            if (r >= 1u) c = a.cover[r];
            if (c >> 63) idx = (c >> 22) & ((1ull << 41) - 1ull);
            else {
                const uint64_t tl = c >> 38, sl = (c >> 22) & 0xFFFFu, tb0 = a.tile_desc_base[tl];
                idx = r >= 1u ? tb0 + sl : 0ull;
                if (a.pad_chunks) {
                    const uint64_t d1 = a.tile_desc_base[tl + 1u] - idx;             // (idx < that unless tile 0 is empty and r = 0: d1 = 0 marks it)
                    pslot = uint32_t(tl * ROWS_PAD + sl);
                    pd1 = uint32_t(d1 < CHUNK_N1_MASK ? d1 : CHUNK_N1_MASK);
                }
            }
        }
        const uint32_t off = uint32_t(c) & PIECE_MAX;
        const uint64_t lastd = r >= a.n_rows ? n_desc - 1u : (off ? idx : idx - 1u);      // last descriptor of a chunk that ends at row r
        // what a chunk ending at row r leaves of its last descriptor behind the cut: the parse wrote it into the row map
        const uint32_t tc = (EMIT && r < a.n_rows && off != 0u) ? (uint32_t(c) >> 11) & PIECE_MAX : 0u;
        // Round 6: the greedy walk in two parts.  (1) EVERY lane as a chunk's first row at once: how many rows does a chunk that starts here
        // take -- the rows behind it while it stays within max_rows, the segment, and max_desc descriptors (the last descriptor of a chunk
        // ending at row e is monotone in e: the admissible ends are a prefix) -- one DPP shift, a compare and an add per candidate row, for
        // all 64 starts together.  (2) The chain 0 -> take[0] -> ... is then one v_readlane per chunk, and the chunks' records are written
        // by their first rows' lanes in parallel (what they need of their END row comes through ds_bpermute).  The serial form spent ~45 wave
        // instructions per chunk with one lane emitting: 0.43 ms of C3 whole's build; this form 0.27 (C2 0.17 -> 0.12, C4 whole 0.34 -> 0.21).
        // (The segment's whole row map requested up front and kept in LDS -- two waits a wave instead of 22 -- was measured with BOTH walks and
        // lost both times: 0.56 against 0.43 with the serial walk, 0.31 against 0.27 with this one; its registers and LDS halve the occupancy.)
        const uint32_t f_lo = uint32_t(idx), lastd_lo = uint32_t(lastd);
        uint32_t take = 0;
        {
            uint32_t x = lastd_lo;
            bool open = true;
#pragma unroll
            for (uint32_t k = 1; k <= max_rows; ++k) {
                x = from_next_lane(0u, x);                                                // the last descriptor of a chunk ending at row r + k
                open = open && r + k <= s1 && lane + k <= 63u && x - f_lo + 1u <= max_desc;       // (mod 2^32: the difference is a chunk's descriptor count)
                take += open ? 1u : 0u;
            }
        }
        uint64_t starts = 0;
        uint32_t cur = 0, n_new = 0;
        bool refused = false;
        for (;;) {
            const uint32_t t = uint32_t(__builtin_amdgcn_readlane(int(take), int(cur)));
            if (t == 0u) { refused = true; break; }
            starts |= 1ull << cur;
            ++n_new;
            cur += t;
            if (b + cur >= s1) break;
            if (cur + max_rows > 63u) break;                                              // the next chunk's candidates leave the registers: reload from its first row
        }
        if (refused) {                                                                    // a row with more descriptors than a chunk may hold
            if (lane == cur) rreport(a.status, idx, STATUS_ROWS_TOO_MANY);
            if (lane == 0 && PASS != 1) a.seg_count[seg] = 0u;                            // (the count is WRITTEN: the scan and the table pass behind this kernel run before the host has looked at the status)
            return;
        }
        if (EMIT) {
            const uint32_t e_lane = lane + take < 64u ? lane + take : 63u;
            const uint32_t end_lastd = uint32_t(__shfl(int(lastd_lo), int(e_lane))), end_tc = uint32_t(__shfl(int(tc), int(e_lane)));
            const bool mine = ((starts >> lane) & 1ull) != 0ull;
            const uint32_t rank = mbcnt(starts);
            if (mine && (PASS != 2 || count + rank < a.chunk_pad)) {
                const uint64_t n = uint64_t(end_lastd - f_lo + 1u);
                // (a.pad_chunks: the record addresses the PADDED array, sir_pack.hpp -- its first descriptor as a slot, and for now how many
                // descriptors its tile holds from there on in the bits that will say n1: rows_chunk_compact_kernel, lane = chunk, turns that
                // into n1 and checks that the chunk ends inside the next tile.)
                const uint64_t first = a.pad_chunks ? uint64_t(pslot) : idx, low = a.pad_chunks ? uint64_t(pd1) : 0ull;
                table[out_k + rank] = Chunk{first | (uint64_t(off) << TB_IDX_BITS) | (uint64_t(end_tc) << (TB_IDX_BITS + TB_SKIP_BITS)), (r * ROW_BYTES) | low | (n << 48) | CHUNK_CLIP | flag};
            }
        }
        last_dst = (b + (63u - uint32_t(__builtin_clzll(starts)))) * ROW_BYTES;
        count += n_new; out_k += n_new;
        r0 = b + cur;
    }
    if (PASS != 1 && lane == 0) {
        a.seg_count[seg] = count;
        if (seg + 1u == a.n_segs) a.totals[2] = last_dst;
        if (PASS == 2 && count > a.chunk_pad) atomicOr(reinterpret_cast<unsigned long long*>(a.totals) + 3, 1ull);
    }
}

__device__ __forceinline__ void rows_chunk_key(const RowsArgs& a, const Chunk& ch, uint64_t k, uint64_t n_desc);

// the padded chunk table of pass 2 -> arena order (seg_base: the scan of the segments' counts): one wave per segment
__global__ __launch_bounds__(256) void rows_chunk_compact_kernel(RowsArgs a)
{
    const uint64_t seg = a.seg0 + uint64_t(blockIdx.x) * 4u + (threadIdx.x >> 6);
    if (seg >= a.seg1) return;
    // (the one call launches this pass before the host has looked at the status word: behind a parse or a cut that was refused the padded
    // table holds whatever the memory held before -- nothing of it is followed)
    if (*a.status != STATUS_CLEAN) return;
    // (... and behind a cut that found a segment with more chunks than its chunk_pad slots -- totals[3] bit 0, raised by the cutter
    // that ran before this kernel on the same stream -- seg_base is the scan of the UNCLAMPED counts: the segments behind the full one
    // would start past the table the caller sized for chunk_pad records per segment.  Nothing is written; the host sees the
    // bit and runs the cutter's emitting pass (or, in the one call, builds in one piece).  chunk_cap bounds every store besides.)
    if (a.totals[3] & 1ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t b0 = a.seg_base[seg], n = a.seg_base[seg + 1u] - b0;
    for (uint32_t k = lane; k < n && k < a.chunk_pad && b0 + k < a.chunk_cap; k += 64u) {
        Chunk ch = a.chunks_pad[seg * a.chunk_pad + k];
        if (a.pad_chunks) {
            // slot-addressed records (sir_pack.hpp): d1 -- the descriptors the first tile holds from the chunk's first one on -- becomes n1
            // (0: the chunk ends inside that tile); a chunk that would go on into a THIRD tile (tiles of a few descriptors) cannot be
            // written down: the host is told (totals[3] bit 1) and builds the dense form instead
            const uint64_t d1 = ch.dst_n & CHUNK_N1_MASK, nd = (ch.dst_n >> 48) & CHUNK_N_MASK;
            const uint64_t tl = (ch.task_begin & TB_IDX_MASK) / ROWS_PAD;
            const uint64_t next = tl + 2u <= a.n_tiles ? a.tile_desc_base[tl + 2u] - a.tile_desc_base[tl + 1u] : 0ull;      // descriptors of the next tile
            if (d1 == 0u || nd > d1 + next) atomicOr(reinterpret_cast<unsigned long long*>(a.totals) + 3, 2ull);
            ch.dst_n = (ch.dst_n & ~CHUNK_N1_MASK) | (nd <= d1 ? 0ull : d1);
            // ... and, the descriptors of a padded image being where they stay, the chunk's keys for the XCD / window order right here
            // (a.bucket: set by the caller that wants them; the dense form computes them in rows_keys_kernel, behind the compaction)
            if (a.bucket) rows_chunk_key(a, ch, b0 + k, a.desc_cap);
        }
        a.chunks_tmp[b0 + k] = ch;
    }
}

// proteome slice and window of a chunk (order_chunks_for_xcds: the first reference read among its first six descriptors)
__device__ __forceinline__ void rows_chunk_key(const RowsArgs& a, const Chunk& ch, uint64_t k, uint64_t n_desc)
{
    const uint64_t tb = ch.task_begin & TB_IDX_MASK;
    const uint32_t n = uint32_t(ch.dst_n >> 48) & CHUNK_N_MASK, n1 = a.pad_chunks ? uint32_t(ch.dst_n & CHUNK_N1_MASK) : 0u;
    uint64_t key = 0;
    for (uint32_t q = 0; q < n && q < 6u; ++q) {
        const uint64_t i = chunk_desc_slot(tb, n1, q);                   // (n_desc: the array's slots when the records address the padded array)
        if (i >= n_desc) break;
        const uint64_t d = a.desc[i];
        const bool snv = (d & SNV3_MARK) == SNV3_MARK || (d >> 60) == 0xDull;
        const uint64_t src = snv ? (d & SNV3_MAX_SRC) : (d & SRC_MASK);
        if ((snv || (d >> 62) == SPACE_PROTEOME) && src < a.proteome_len) { key = src; break; }
    }
    const uint64_t per = (a.proteome_len + 7) / 8;
    const uint64_t bk = per ? key / per : 0;
    const uint8_t bucket = uint8_t(bk < 8 ? bk : 7);
    a.bucket[k] = bucket;
    a.sub[k] = a.hap_major ? uint8_t(0) : xcd_sub_window(key, bucket, per);      // (haplotype-major inside a slice: every window key equal, the stable sort keeps the arena's order)
}
__global__ __launch_bounds__(256) void rows_keys_kernel(RowsArgs a, uint64_t n_chunks, uint64_t n_desc)
{
    const uint64_t k = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (k >= n_chunks) return;
    rows_chunk_key(a, a.chunks_tmp[k], k, n_desc);
}

}  // namespace

uint64_t rows_scan_scratch_entries(uint64_t n) { return (n + RS_TILE - 1) / RS_TILE + 2; }

hipError_t launch_rows_tile_bytes(const RowsArgs& a, uint64_t* scan_scratch, hipStream_t stream)
{
    if (a.K >= 32u) hipLaunchKernelGGL(rows_tile_bytes_wave_kernel, dim3(uint32_t((a.n_tiles + 3) / 4)), dim3(256), 0, stream, a);
    else {
        hipError_t e = hipMemsetAsync(a.tile_bytes, 0, a.n_tiles * 8, stream);
        if (e != hipSuccess) return e;
        if (a.n_tx) hipLaunchKernelGGL(rows_tile_bytes_kernel, dim3(uint32_t((a.n_tx + 255) / 256)), dim3(256), 0, stream, a);
    }
    const uint64_t n = a.n_tiles, n_t = (n + RS_TILE - 1) / RS_TILE;
    hipLaunchKernelGGL(rows_scan_sums, dim3(uint32_t(n_t)), dim3(256), 0, stream, a.tile_bytes, n, scan_scratch);
    hipLaunchKernelGGL(rows_scan_tiles, dim3(1), dim3(1024), 0, stream, scan_scratch, n_t);
    hipLaunchKernelGGL(rows_scan_apply, dim3(uint32_t(n_t)), dim3(256), 0, stream, a.tile_bytes, n, scan_scratch, a.tile_res_base);
    return hipGetLastError();
}

static_assert(ROWS_PAD == ROWS_PAD_SLOTS && ROWS_PAD == ROWS_TILE_SLOTS, "build_rows.h, sir_pack.hpp");

// every source offset a Task of this launch can name fits 32 bits, with room for the longest run behind it (the header table sits behind the proteome)
static bool rows_src32(const RowsArgs& a) { return a.proteome_len + a.headers_len + (1ull << 16) < (1ull << 32) && a.n_alt + (1ull << 16) < (1ull << 32); }

template <int MODE, bool FASTA>
static hipError_t launch_parse_t(const RowsArgs& a, int phase, hipStream_t stream)
{
    const dim3 grid{uint32_t(a.tile1 - a.tile0)};
    // (the one-pass form -- what every cohort that fits its tiles runs -- has a 32-bit instance; the two-pass form keeps 64-bit sources)
    if (phase == PH_PAD && rows_src32(a)) hipLaunchKernelGGL((rows_parse_kernel<MODE, FASTA, PH_PAD, true>), grid, dim3(64), 0, stream, a);
    else if (phase == PH_PAD) hipLaunchKernelGGL((rows_parse_kernel<MODE, FASTA, PH_PAD, false>), grid, dim3(64), 0, stream, a);
    else if (phase == PH_COUNT) hipLaunchKernelGGL((rows_parse_kernel<MODE, FASTA, PH_COUNT, false>), grid, dim3(64), 0, stream, a);
    else hipLaunchKernelGGL((rows_parse_kernel<MODE, FASTA, PH_DIRECT, false>), grid, dim3(64), 0, stream, a);
    return hipGetLastError();
}

// (the launchers resolve the range: tile1 / seg1 == 0 mean "to the end")
static RowsArgs ranged(const RowsArgs& a0)
{
    RowsArgs a = a0;
    rows_ranges(a0, a.tile0, a.tile1, a.seg0, a.seg1);
    return a;
}

hipError_t launch_rows_parse(const RowsArgs& a0, int mode, bool fasta, int phase, hipStream_t stream)
{
    const RowsArgs a = ranged(a0);
    if (a.tile1 <= a.tile0) return hipSuccess;
    if (a.tile1 - a.tile0 > 0x7FFFFFFFull) return hipErrorInvalidValue;
    if (mode == ROWS_DENSE) return fasta ? launch_parse_t<ROWS_DENSE, true>(a, phase, stream) : launch_parse_t<ROWS_DENSE, false>(a, phase, stream);
    if (mode == ROWS_TILES) {
        if (phase != PH_PAD || a.tile_slots == 0u || a.tile_slots > 2048u || a.tile_span_max > 16368u || !rows_src32(a)) return hipErrorInvalidValue;   // (a piece's source field: 31 bits)
        const dim3 grid{uint32_t(a.tile1 - a.tile0)};
        if (fasta) hipLaunchKernelGGL((rows_parse_kernel<ROWS_TILES, true, PH_PAD, true>), grid, dim3(64), 0, stream, a);
        else hipLaunchKernelGGL((rows_parse_kernel<ROWS_TILES, false, PH_PAD, true>), grid, dim3(64), 0, stream, a);
        return hipGetLastError();
    }
    return fasta ? launch_parse_t<ROWS_WAVE, true>(a, phase, stream) : launch_parse_t<ROWS_WAVE, false>(a, phase, stream);
}

hipError_t launch_rows_compact(const RowsArgs& a0, hipStream_t stream)
{
    const RowsArgs a = ranged(a0);
    if (a.tile1 <= a.tile0) return hipSuccess;
    hipLaunchKernelGGL(rows_compact_kernel, dim3(uint32_t((a.tile1 - a.tile0 + 3) / 4)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_rows_hap_begin(const RowsArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(rows_hap_begin_kernel, dim3(uint32_t((a.n_haps + 1 + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_rows_cut(const RowsArgs& a0, int mode, int pass, hipStream_t stream)
{
    const RowsArgs a = ranged(a0);
    if (a.seg1 <= a.seg0) return hipSuccess;
    const uint32_t n_launch = uint32_t(a.seg1 - a.seg0);
    const uint32_t max_desc = mode == ROWS_DENSE ? CHUNK_TASKS_DEEP : CHUNK_TASKS_WAVE;
    const uint64_t flag = mode == ROWS_DENSE ? CHUNK_DENSE : CHUNK_WAVE;
    static_assert(ROWS_MAX_WAVE == 10 && ROWS_MAX_DENSE == 12, "the cutter's instances");
    if (mode == ROWS_DENSE) {
        if (pass == 1) hipLaunchKernelGGL((rows_cut_kernel<1, ROWS_MAX_DENSE>), dim3(n_launch), dim3(64), 0, stream, a, max_desc, flag);
        else if (pass == 2) hipLaunchKernelGGL((rows_cut_kernel<2, ROWS_MAX_DENSE>), dim3(n_launch), dim3(64), 0, stream, a, max_desc, flag);
        else hipLaunchKernelGGL((rows_cut_kernel<0, ROWS_MAX_DENSE>), dim3(n_launch), dim3(64), 0, stream, a, max_desc, flag);
    } else {
        if (pass == 1) hipLaunchKernelGGL((rows_cut_kernel<1, ROWS_MAX_WAVE>), dim3(n_launch), dim3(64), 0, stream, a, max_desc, flag);
        else if (pass == 2) hipLaunchKernelGGL((rows_cut_kernel<2, ROWS_MAX_WAVE>), dim3(n_launch), dim3(64), 0, stream, a, max_desc, flag);
        else hipLaunchKernelGGL((rows_cut_kernel<0, ROWS_MAX_WAVE>), dim3(n_launch), dim3(64), 0, stream, a, max_desc, flag);
    }
    return hipGetLastError();
}

hipError_t launch_rows_chunk_compact(const RowsArgs& a0, hipStream_t stream)
{
    const RowsArgs a = ranged(a0);
    if (a.seg1 <= a.seg0) return hipSuccess;
    hipLaunchKernelGGL(rows_chunk_compact_kernel, dim3(uint32_t((a.seg1 - a.seg0 + 3) / 4)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// what the host looks at once per slice, gathered into the totals block: [4] descriptors up to the slice's last tile, [5] chunks up to
// its last segment, [6] the status word -- one 64-byte copy instead of four round trips (the GPU waits for the host there)
__global__ void rows_summary_kernel(const uint64_t* desc_end, const uint64_t* chunk_end, const unsigned long long* status, uint64_t* totals)
{
    totals[4] = desc_end ? *desc_end : 0ull;
    totals[5] = chunk_end ? *chunk_end : 0ull;
    totals[6] = *status;
}
hipError_t launch_rows_summary(const uint64_t* desc_end, const uint64_t* chunk_end, const unsigned long long* status, uint64_t* totals, hipStream_t stream)
{
    hipLaunchKernelGGL(rows_summary_kernel, dim3(1), dim3(1), 0, stream, desc_end, chunk_end, status, totals);
    return hipGetLastError();
}

hipError_t launch_rows_keys(const RowsArgs& a, uint64_t n_chunks, uint64_t n_desc, hipStream_t stream)
{
    if (n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(rows_keys_kernel, dim3(uint32_t((n_chunks + 255) / 256)), dim3(256), 0, stream, a, n_chunks, n_desc);
    return hipGetLastError();
}

// the records of a PADDED image (sir_pack.hpp) as records of the dense one: slot -> tile_desc_base[tile] + slot inside the tile
__global__ __launch_bounds__(256) void rows_chunks_dense_kernel(const Chunk* in, uint64_t n, const uint64_t* __restrict__ tile_desc_base, Chunk* out)      // (in == out: in place)
{
    const uint64_t k = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (k >= n) return;
    const Chunk ch = in[k];
    const uint64_t slot = ch.task_begin & TB_IDX_MASK;
    const uint64_t dense = tile_desc_base[slot / ROWS_TILE_SLOTS] + slot % ROWS_TILE_SLOTS;
    out[k] = Chunk{(ch.task_begin & ~TB_IDX_MASK) | dense, ch.dst_n & ~CHUNK_N1_MASK};
}
hipError_t launch_rows_chunks_dense(const Chunk* in, uint64_t n, const uint64_t* tile_desc_base, Chunk* out, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(rows_chunks_dense_kernel, dim3(uint32_t((n + 255) / 256)), dim3(256), 0, stream, in, n, tile_desc_base, out);
    return hipGetLastError();
}

__global__ void code_object_loader_d() {}
hipError_t preload_build_rows(hipStream_t stream)
{
    hipLaunchKernelGGL(code_object_loader_d, dim3(1), dim3(64), 0, stream);
    return hipGetLastError();
}

}  // namespace v2p
