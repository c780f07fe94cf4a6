// stitch_wave.hip -- stitchw_kernel: the SIR executor with ONE WAVE PER CHUNK (gfx950, wave64; no MFMA: byte/index work).
//
// Same contract as the other stitch kernels (task.rs:38-50 for a whole batch of haplotypes per launch, descriptors and chunk
// table of sir_pack.hpp), built from what the round-2 measurements say limits them on MI355X:
//   * a workgroup of four waves pays five barriers and a cross-wave combine for every scan, and its wave slots stay idle
//     until its slowest wave has stored its last row -- a third of the long-run kernel's time was per-workgroup cost;
//   * the per-block kernel is VALU-bound because every lane runs the multi-source merge for every block;
//   * a fused substitution was expanded back into three tasks, which tripled every table the set-up builds.
// Here a chunk is at most 64 descriptors and 10 KiB of result and belongs to ONE wave: lane = descriptor in the set-up, lane =
// 16-byte block in the copy.  Nothing crosses a wave, so there is no s_barrier at all (LDS operations of one wave execute in
// order), scans are one DPP pass with the total read by v_readlane, and a wave slot is free the moment its own ten rows are
// stored.  A descriptor is ONE record -- a fused substitution is a reference run with one residue replaced, not three tasks:
//   A  descriptor -> record {source address - start, start, end, substituted position + byte}; DPP scan of the lengths;
//      +1 scattered into a byte-per-block map at the first block starting inside or after each record
//   C  in-lane SWAR prefix + wave scan of the map: map[k] = record covering the first byte of block k (8 blocks per lane)
//   P  lane = record: a record that starts inside a block fetches its own stream at that block (ONE gather per lane); the first
//      record to start inside a block overlays the pieces of the records that follow it there and parks the block's tail in LDS
//   K  lane = block, ten 1 KiB rows per wave: one map byte + one record per block, then one byte-granular dwordx4 gather of
//      the covering record's stream for EVERY block; at store time a block its record ends in takes the parked tail from the
//      record's end on (mask from a 17-entry LDS table), a replaced residue inside the block is placed.  All ten rows are
//      gathered before the first store (gfx950 counts loads and stores in one in-order counter), then leave as aligned
//      non-temporal dwordx4 buffer stores, 1 KiB per instruction.
// An immediate descriptor's literal bytes are read as a stream too: out of the descriptor array itself (their record's source
// address is the descriptor's own address), which is why the array needs 16 readable bytes before and 32 behind it.
// Twelve vector-memory instructions per 10 KiB on the read side (descriptors, patch gather, ten row gathers), ten on the write
// side: the kernel is bound by the CU's vector-memory pipeline and the latency chain of a wave, not by its ~450 VALU instructions.
// A descriptor that would read out of bounds is reported in the device status word and its chunk is not executed; nothing is
// ever read or written outside the buffers (sources carry PAD_BYTES of readable slack, as for the other kernels).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "stitch_kernels.h"
#include "stitch_device.hpp"

namespace v2p {

struct __attribute__((aligned(16))) WRec {
    // positions are in BLOCK SPACE: (chunk's result offset & 15) + offset inside the chunk, so block b covers [16b, 16b + 16)
    uint32_t a_lo, a_hi;   // source address minus the record's start position (an immediate record: the address of its own descriptor, whose low bytes are the literal)
    uint32_t se;           // start | end << 16
    uint32_t lit;          // position of the substituted residue (WREC_NOLIT: none) | its byte << 16
};
constexpr uint32_t WREC_NOLIT = 0xFFFFu;

__device__ __forceinline__ uint64_t wrec_adj(const WRec& t) { return (uint64_t(t.a_hi) << 32) | t.a_lo; }

// byte q (0..15) of v replaced by `byte`
__device__ __forceinline__ u32x4 put_byte(u32x4 v, uint32_t q, uint32_t byte)
{
    const uint32_t sh = 8u * (q & 3u), m = 0xFFu << sh, bv = byte << sh, k = q >> 2;
    v[0] = k == 0u ? (v[0] & ~m) | bv : v[0];
    v[1] = k == 1u ? (v[1] & ~m) | bv : v[1];
    v[2] = k == 2u ? (v[2] & ~m) | bv : v[2];
    v[3] = k == 3u ? (v[3] & ~m) | bv : v[3];
    return v;
}

// Development builds only (tools/build_variant.sh -DV2P_WAVE_CHECK): every gather is checked against the readable range of the
// three source buffers; an address outside them is reported (reason 100 + site) and not followed.  The product build has no such code.
struct WChk {
#ifdef V2P_WAVE_CHECK
    uint64_t lo[4], hi[4];
    unsigned long long* status;
    uint64_t tb;
#endif
};
// (V2P_WAVE_ABLATE, development builds only, results are wrong: bit 0 the copy phase gathers the dots, bit 1 the patch phase does,
// bit 2 no result store)
#if !defined(V2P_WAVE_ABLATE) || !defined(V2P_BENCH_VARIANTS)
#undef V2P_WAVE_ABLATE
#define V2P_WAVE_ABLATE 0                    /* (the engine library never ablates) */
#endif
__device__ __forceinline__ u32x4 wgather(uint64_t addr, const WChk& k, uint32_t site)
{
#ifdef V2P_WAVE_CHECK
    bool ok = false;
    for (int q = 0; q < 4; ++q) ok = ok || (addr >= k.lo[q] && addr + 16u <= k.hi[q]);
    if (!ok) { report(k.status, (k.tb << 8) | (threadIdx.x & 63u), 100u + site); return u32x4{0u, 0u, 0u, 0u}; }
#endif
    (void)k; (void)site;
    return gather16(addr);
}

// byte q of v replaced by the residue of `lit` when it lies in the block at b16 (no-op otherwise: q >= 16 matches no dword)
__device__ __forceinline__ u32x4 wrec_lit(u32x4 v, uint32_t lit, uint32_t b16)
{
    return put_byte(v, (lit & 0xFFFFu) - b16, (lit >> 16) & 0xFFu);
}

__device__ __forceinline__ u32x4 wmerge(u32x4 v, u32x4 ld, u32x4 m)      // bytes of ld where m is set
{
    v[0] = (ld[0] & m[0]) | (v[0] & ~m[0]);
    v[1] = (ld[1] & m[1]) | (v[1] & ~m[1]);
    v[2] = (ld[2] & m[2]) | (v[2] & ~m[2]);
    v[3] = (ld[3] & m[3]) | (v[3] & ~m[3]);
    return v;
}

// WPG waves per workgroup, each with its own chunk and its own LDS tables; the waves of a workgroup share nothing but the
// (identical) mask and selector tables.
//
// (__launch_bounds__(.., 8): eight waves per SIMD, i.e. 64 VGPRs -- the kernel lives on occupancy; with ten rows in flight it
// takes 62 and no scratch; twelve rows spill the gathers' destination registers.)
// Everything between the descriptor load and the last store is STRAIGHT-LINE code: selects, dummy addresses (a readable buffer of
// dots) and range-checked buffer stores instead of branches.  hipcc cannot count outstanding memory operations across a branch
// that may or may not issue one; it then waits vmcnt(0) at the next use -- round 2's stitch4_kernel had "if (patch) read LDS
// else gather" per row and a lane-conditional store per row, and its code waited for every gather before issuing the next one
// and for every store's acknowledgement before issuing the next store (its own stamps: 7 500 + 4 900 cycles of a workgroup's
// 24 400).  Here the ten gathers issue back to back and every wait is a counted one.
#ifndef V2P_WAVE_OCC
#define V2P_WAVE_OCC 8
#endif
// RIMG: the image is a rows image (sir_pack.hpp: CHUNK_CLIP) -- every chunk starts on a 1 KiB row (no ragged head), may skip the
// head of its first descriptor and clip its last one; an instance of its own, so that the host packer's images run the code they
// always ran (the kernel sits at 62-64 VGPRs: a handful of extra live values made hipcc spill a gathered row).
// STG (rows images launched in phases, the ride form): the chunk's descriptors are read from row c of a STAGING buffer that the read-ahead
// of this phase filled (touch_chunks: 64 slots per chunk, in launch order) -- an address the wave knows before its chunk record has
// arrived, so record and descriptors are requested together -- and the trailing workgroups fill the other buffer for the next phase.
template <int WPG, bool NT, bool SC1 = false, bool RIMG = false, bool STG = false>
__global__ __launch_bounds__(64 * WPG, V2P_WAVE_OCC) void stitchw_kernel(const uint64_t* __restrict__ p_desc, const Chunk* __restrict__ p_chunks,
                                                            const uint8_t* __restrict__ p_src0, const uint8_t* __restrict__ p_src1,
                                                            uint8_t* __restrict__ p_out, unsigned long long* __restrict__ p_status,
                                                            const uint8_t* __restrict__ p_dots,
                                                            uint32_t n_chunks, uint64_t n_desc, uint64_t src0_len, uint64_t src1_len, uint64_t out_len,
                                                            const Chunk* __restrict__ p_next, uint32_t n_next, uint32_t phase_chunks,
                                                            const uint64_t* __restrict__ p_stage = nullptr, uint64_t* __restrict__ p_stage_next = nullptr)
{
    static_assert(!STG || (RIMG && WPG == 1), "staged descriptors: rows images, one wave per workgroup");
    constexpr uint32_t ROWS = CHUNK_BYTES_WAVE / 1024u;              // 1 KiB rows of a chunk: all gathered before the first store
    static_assert(ROWS <= 16u, "a lane's map bytes and record indices are packed four rows to a register");
    constexpr uint32_t ND = (ROWS + 3u) / 4u;                        // map dwords per lane (4 * ND >= ROWS one-byte counters)
    struct WaveLds {
        uint32_t map32[64u * ND];                                    // one byte per 16-byte block: record covering its first byte
        WRec rec[CHUNK_TASKS_WAVE + 4];                              // + sentinels
        u32x4 patch[CHUNK_TASKS_WAVE + 1];                           // [t]: record t's own piece, then (owners) the parked tail of the block it starts in; [64]: scrap
    };
    __shared__ __attribute__((aligned(16))) WaveLds s_all[WPG];
    __shared__ u32x4 s_mask[17];                                     // s_mask[j]: bytes >= j of a block
    __shared__ u32x4 s_one[17];                                      // s_one[j]: byte j of a block alone (16: none) -- where a replaced residue goes

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = WPG == 1 ? 0u : uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));
    uint32_t c = blockIdx.x * uint32_t(WPG) + wid;
    if (WPG == 1 && phase_chunks != 0u) {
        // ONE launch for all phases (launch_stitch): the grid is, phase after phase, [read-ahead workgroups of phase g + 1][stitch
        // workgroups of phase g] -- workgroups are dispatched in index order, so a phase's image is read while the phase before it
        // is being stitched, without a kernel boundary (its tail, its launch gap) between two phases.  Both group sizes are
        // multiples of 8: workgroup b stays on XCD b % 8 = the XCD of the chunks it stitches or reads ahead.
        const uint32_t tw = 8u * touch_waves_per_xcd(phase_chunks), per = tw + phase_chunks;
        const uint32_t g = blockIdx.x / per, r = blockIdx.x - g * per;
        if (r < tw) {
            const uint64_t base = uint64_t(g + 1u) * phase_chunks;
            if (base < n_chunks)
                touch_chunks(p_desc, p_chunks + base, touch_wave_first(r & 7u, r >> 3), uint32_t(n_chunks - base < phase_chunks ? n_chunks - base : phase_chunks),
                             n_desc, p_src1, src1_len, lane);
            return;
        }
        c = g * phase_chunks + (r - tw);
        if (c >= n_chunks) return;
    } else if (c >= n_chunks) {
        // the launch's trailing workgroups: they are dispatched while its last chunks drain, and read the NEXT phase's chunk
        // records, descriptors and payload lines into the caches (launch_stitch: phases) -- the read-ahead costs no launch of its own
        // (they start at a workgroup index that is a multiple of 8: workgroup b is on XCD b % 8 and reads chunks of that residue)
        const uint32_t first_wg = (((n_chunks + uint32_t(WPG) - 1u) / uint32_t(WPG)) + 7u) & ~7u;
        if (blockIdx.x >= first_wg)
            touch_chunks(p_desc, p_next, touch_wave_first(blockIdx.x & 7u, ((blockIdx.x - first_wg) >> 3) * uint32_t(WPG) + wid), n_next, n_desc, p_src1, src1_len, lane,
                         STG ? p_stage_next : nullptr);
        return;
    }
    WaveLds& L = s_all[wid];
    // (STG: requested before the chunk record is looked at -- lanes past the chunk's last descriptor read slots the read-ahead filled with fills)
    uint64_t d_stg = 0ull;
    if (STG) d_stg = p_stage[uint64_t(c) * STAGE_SLOTS + lane];
    const uint64_t tb_raw = p_chunks[c].task_begin, dn = p_chunks[c].dst_n;
    if (!(dn & CHUNK_WAVE)) return;                                  // the chunks of another kernel
    // ROWS images (sir_pack.hpp): the chunk's first descriptor may begin in the chunk before -- skip its first `hskip` bytes --
    // and its last one may go on into the next -- stop after `clip_bytes` (whole 1 KiB rows; 0: wherever the descriptors end)
    const uint64_t tb = RIMG ? tb_raw & TB_IDX_MASK : tb_raw;
    const uint32_t hskip = RIMG ? uint32_t(tb_raw >> TB_IDX_BITS) & PIECE_MAX : 0u;
    const uint32_t tclip = RIMG ? uint32_t(tb_raw >> (TB_IDX_BITS + TB_SKIP_BITS)) : 0u;
    const uint32_t n_hdr = uint32_t(dn >> 48) & CHUNK_N_MASK;
    if (RIMG != ((dn & CHUNK_CLIP) != 0ull)) { if (lane == 0u) report(p_status, tb, STATUS_RES_OOB); return; }      // (the launcher picked the wrong instance: refused, not guessed)
    // a PADDED rows image (sir_pack.hpp): the chunk's first n1 descriptors at tb, the others from the next tile's first slot (n1 = 0: all at tb)
    const uint32_t n1 = RIMG ? uint32_t(dn & CHUNK_N1_MASK) : 0u;
    const uint64_t dst = RIMG ? dn & (DST_MASK & ~CHUNK_N1_MASK) : dn & DST_MASK;
    const uint32_t head = RIMG ? 0u : uint32_t(dst) & 15u;
    const uint64_t hop = RIMG && n1 != 0u ? chunk_next_tile(tb) - tb - n1 : 0ull;      // what a lane behind the first n1 adds to tb + lane
    // a chunk table that points outside the descriptor array is refused, not followed
    const bool hdr_ok = RIMG ? n_hdr <= CHUNK_TASKS_WAVE && tb <= n_desc && n_hdr + hop <= n_desc - tb && n1 <= n_hdr
                             : n_hdr <= CHUNK_TASKS_WAVE && tb <= n_desc && n_hdr <= n_desc - tb;
    const uint32_t n = hdr_ok ? n_hdr : 0u;
    const uint32_t lofs = RIMG ? lane + (n1 != 0u && lane >= n1 ? uint32_t(hop) : 0u) : lane;      // the lane's descriptor, from tb (hop <= 256)
    // (no branch around the load: lanes past the chunk's last descriptor read the chunk header and drop it.  A prefetch of a later
    // chunk's descriptor lines into the L2 from here -- 1 Ki, 4 Ki, 16 Ki chunks ahead, issued behind this load and waited for by
    // nobody before the patch phase -- was measured: C2 +3.5 %, C3 +1 % SLOWER; not kept.)
    const uint64_t d_raw = STG ? d_stg : *(lane < n ? p_desc + tb + lofs : reinterpret_cast<const uint64_t*>(p_chunks + c));
    {   // the byte-mask table, without a branch (every lane writes an entry; lanes and waves that share one write the same value)
        const uint32_t jm = lane < 16u ? lane : 16u;
        u32x4 m;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) m[k] = jm <= 4u * k ? 0xFFFFFFFFu : (jm >= 4u * k + 4u ? 0u : 0xFFFFFFFFu << (8u * (jm - 4u * k)));
        s_mask[jm] = m;
        u32x4 o1;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) o1[k] = (jm >> 2) == k && jm < 16u ? 0xFFu << (8u * (jm & 3u)) : 0u;
        s_one[jm] = o1;
    }
#pragma unroll
    for (uint32_t k = 0; k < ND; ++k) L.map32[ND * lane + k] = 0u;
    const uint64_t d = lane < n ? d_raw : 0ull;

    // ---- A: lane = descriptor -> one record ----
    const uint64_t dots16 = reinterpret_cast<uint64_t>(p_dots) + 32u;
    const uint32_t dlo = uint32_t(d), dhi = uint32_t(d >> 32);
    const uint32_t space = dhi >> 30;
    const bool snv = (dhi >> 29) == 7u;                              // fused substitution: src 0..28, len1 29..40, len2 41..52, byte 53..60
    const bool imm = !snv && space == SPACE_IMM;                     // (a two-substitution descriptor of a dense image lands here with a huge length: refused)
    const uint32_t len1 = snv ? (dlo >> 29) | ((dhi & 0x1FFu) << 3) : (dhi >> 8) & 0x3FFFFFu;
    const uint32_t bytes0 = snv ? len1 + 1u + ((dhi >> 9) & 0xFFFu) : len1;          // the whole descriptor
    // (a record is a stream: skipping its head or dropping its tail is an address and a length)
    const uint32_t hs = RIMG && lane == 0u ? hskip : 0u, tcl = RIMG && lane + 1u == n ? tclip : 0u;
    const uint64_t src = snv ? uint64_t(dlo & 0x1FFFFFFFu) : ((uint64_t(dhi & 0xFFu) << 32) | dlo);
    const bool gathers = !imm && bytes0 != 0u && (snv || space != SPACE_FILL);      // ('.' fill, idle lanes, empty records and immediates read the dots)
    const bool ref = snv || space == SPACE_PROTEOME;
    const bool bad = (RIMG && hs + tcl != 0u && hs + tcl >= bytes0) || (imm ? len1 > IMM_MAX_BYTES : (gathers && src + bytes0 > (ref ? src0_len : src1_len)));   // never read out of bounds: task.rs would panic
    // (an immediate record's bytes ARE in memory: the low bytes of its own descriptor, just loaded -- it is a stream like any other)
    const uint64_t a = (imm ? (STG ? reinterpret_cast<uint64_t>(p_stage + uint64_t(c) * STAGE_SLOTS + lane) : reinterpret_cast<uint64_t>(p_desc + tb + lofs)) : (gathers ? reinterpret_cast<uint64_t>(ref ? p_src0 : p_src1) + src : dots16)) + (gathers || imm ? hs : 0u);
    const uint32_t bytes = bytes0 - hs - tcl;
    const uint32_t incl = wave_incl_scan(bad ? 0u : bytes);
    const uint32_t total = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
    const uint32_t ptotal = head + total;                            // end of the chunk in block space
    const uint32_t nblk = total ? (ptotal + 15u) >> 4 : 0u;
    const bool any_bad = __ballot(bad) != 0ull;
    if (!(hdr_ok && dst + total <= out_len && nblk <= CHUNK_BYTES_WAVE / 16u && (!RIMG || (uint32_t(dst) & 1023u) == 0u)) || any_bad) {     // never write out of bounds
        if (bad) report(p_status, tb + lane, STATUS_SRC_OOB);        // reported, and the chunk is not executed
        if (!any_bad && lane == 0u) report(p_status, tb, STATUS_RES_OOB);
        return;
    }
    if (total == 0u) return;
    WChk chk;
#ifdef V2P_WAVE_CHECK
    chk.lo[0] = reinterpret_cast<uint64_t>(p_src0) - PAD_BYTES; chk.hi[0] = reinterpret_cast<uint64_t>(p_src0) + src0_len + PAD_BYTES;
    chk.lo[1] = reinterpret_cast<uint64_t>(p_src1) - PAD_BYTES; chk.hi[1] = reinterpret_cast<uint64_t>(p_src1) + src1_len + PAD_BYTES;
    chk.lo[2] = reinterpret_cast<uint64_t>(p_dots); chk.hi[2] = reinterpret_cast<uint64_t>(p_dots) + DOTS_BYTES;
    chk.lo[3] = reinterpret_cast<uint64_t>(p_desc) - 16u; chk.hi[3] = reinterpret_cast<uint64_t>(p_desc + n_desc) + PAD_BYTES;      // (immediate records read their own descriptors)
    if (STG) { chk.lo[3] = reinterpret_cast<uint64_t>(p_stage) - 16u; chk.hi[3] = reinterpret_cast<uint64_t>(p_stage + uint64_t(n_chunks) * STAGE_SLOTS) + PAD_BYTES; }
    chk.status = p_status; chk.tb = c;
#endif
    const uint32_t start = ptotal - (total - (incl - bytes));        // = head + exclusive prefix; lanes >= n sit at ptotal
    const uint32_t end = start + bytes;
    const uint32_t lit_rel = len1 - hs;                              // (fused substitutions only; a skipped or clipped residue is nobody's)
    const uint32_t lit_pos = start + lit_rel;
    const uint32_t lit_byte = (dhi >> 21) & 0xFFu;
    const uint64_t adj = a - start;
    const uint32_t my_lit = snv && hs <= len1 && lit_rel < bytes ? lit_pos | (lit_byte << 16) : WREC_NOLIT;
    {
        WRec t;
        t.a_lo = uint32_t(adj); t.a_hi = uint32_t(adj >> 32);
        t.se = lane < n ? start | (end << 16) : ptotal | 0xFFFF0000u;     // sentinels past the last record: dots, ends beyond every block
        t.lit = my_lit;
        L.rec[lane] = t;
        if (lane < 4u) L.rec[CHUNK_TASKS_WAVE + lane] = WRec{uint32_t(dots16 - ptotal), uint32_t((dots16 - ptotal) >> 32), ptotal | 0xFFFF0000u, WREC_NOLIT};
        const uint32_t kmin = (start + 15u) >> 4;                    // first block starting at or after the record's start
        if (lane >= 1u && lane < n && kmin < nblk) atomicAdd(&L.map32[kmin >> 2], 1u << (8u * (kmin & 3u)));
    }
    asm volatile("" ::: "memory");                                   // (one wave: its LDS operations execute in order; this only pins the compiler)

    // ---- P, first half: lane = record.  A record that starts inside a block (at a non-zero offset) fetches ITS OWN stream at that
    //      block -- one gather per lane, unconditionally (a lane with nothing to fetch reads the dots): no branch, it flies under
    //      the map and the copy phase's look-ups.  What precedes the record in the block is the covering record's stream, which the
    //      copy phase gathers anyway; what follows it comes from the later records' own pieces. ----
    constexpr bool PG = !(V2P_WAVE_ABLATE & 2);
    const u32x4 g1 = wgather(PG && lane < n && (start & 15u) != 0u && bytes != 0u ? adj + (start & ~15u) : dots16, chk, 1u);

    // ---- C: block map = inclusive prefix sum of the marks; ROWS one-byte counters per lane (a chunk has at most 63 marks) ----
    {
        uint32_t y[ND];
        uint32_t run = 0u;                                           // marks of the lane's earlier dwords, in every byte
#pragma unroll
        for (uint32_t k = 0; k < ND; ++k) {
            uint32_t x = L.map32[ND * lane + k];
            x += x << 8; x += x << 16;
            y[k] = x + run;
            run = (y[k] >> 24) * 0x01010101u;
        }
        const uint32_t tsum = run & 0xFFu;
        const uint32_t before = (wave_incl_scan(tsum) - tsum) * 0x01010101u;
#pragma unroll
        for (uint32_t k = 0; k < ND; ++k) L.map32[ND * lane + k] = y[k] + before;
    }
    asm volatile("" ::: "memory");

    // ---- K, first half: lane = block.  Per group of four rows: the map bytes, the four records (they land in the registers the
    //      gathers will fill), then per row the gather of the covering record's stream -- for EVERY block, cut or not.  The ten
    //      gathers issue back to back, behind the patch phase's one. ----
    const uint8_t* const map8 = reinterpret_cast<const uint8_t*>(L.map32);
    uint8_t* const out0 = p_out + (dst - head);                      // 16-byte aligned
    u32x4 v[ROWS];
    uint32_t pk[ND];                                                 // record of the lane's block in row j: byte j & 3 of pk[j >> 2]
    const uint32_t lane16 = lane << 4;
#pragma unroll
    for (uint32_t g = 0; g < ND; ++g) {
        uint32_t rr[4];
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
            const uint32_t b16 = ((4u * g + q) << 10) + lane16;
            rr[q] = (4u * g + q < ROWS && b16 < ptotal) ? uint32_t(map8[b16 >> 4]) : n;     // idle lanes look at a sentinel (dots)
        }
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) if (4u * g + q < ROWS) v[4u * g + q] = reinterpret_cast<const u32x4*>(L.rec)[rr[q]];   // (one 16-byte read: field by field, the address half is sunk into a branch)
        pk[g] = rr[0] | (rr[1] << 8) | (rr[2] << 16) | (rr[3] << 24);
        asm volatile("" : "+v"(pk[g]));                              // (packed NOW: four live registers become one)
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
            const uint32_t j = 4u * g + q, b16 = (j << 10) + lane16;
            if (j >= ROWS) break;
            const u32x4 t = v[j];
            const uint64_t A = ((uint64_t(t[1]) << 32) | t[0]) + b16;
            const uint64_t X = (V2P_WAVE_ABLATE & 1) != 0 ? dots16 : A;
            v[j] = wgather(X, chk, 5u);
        }
    }

    // ---- P, second half (the copy phase's gathers are in flight): every record's own piece H[t] = its stream at the block it
    //      starts in, its replaced residue placed; the FIRST record to start inside a block then parks the block's tail: its own
    //      piece, overlaid from their starts on by the pieces of the records that follow it inside the same block. ----
    {
        // (the lane's own record comes back from LDS, field by field as it is needed: nothing of the set-up stays in registers across
        // the copy phase's look-ups, whose 32 destination registers are in flight now)
        const uint32_t* const rec32 = reinterpret_cast<const uint32_t*>(L.rec);
        const uint32_t se0 = rec32[4u * lane + 2u], lit0 = rec32[4u * lane + 3u];
        const uint32_t s0 = se0 & 0xFFFFu, e0 = se0 >> 16, b0 = s0 & ~15u;
        L.patch[lane] = wrec_lit(g1, lit0, b0);                      // (every lane: H[t]; slot t becomes the parked block below)
        // (in block 0 of a chunk with a ragged head record 0 itself starts inside the block: record 1 owns it)
        const uint32_t sp = rec32[4u * (lane - (lane != 0u ? 1u : 0u)) + 2u] & 0xFFFFu;
        const bool owner = lane >= 1u && lane < n && s0 != b0 && (sp <= b0 || (b0 == 0u && lane == 1u));
        const uint32_t phi = b0 + 16u < ptotal ? b0 + 16u : ptotal;
        const bool need2 = owner && e0 < phi;                        // the next record starts inside the block too
        const uint32_t sb16 = b0;
        u32x4 p = wmerge(L.patch[lane], L.patch[lane + 1u], s_mask[need2 ? e0 - b0 : 16u]);      // (slot 64: scrap)
        if (need2 && (rec32[4u * (lane + 1u) + 2u] >> 16) < phi) {   // a third record begins in the block (rare)
            uint32_t r = lane + 1u;
            uint32_t se = rec32[4u * r + 2u];
            while ((se >> 16) < phi) {
                se = rec32[4u * ++r + 2u];
                p = wmerge(p, L.patch[r], s_mask[(se & 0xFFFFu) - sb16]);
            }
        }
        asm volatile("" ::: "memory");                               // (every H[] read of the wave precedes the overwrite of slot t)
        L.patch[owner ? lane : CHUNK_TASKS_WAVE] = p;
    }
    asm volatile("" ::: "memory");

    // ---- K, second half: per row -- the covering record again (LDS), the parked tail merged in where the record ends inside the
    //      block, its replaced residue placed, the store.  Whole blocks of the chunk leave as 16-byte stores of a buffer resource over
    //      the chunk's result range: a lane outside it (ragged edge blocks, rows past the chunk's end) gets an out-of-range offset
    //      and the hardware drops its store -- no branch. ----
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out0, 0, int(ptotal), 0x00020000);
    const int32_t last16 = int32_t(ptotal) - 16;                     // a block at b16 <= last16 ends inside the chunk
    uint32_t l16 = lane16;
    asm volatile("" : "+v"(l16));                                    // (the rows' positions are recomputed from here, not kept in ten registers since the look-ups)
#pragma unroll
    for (uint32_t j = 0; j < ROWS; ++j) {
        const uint32_t b16 = (j << 10) + l16;
        const uint32_t r = (pk[j >> 2] >> (8u * (j & 3u))) & 0xFFu;
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 t23 = reinterpret_cast<const u32x2*>(L.rec)[2u * r + 1u];       // (start|end, literal: the record's second half)
        const uint32_t e = t23[0] >> 16;
        const bool cut = e < b16 + 16u;
        const u32x4 p = L.patch[cut ? r + 1u : CHUNK_TASKS_WAVE];
        const uint32_t ja = e - b16;                               // (1 .. 15 for a cut block inside the chunk)
        const u32x4 m = s_mask[cut && ja < 16u ? ja : 16u];
        // (the replaced residue of a fused substitution through a one-byte mask from LDS and four v_bfi, like the tail merge:
        // placing it arithmetically -- shift, compare and select per dword -- was 17 VALU a row; this is 8 and one ds_read_b128)
        const uint32_t lq = (t23[1] & 0xFFFFu) - b16;
        const u32x4 lm = s_one[lq < 16u ? lq : 16u];
        const uint32_t lb = ((t23[1] >> 16) & 0xFFu) * 0x01010101u;
        u32x4 o = v[j];
        o[0] = (lb & lm[0]) | (o[0] & ~lm[0]); o[1] = (lb & lm[1]) | (o[1] & ~lm[1]); o[2] = (lb & lm[2]) | (o[2] & ~lm[2]); o[3] = (lb & lm[3]) | (o[3] & ~lm[3]);
        o = wmerge(o, p, m);
        const bool whole = int32_t(l16) <= last16 - int32_t(j << 10) && (j != 0u || l16 >= head);
        const uint32_t off = (whole && !(V2P_WAVE_ABLATE & 4)) ? l16 : 0x80000000u;
        // (soffset stays the immediate 0: with an SGPR there, hipcc (ROCm 7.2) leaves out the wait states between a dwordx4 store and
        // a VALU write to its data registers -- seen on gfx950: "buffer_store_dwordx4 v[10:13], .., s6 offen" followed at once by
        // "v_mov_b32 v10, 0" stored a zero first dword in the last four lanes of every row of sixteen)
#ifndef V2P_WAVE_STORE_AUX
#define V2P_WAVE_STORE_AUX 2                                         /* nt */
#endif
        // (SC1: "sc1 nt" -- written through at agent scope -- for images whose descriptor stream is thin: C2 -1.5 %, C3 +6 %)
        __builtin_amdgcn_raw_buffer_store_b128(o, rsrc, int(off + (j << 10)), 0, NT ? (SC1 ? 18 : V2P_WAVE_STORE_AUX) : 0);
        __builtin_amdgcn_sched_barrier(0);                           // (row by row: hoisting the next rows' LDS reads here costs the registers the kernel does not have)
    }
    // ragged first / last block of a chunk whose cut is not 16-byte aligned (rare): one lane each, byte stores
    if (lane < 2u) {
        const uint32_t b16 = lane == 0u ? 0u : (nblk - 1u) << 4;
        if ((b16 < head || b16 + 16u > ptotal) && (lane == 0u || b16 != 0u)) {
            const uint32_t hi = b16 + 16u < ptotal ? b16 + 16u : ptotal;
            uint32_t r = uint32_t(map8[b16 >> 4]);
            WRec t = L.rec[r];
            u32x4 o = wgather(wrec_adj(t) + b16, chk, 6u);
            o = wrec_lit(o, t.lit, b16);
            while ((t.se >> 16) < hi) {
                t = L.rec[++r];
                const u32x4 g = wgather(wrec_adj(t) + b16, chk, 7u);
                o = wmerge(o, g, s_mask[(t.se & 0xFFFFu) - b16]);
                o = wrec_lit(o, t.lit, b16);
            }
            const uint32_t ka = b16 < head ? head - b16 : 0u, kb = hi - b16;
            for (uint32_t q = ka; q < kb; ++q) out0[b16 + q] = uint8_t(o[q >> 2] >> (8u * (q & 3u)));
        }
    }
}

hipError_t launch_stitch_wave(const StitchArgs& a, hipStream_t stream, bool nt, int waves_per_group)
{
    if (a.n_chunks == 0) return hipSuccess;
    if (a.phase_chunks != 0u && waves_per_group == 1) {              // every phase in one launch, read-ahead workgroups interleaved (see the kernel)
        const uint32_t P = a.phase_chunks, tw = 8u * touch_waves_per_xcd(P);
        const uint64_t groups = (uint64_t(a.n_chunks) + P - 1u) / P, grid = groups * (uint64_t(tw) + P);
        if (grid > 0x7FFFFFFFull) return hipErrorInvalidValue;
        if (a.rows) {                                                // (an A/B switch: v2p_set_launch_opts, variant 16)
            if (nt) hipLaunchKernelGGL((stitchw_kernel<1, true, false, true>), dim3(uint32_t(grid)), dim3(64), 0, stream, a.desc, a.chunks, a.src0, a.src1, a.out, a.status, a.dots,
                                       a.n_chunks, a.n_desc, a.src0_len, a.src1_len, a.out_len, nullptr, 0u, P);
            else hipLaunchKernelGGL((stitchw_kernel<1, false, false, true>), dim3(uint32_t(grid)), dim3(64), 0, stream, a.desc, a.chunks, a.src0, a.src1, a.out, a.status, a.dots,
                                    a.n_chunks, a.n_desc, a.src0_len, a.src1_len, a.out_len, nullptr, 0u, P);
            return hipGetLastError();
        }
        if (nt) hipLaunchKernelGGL((stitchw_kernel<1, true>), dim3(uint32_t(grid)), dim3(64), 0, stream, a.desc, a.chunks, a.src0, a.src1, a.out, a.status, a.dots,
                                   a.n_chunks, a.n_desc, a.src0_len, a.src1_len, a.out_len, nullptr, 0u, P);
        else hipLaunchKernelGGL((stitchw_kernel<1, false>), dim3(uint32_t(grid)), dim3(64), 0, stream, a.desc, a.chunks, a.src0, a.src1, a.out, a.status, a.dots,
                                a.n_chunks, a.n_desc, a.src0_len, a.src1_len, a.out_len, nullptr, 0u, P);
        return hipGetLastError();
    }
    // (one launch.  Sub-launches of 8 Ki ... 128 Ki chunks -- the waves of a fresh launch read their descriptors together and store
    // together, which a bare copy kernel rewards: tools/wave_copy_bench.py -- cost this kernel 1 ... 17 %: V2P_WAVE_SUB, experiments only)
    uint32_t sub = 0;
#ifdef V2P_BENCH_VARIANTS
    if (const char* e = getenv("V2P_WAVE_SUB")) sub = uint32_t(strtoul(e, nullptr, 10));
#endif
    if (sub == 0 || sub > a.n_chunks) sub = a.n_chunks;
    sub = (sub + 7u) & ~7u;                                          // (keeps workgroup b on the XCD the chunk order dealt chunk b to)
    for (uint32_t c0 = 0; c0 < a.n_chunks; c0 += sub) {
        const uint32_t nc = a.n_chunks - c0 < sub ? a.n_chunks - c0 : sub;
        const Chunk* ch = a.chunks + c0;
        const bool last = c0 + sub >= a.n_chunks;                    // (the read-ahead rides on the launch's last sub-launch)
        const uint32_t n_next = last && a.next_chunks ? a.n_next : 0u;
        const uint32_t tw = n_next ? touch_waves_per_xcd(n_next) : 0u;                         // read-ahead waves per XCD
#define V2P_LW(WW, NTT) hipLaunchKernelGGL((stitchw_kernel<WW, NTT>), dim3(tw ? ((((nc + (WW) - 1u) / (WW)) + 7u) & ~7u) + 8u * ((tw + (WW) - 1u) / (WW)) : (nc + (WW) - 1u) / (WW)), dim3(64 * (WW)), 0, stream, \
        a.desc, ch, a.src0, a.src1, a.out, a.status, a.dots, nc, a.n_desc, a.src0_len, a.src1_len, a.out_len, a.next_chunks, n_next, 0u)
#define V2P_LWR(NTT, SCC) hipLaunchKernelGGL((stitchw_kernel<1, NTT, SCC, true>), dim3(tw ? (((nc + 7u) & ~7u) + 8u * tw) : nc), dim3(64), 0, stream, \
        a.desc, ch, a.src0, a.src1, a.out, a.status, a.dots, nc, a.n_desc, a.src0_len, a.src1_len, a.out_len, a.next_chunks, n_next, 0u, nullptr, nullptr)
#define V2P_LWS(NTT, SCC) hipLaunchKernelGGL((stitchw_kernel<1, NTT, SCC, true, true>), dim3(tw ? (((nc + 7u) & ~7u) + 8u * tw) : nc), dim3(64), 0, stream, \
        a.desc, ch, a.src0, a.src1, a.out, a.status, a.dots, nc, a.n_desc, a.src0_len, a.src1_len, a.out_len, a.next_chunks, n_next, 0u, a.stage_cur, a.stage_next)
        if (a.rows && a.stage_cur && c0 == 0u && nc == a.n_chunks) {      // (staged descriptors: the whole phase is one launch, row c of the buffer = chunk c)
            if (nt && a.store_sc1) V2P_LWS(true, true); else if (nt) V2P_LWS(true, false); else V2P_LWS(false, false);
        } else if (a.rows) { if (nt && a.store_sc1) V2P_LWR(true, true); else if (nt) V2P_LWR(true, false); else V2P_LWR(false, false); }
#ifdef V2P_BENCH_VARIANTS
        else if (waves_per_group == 4) { if (nt) V2P_LW(4, true); else V2P_LW(4, false); }
        else if (waves_per_group == 2) { if (nt) V2P_LW(2, true); else V2P_LW(2, false); }
#endif
        else if (nt && a.store_sc1) hipLaunchKernelGGL((stitchw_kernel<1, true, true>), dim3(tw ? (((nc + 7u) & ~7u) + 8u * tw) : nc), dim3(64), 0, stream,
                                                      a.desc, ch, a.src0, a.src1, a.out, a.status, a.dots, nc, a.n_desc, a.src0_len, a.src1_len, a.out_len, a.next_chunks, n_next, 0u);
        else { if (nt) V2P_LW(1, true); else V2P_LW(1, false); }
#undef V2P_LW
#undef V2P_LWR
#undef V2P_LWS
    }
    return hipGetLastError();
}

__global__ void code_object_loader_b() {}
hipError_t preload_stitch_wave(hipStream_t stream)
{
    hipLaunchKernelGGL(code_object_loader_b, dim3(1), dim3(64), 0, stream);
    return hipGetLastError();
}

}  // namespace v2p
