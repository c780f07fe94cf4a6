// stitch_wave.hip -- stitchw_kernel: the SIR executor with ONE WAVE PER CHUNK (gfx950, wave64; no MFMA: byte/index work).
//
// Same contract as the other stitch kernels (task.rs:38-50 for a whole batch of haplotypes per launch, descriptors and chunk
// table of sir_pack.hpp), built from what the round-2 measurements say limits them on MI355X:
//   * a workgroup of four waves pays five barriers and a cross-wave combine for every scan, and its wave slots stay idle
//     until its slowest wave has stored its last row -- a third of the long-run kernel's time was per-workgroup cost;
//   * the per-block kernel is VALU-bound because every lane runs the multi-source merge for every block;
//   * a fused substitution was expanded back into three tasks, which tripled every table the set-up builds.
// Here a chunk is at most 64 descriptors and 8 KiB of result and belongs to ONE wave: lane = descriptor in the set-up, lane =
// 16-byte block in the copy.  Nothing crosses a wave, so there is no s_barrier at all (LDS operations of one wave execute in
// order), scans are one DPP pass with the total read by v_readlane, and a wave slot is free the moment its own eight rows are
// stored.  A descriptor is ONE record -- a fused substitution is a reference run with one residue replaced, not three tasks:
//   A  descriptor -> record {source address - start, start, end, substituted position + byte}; DPP scan of the lengths;
//      +1 scattered into a byte-per-block map at the first block starting inside or after each record
//   C  in-lane SWAR prefix + wave scan of the map: map[k] = record covering the first byte of block k (8 blocks per lane)
//   P  lane = record: the first record to start strictly inside a block assembles that block once (every source that touches
//      it, masks from a 17-entry LDS table) and parks it in an LDS patch table; a record whose substituted residue lies in a
//      block it covers whole parks that block too (one gather + one byte insert).  Their gathers fly under C.
//   K  lane = block, eight 1 KiB rows per wave: one map byte + one record per block; a block no record ends in and no
//      substitution touches is 16 bytes of its record's stream (one byte-granular dwordx4 gather), anything else is read
//      from the patch table into the same registers.  All eight rows are gathered before the first store (gfx950 counts
//      loads and stores in one in-order counter), then leave as aligned non-temporal dwordx4 stores, 1 KiB per instruction.
// A descriptor that would read out of bounds is reported in the device status word and its chunk is not executed; nothing is
// ever read or written outside the buffers (sources carry PAD_BYTES of readable slack, as for the other kernels).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "stitch_kernels.h"
#include "stitch_device.hpp"

namespace v2p {

struct __attribute__((aligned(16))) WRec {
    // positions are in BLOCK SPACE: (chunk's result offset & 15) + offset inside the chunk, so block b covers [16b, 16b + 16)
    uint32_t a_lo, a_hi;   // source address minus the record's start position; an immediate record: its literal bytes
    uint32_t se;           // start | end << 16
    uint32_t lit;          // position of the substituted residue (WREC_NOLIT: none) | its byte << 16 | WREC_IMM
};
constexpr uint32_t WREC_IMM = 0x80000000u, WREC_NOLIT = 0xFFFFu;

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

// 16 bytes of record `t` for the block at position b16 (bytes before the record's start / after its end are whatever lies there)
__device__ __forceinline__ u32x4 wrec_fetch(const WRec& t, uint32_t b16)
{
    if (t.lit & WREC_IMM) return imm_block(wrec_adj(t), int32_t((t.se & 0xFFFFu) - b16));
    u32x4 v = gather16(wrec_adj(t) + b16);
    const uint32_t q = (t.lit & 0xFFFFu) - b16;
    if (q < 16u) v = put_byte(v, q, (t.lit >> 16) & 0xFFu);
    return v;
}

__device__ __forceinline__ u32x4 wmerge(u32x4 v, u32x4 ld, u32x4 m)      // bytes of ld where m is set
{
    v[0] = (ld[0] & m[0]) | (v[0] & ~m[0]);
    v[1] = (ld[1] & m[1]) | (v[1] & ~m[1]);
    v[2] = (ld[2] & m[2]) | (v[2] & ~m[2]);
    v[3] = (ld[3] & m[3]) | (v[3] & ~m[3]);
    return v;
}

// The block at position b16 whose first byte lies in record r: that record's stream, overwritten from their start on by every
// record that begins before the block (or the chunk) ends.  In two halves, so that a lane's gathers are in flight together and
// under other work: `issue` starts the fetches of the first three sources, `finish` merges them (a fourth and later source: a loop).
struct WFetch { u32x4 v, g1, g2; uint32_t ja1, ja2, next; };   // ja: where source 1 / 2 starts inside the block (16: unused); next: rank to go on with (0: done)
__device__ __forceinline__ WFetch wblock_issue(const WRec* rec, uint32_t r, uint32_t b16, uint32_t ptotal)
{
    const uint32_t hi = b16 + 16u < ptotal ? b16 + 16u : ptotal;
    const WRec t0 = rec[r], t1 = rec[r + 1u], t2 = rec[r + 2u];
    const bool need1 = (t0.se >> 16) < hi, need2 = need1 && (t1.se >> 16) < hi;
    WFetch f;
    f.v = wrec_fetch(t0, b16); f.g1 = f.v; f.g2 = f.v;
    if (need1) f.g1 = wrec_fetch(t1, b16);
    if (need2) f.g2 = wrec_fetch(t2, b16);
    f.ja1 = need1 ? (t1.se & 0xFFFFu) - b16 : 16u;
    f.ja2 = need2 ? (t2.se & 0xFFFFu) - b16 : 16u;
    f.next = (need2 && (t2.se >> 16) < hi) ? r + 2u : 0u;
    return f;
}
__device__ __forceinline__ u32x4 wblock_finish(const WFetch& f, const WRec* rec, const u32x4* s_mask, uint32_t b16, uint32_t ptotal)
{
    u32x4 v = wmerge(f.v, f.g1, s_mask[f.ja1]);
    v = wmerge(v, f.g2, s_mask[f.ja2]);
    if (f.next) {                                             // four or more records in this block
        const uint32_t hi = b16 + 16u < ptotal ? b16 + 16u : ptotal;
        uint32_t r = f.next;
        WRec t = rec[r];
        while ((t.se >> 16) < hi) {
            t = rec[++r];
            v = wmerge(v, wrec_fetch(t, b16), s_mask[(t.se & 0xFFFFu) - b16]);
        }
    }
    return v;
}

// WPG waves per workgroup, each with its own chunk and its own LDS tables; the waves of a workgroup share nothing but the
// (identical) byte-mask table.  WPG = 1: a wave slot is refilled the moment its wave ends.
template <int WPG, bool NT>
__global__ __launch_bounds__(64 * WPG) void stitchw_kernel(const uint64_t* __restrict__ p_desc, const Chunk* __restrict__ p_chunks,
                                                            const uint8_t* __restrict__ p_src0, const uint8_t* __restrict__ p_src1,
                                                            uint8_t* __restrict__ p_out, unsigned long long* __restrict__ p_status,
                                                            const uint8_t* __restrict__ p_dots,
                                                            uint32_t n_chunks, uint64_t n_desc, uint64_t src0_len, uint64_t src1_len, uint64_t out_len)
{
    constexpr uint32_t ROWS = CHUNK_BYTES_WAVE / 1024u;              // 1 KiB rows of a chunk: all gathered before the first store
    struct WaveLds {
        uint32_t map32[CHUNK_BYTES_WAVE / 64u];                      // one byte per 16-byte block: record covering its first byte
        WRec rec[CHUNK_TASKS_WAVE + 4];                              // + sentinels
        u32x4 patch[2 * CHUNK_TASKS_WAVE];                           // [t]: the block record t is the first to start in; [64 + r]: the block of r's substituted residue
    };
    __shared__ __attribute__((aligned(16))) WaveLds s_all[WPG];
    __shared__ u32x4 s_mask[17];                                     // s_mask[j]: bytes >= j of a block

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = WPG == 1 ? 0u : uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));
    const uint32_t c = blockIdx.x * uint32_t(WPG) + wid;
    if (c >= n_chunks) return;
    WaveLds& L = s_all[wid];
    const uint64_t tb = p_chunks[c].task_begin, dn = p_chunks[c].dst_n;
    if (!(dn & CHUNK_WAVE)) return;                                  // the chunks of another kernel
    const uint32_t n_hdr = uint32_t(dn >> 48) & CHUNK_N_MASK;
    const uint64_t dst = dn & DST_MASK;
    const uint32_t head = uint32_t(dst) & 15u;
    // a chunk table that points outside the descriptor array is refused, not followed
    const bool hdr_ok = n_hdr <= CHUNK_TASKS_WAVE && tb <= n_desc && n_hdr <= n_desc - tb;
    const uint32_t n = hdr_ok ? n_hdr : 0u;
    const uint64_t d = lane < n ? p_desc[tb + lane] : 0ull;
    if (lane < 17u) {                                                // (every wave of the workgroup writes the same values)
        u32x4 m;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) m[k] = lane <= 4u * k ? 0xFFFFFFFFu : (lane >= 4u * k + 4u ? 0u : 0xFFFFFFFFu << (8u * (lane - 4u * k)));
        s_mask[lane] = m;
    }
    reinterpret_cast<uint64_t*>(L.map32)[lane] = 0ull;

    // ---- A: lane = descriptor -> one record ----
    const uint64_t dots16 = reinterpret_cast<uint64_t>(p_dots) + 32u;
    const uint32_t dlo = uint32_t(d), dhi = uint32_t(d >> 32);
    const uint32_t space = dhi >> 30;
    const bool snv = (dhi >> 29) == 7u;                              // fused substitution: src 0..28, len1 29..40, len2 41..52, byte 53..60
    const bool imm = !snv && space == SPACE_IMM;                     // (a two-substitution descriptor of a dense image lands here with a huge length: refused)
    const uint32_t len1 = snv ? (dlo >> 29) | ((dhi & 0x1FFu) << 3) : (dhi >> 8) & 0x3FFFFFu;
    const uint32_t bytes = snv ? len1 + 1u + ((dhi >> 9) & 0xFFFu) : len1;
    const uint64_t src = snv ? uint64_t(dlo & 0x1FFFFFFFu) : ((uint64_t(dhi & 0xFFu) << 32) | dlo);
    uint64_t a = dots16;                                             // '.' fill, idle lanes, empty records, immediates (a readable dummy)
    bool bad = false;
    if (imm) bad = len1 > IMM_MAX_BYTES;
    else if (bytes != 0u && (snv || space != SPACE_FILL)) {
        const bool ref = snv || space == SPACE_PROTEOME;
        bad = src + bytes > (ref ? src0_len : src1_len);             // never read out of bounds: task.rs would panic
        a = reinterpret_cast<uint64_t>(ref ? p_src0 : p_src1) + src;
    }
    const uint32_t incl = wave_incl_scan(bad ? 0u : bytes);
    const uint32_t total = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
    const uint32_t ptotal = head + total;                            // end of the chunk in block space
    const uint32_t nblk = total ? (ptotal + 15u) >> 4 : 0u;
    const bool any_bad = __ballot(bad) != 0ull;
    if (bad) report(p_status, tb + lane, STATUS_SRC_OOB);            // reported, and the chunk is not executed
    if (!(hdr_ok && dst + total <= out_len && nblk <= CHUNK_BYTES_WAVE / 16u) || any_bad) {     // never write out of bounds
        if (!any_bad && lane == 0u) report(p_status, tb, STATUS_RES_OOB);
        return;
    }
    if (total == 0u) return;
    const uint32_t start = ptotal - (total - (incl - bytes));        // = head + exclusive prefix; lanes >= n sit at ptotal
    const uint32_t end = start + bytes;
    const uint32_t lit_pos = start + len1;                           // (fused substitutions only)
    {
        const uint64_t adj = imm ? src : a - start;
        WRec t;
        t.a_lo = uint32_t(adj); t.a_hi = uint32_t(adj >> 32);
        t.se = lane < n ? start | (end << 16) : ptotal | 0xFFFF0000u;     // sentinels past the last record: dots, ends beyond every block
        t.lit = (snv ? lit_pos | (((dhi >> 21) & 0xFFu) << 16) : WREC_NOLIT) | (imm ? WREC_IMM : 0u);
        L.rec[lane] = t;
        if (lane < 4u) L.rec[CHUNK_TASKS_WAVE + lane] = WRec{uint32_t(dots16 - ptotal), uint32_t((dots16 - ptotal) >> 32), ptotal | 0xFFFF0000u, WREC_NOLIT};
        const uint32_t kmin = (start + 15u) >> 4;                    // first block starting at or after the record's start
        if (lane >= 1u && lane < n && kmin < nblk) atomicAdd(&L.map32[kmin >> 2], 1u << (8u * (kmin & 3u)));
    }
    asm volatile("" ::: "memory");                                   // (one wave: its LDS operations execute in order; this only pins the compiler)

    // ---- P, first half: lane = record.  Record t owns the block it starts in when it is the first record to start there (at a
    //      non-zero offset); a fused substitution owns the block of its replaced residue when it covers that block whole. ----
    const uint32_t prev_start = uint32_t(__builtin_amdgcn_update_dpp(0, int(start), 0x138, 0xf, 0xf, false));   // wave_shr:1
    const uint32_t sb16 = start & ~15u;
    const bool owner = lane >= 1u && lane < n && start != sb16 && (sb16 == 0u ? lane == 1u : prev_start <= sb16);
    const uint32_t lb16 = lit_pos & ~15u;
    const bool lit_owner = snv && start <= lb16 && end >= lb16 + 16u;
    WFetch pf;
    u32x4 lv = {0u, 0u, 0u, 0u};
    if (owner) pf = wblock_issue(L.rec, lane - 1u, sb16, ptotal);
    if (lit_owner) lv = gather16(a - start + lb16);

    // ---- C: block map = inclusive prefix sum of the marks; 8 one-byte counters per lane (a chunk has at most 63 marks) ----
    {
        uint64_t x = reinterpret_cast<const uint64_t*>(L.map32)[lane];
        uint32_t y0 = uint32_t(x), y1 = uint32_t(x >> 32);
        y0 += y0 << 8; y0 += y0 << 16;
        y1 += y1 << 8; y1 += y1 << 16;
        y1 += (y0 >> 24) * 0x01010101u;
        const uint32_t tsum = y1 >> 24;
        const uint32_t before = (wave_incl_scan(tsum) - tsum) * 0x01010101u;
        y0 += before; y1 += before;
        reinterpret_cast<uint64_t*>(L.map32)[lane] = (uint64_t(y1) << 32) | y0;
    }
    // ---- P, second half: merge and park ----
    if (owner) L.patch[lane] = wblock_finish(pf, L.rec, s_mask, sb16, ptotal);
    if (lit_owner) L.patch[CHUNK_TASKS_WAVE + lane] = put_byte(lv, lit_pos & 15u, (dhi >> 21) & 0xFFu);
    asm volatile("" ::: "memory");

    // ---- K: lane = block.  Look-ups first (LDS only), then the gathers back to back, then the stores back to back. ----
    const uint8_t* const map8 = reinterpret_cast<const uint8_t*>(L.map32);
    uint8_t* const out0 = p_out + (dst - head);                      // 16-byte aligned
    uint64_t X[ROWS];
#pragma unroll
    for (uint32_t j = 0; j < ROWS; ++j) {
        const uint32_t b16 = (j << 10) + (lane << 4);
        const uint32_t r = b16 < ptotal ? uint32_t(map8[b16 >> 4]) : n;     // idle lanes look at a sentinel (dots)
        const WRec t = L.rec[r];
        X[j] = wrec_adj(t) + b16;
        // a record ends inside the block: the block the next record parked (address 0 | patch index); a replaced residue inside it:
        // the block its record parked
        if ((t.se >> 16) < b16 + 16u) X[j] = uint64_t(r + 1u);
        else if ((t.lit & 0xFFFFu) - b16 < 16u) X[j] = uint64_t(CHUNK_TASKS_WAVE + r);
    }
    u32x4 v[ROWS];
#pragma unroll
    for (uint32_t j = 0; j < ROWS; ++j) {
        if ((j << 10) >= ptotal) continue;                           // (uniform: rows past the chunk's end)
        // (if / else, not a select: gather and patch land in the same registers, lanes disjoint)
        if (uint32_t(X[j] >> 32) == 0u) v[j] = L.patch[uint32_t(X[j])];
        else v[j] = gather16(X[j]);
    }
#pragma unroll
    for (uint32_t j = 0; j < ROWS; ++j) {
        if ((j << 10) >= ptotal) continue;
        const uint32_t b16 = (j << 10) + (lane << 4);
        if (b16 >= head && b16 + 16u <= ptotal) {                    // whole blocks of the chunk; ragged edge blocks are written below
            if (NT) __builtin_nontemporal_store(v[j], reinterpret_cast<u32x4*>(out0 + b16));
            else *reinterpret_cast<u32x4*>(out0 + b16) = v[j];
        }
    }
    // ragged first / last block of a chunk whose cut is not 16-byte aligned (rare): one lane each, byte stores
    if (lane < 2u) {
        const uint32_t b16 = lane == 0u ? 0u : (nblk - 1u) << 4;
        if ((b16 < head || b16 + 16u > ptotal) && (lane == 0u || b16 != 0u)) {
            const WFetch f = wblock_issue(L.rec, uint32_t(map8[b16 >> 4]), b16, ptotal);
            const u32x4 o = wblock_finish(f, L.rec, s_mask, b16, ptotal);
            const uint32_t ka = b16 < head ? head - b16 : 0u, kb = (b16 + 16u < ptotal ? b16 + 16u : ptotal) - b16;
            for (uint32_t q = ka; q < kb; ++q) out0[b16 + q] = uint8_t(o[q >> 2] >> (8u * (q & 3u)));
        }
    }
}

hipError_t launch_stitch_wave(const StitchArgs& a, hipStream_t stream, bool nt, int waves_per_group)
{
    if (a.n_chunks == 0) return hipSuccess;
#define V2P_LW(WW, NTT) hipLaunchKernelGGL((stitchw_kernel<WW, NTT>), dim3((a.n_chunks + (WW) - 1u) / (WW)), dim3(64 * (WW)), 0, stream, \
        a.desc, a.chunks, a.src0, a.src1, a.out, a.status, a.dots, a.n_chunks, a.n_desc, a.src0_len, a.src1_len, a.out_len)
    if (waves_per_group == 4) { if (nt) V2P_LW(4, true); else V2P_LW(4, false); }
    else if (waves_per_group == 2) { if (nt) V2P_LW(2, true); else V2P_LW(2, false); }
    else { if (nt) V2P_LW(1, true); else V2P_LW(1, false); }
#undef V2P_LW
    return hipGetLastError();
}

}  // namespace v2p
