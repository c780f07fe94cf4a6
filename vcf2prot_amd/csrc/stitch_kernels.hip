// stitch_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the SIR executor.
//
// Replaces the reference's step-6 loop
//     for task in g_rep { res[dst..dst+len] = (code==0 ? ref : alt)[src..src+len] }
// (task.rs:38-50, gir.rs:230-234) for a whole batch of haplotypes per launch.
// Integer/byte work only: HBM-bound gather/scatter, no MFMA.
//
// Kernels
//   stitch_kernel   K2 (+K0 fused, +in-chunk K1): one 256-lane workgroup per chunk of
//                   <=256 descriptors / <64 KiB of result.  Lanes load one 8-byte
//                   descriptor each (coalesced); a wave64 DPP prefix scan + ballot ranks
//                   turn lengths into result offsets and compact the non-empty tasks in
//                   LDS; a scatter + SWAR prefix sum builds a block->task map; then every
//                   lane owns 16-byte aligned result blocks: one unaligned dwordx4 gather
//                   per overlapping task, tail-overwrite merge with 64-bit masks, and one
//                   aligned dwordx4 store.  Result stores are full 16-byte and fully
//                   coalesced (1 KiB per wave instruction) whatever the source alignments.
//   stitch4_kernel  long-run images (C2): see the comment at the kernel.
//   stitch_dense_kernel  images of short tasks (C5): lane = task, LDS image of the chunk, see the comment at the kernel.
//   ordered_kernel  reference-order execution for non-canonical Task vectors
//                   (overlapping / descending result ranges): one workgroup, tasks in
//                   order, barrier between tasks => "later task wins" as on the CPU.
//   validate_kernel DEBUG_GPU: first row that breaks code / bounds / the contiguity
//                   predicate of gir.rs:208-226 (wave ballot + one atomicMin per wave).
//   digest_kernel   per-haplotype position-sensitive checksum of the result arena.
#include <hip/hip_runtime.h>
#include <utility>
#include <vector>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
#include "stitch_kernels.h"
#include "stitch_device.hpp"

namespace v2p {

// Per chunk (one workgroup, 256 lanes, TPT consecutive descriptors per lane):
//   A  decode + bounds-check the descriptors; wave64 DPP scan of the lengths; ballot/mbcnt
//      rank among the non-empty tasks; zero the block map
//   B  compact the non-empty tasks by rank into LDS: s_off[r] (result offset inside the
//      chunk) and s_adj[r] (source address minus that offset; '.' fill tasks point into a
//      device buffer of dots; immediate tasks keep their literal bytes), and scatter "+1" into
//      the block map at the first 16-byte block that starts inside or after task r (r >= 1)
//   C,D prefix-sum the block map (16 two-byte counters per lane, SWAR + wave scan), so
//      map[k] = rank of the task covering the first byte of result block k -- the
//      per-block search costs one LDS read instead of a binary search
//   K2 every lane assembles aligned 16-byte result blocks, 64 consecutive blocks per wave and pass.
//      The task r covering a block's first byte is its *primary* stream: the lane loads the ALIGNED
//      16-byte source block holding that byte (a wave touches every 128-byte line once) and takes the
//      rest from the next lane, which holds the following aligned block of the same stream
//      (DPP wave_shl:1 + a funnel shift by the stream's misalignment); only lanes whose neighbour is on
//      another stream (task boundaries, lane 63) load a second block themselves.  Tasks r+1, r+2 that
//      begin inside the block are merged over it by tail-overwrite with 64-bit byte masks: immediate
//      tasks (SNV / short indel payloads) come out of the LDS table with no memory access, a task that
//      continues the primary stream (the reference after an SNV) reuses the primary bytes, everything
//      else is one exec-masked unaligned gather.  Blocks cut by four or more tasks take an extra loop.
//      Then one aligned, non-temporal dwordx4 store -- 1 KiB per wave instruction.
// VAR 1 = legacy K2 (one byte-granular dwordx4 gather per overlapping task), kept for A/B runs.
// DBG != 0: timing-only ablations (results are wrong): 1 = no loads, 2 = no stores; 20 = s_memtime stamps.
__device__ __forceinline__ u32x4 overwrite_tail(u32x4 v, u32x4 ld, uint32_t ja, bool take)
{
    // bytes >= ja of the block come from ld (ja in 1..15)
    const uint64_t x = ~0ull << (8u * (ja & 7u));
    const uint64_t mlo = (take && ja < 8u) ? x : 0ull;
    const uint64_t mhi = take ? (ja < 8u ? ~0ull : x) : 0ull;
    const uint32_t k0 = uint32_t(mlo), k1 = uint32_t(mlo >> 32), k2 = uint32_t(mhi), k3 = uint32_t(mhi >> 32);
    v[0] = (v[0] & ~k0) | (ld[0] & k0);
    v[1] = (v[1] & ~k1) | (ld[1] & k1);
    v[2] = (v[2] & ~k2) | (ld[2] & k2);
    v[3] = (v[3] & ~k3) | (ld[3] & k3);
    return v;
}

// 16 bytes of a non-primary task for the block at `rel`: literal bytes placed at block position q, or a gather
template <bool DW>
__device__ __forceinline__ u32x4 fetch_task(uint64_t a, int32_t rel, int32_t q)
{
    if (a & ADJ_IMM) return imm_block(a & ADJ_LIT, q);
    return DW ? gather16_dw(a + int64_t(rel)) : gather16(a + int64_t(rel));
}

// TPT = descriptors per lane: a chunk holds up to 256*TPT tasks.  The per-chunk set-up (two
// dependent HBM latencies, four barriers) is the same for any TPT, so larger chunks amortise it.
template <int TPT, bool NT, int VAR = 0, int DBG = 0>
__global__ __launch_bounds__(256) void stitch_kernel(const uint64_t* __restrict__ p_desc, const Chunk* __restrict__ p_chunks,
                                                     const uint8_t* __restrict__ p_src0, const uint8_t* __restrict__ p_src1,
                                                     uint8_t* __restrict__ p_out, unsigned long long* __restrict__ p_status,
                                                     const uint8_t* __restrict__ p_dots,
                                                     uint32_t n_chunks, uint64_t n_desc, uint64_t src0_len, uint64_t src1_len, uint64_t out_len,
                                                     uint32_t filter)
{
    // explicit __restrict__ pointers (not a by-value struct): the chunk header becomes a scalar
    // (s_load) access, because the compiler can prove the result stores never clobber the inputs
    struct { const uint64_t* desc; const Chunk* chunks; uint32_t n_chunks; uint64_t n_desc; const uint8_t* src0; uint64_t src0_len;
             const uint8_t* src1; uint64_t src1_len; uint8_t* out; uint64_t out_len; unsigned long long* status; const uint8_t* dots; }
        a{p_desc, p_chunks, n_chunks, n_desc, p_src0, src0_len, p_src1, src1_len, p_out, out_len, p_status, p_dots};

    constexpr uint32_t K = 256u * TPT;
    constexpr bool DW = TPT >= 4;                         // dense descriptors (short tasks): dword-aligned gathers win
    __shared__ __attribute__((aligned(16))) uint32_t s_map32[2048 + 8];     // 4096 two-byte block->rank entries
    __shared__ uint64_t s_adj[K + 8];
    __shared__ uint32_t s_off[K + 8];
    __shared__ uint32_t s_w[3][4];
    // The kernel is VALU-bound (SQ_INSTS_VALU x 4 cycles / 1024 SIMDs = its duration): byte masks and literal placement come out
    // of two small LDS tables instead of 64-bit shift / select sequences (13 -> 6 and 17 -> 6 VALU instructions per use).
    __shared__ u32x4 s_tail[17];                          // s_tail[j]: bytes >= j of a 16-byte block (j = 16: none)
    __shared__ u32x4 s_sel[20];                           // s_sel[q + 4]: v_perm_b32 selectors that place a literal's bytes at block position q (-4..15)

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    if (tid < 17u) {
        u32x4 m;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) m[k] = tid <= 4u * k ? 0xFFFFFFFFu : (tid >= 4u * k + 4u ? 0u : 0xFFFFFFFFu << (8u * (tid - 4u * k)));
        s_tail[tid] = m;
    } else if (tid >= 32u && tid < 52u) {
        const int32_t q = int32_t(tid) - 36;             // block position of the literal's first byte
        u32x4 sel;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t w = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int32_t t = 4 * k + j - q;          // which literal byte lands on block byte 4k + j (the literal is 8 bytes {hi, lo}, bytes 5..7 zero)
                w |= ((t >= 0 && t < 8) ? uint32_t(t) : 0x0Cu) << (8 * j);     // selector 0x0C: constant zero
            }
            sel[k] = w;
        }
        s_sel[tid - 32u] = sel;
    }
    // literal bytes (<= 5, first byte lowest) placed at byte position q of a block
    auto place = [&](uint64_t lit, int32_t q) -> u32x4 {
        const u32x4 sel = s_sel[uint32_t(q + 4)];        // (callers are lanes with a block of the chunk: q is in -4..15)
        const uint32_t lo = uint32_t(lit), hi = uint32_t(lit >> 32);
        return u32x4{__builtin_amdgcn_perm(hi, lo, sel[0]), __builtin_amdgcn_perm(hi, lo, sel[1]), __builtin_amdgcn_perm(hi, lo, sel[2]), __builtin_amdgcn_perm(hi, lo, sel[3])};
    };
    // bytes >= ja of the block come from ld (ja in 1..15) when `take`
    auto tail = [&](u32x4 v, u32x4 ld, uint32_t ja, bool take) -> u32x4 {
        const u32x4 m = s_tail[take ? (ja & 15u) : 16u];
        v[0] = (v[0] & ~m[0]) | (ld[0] & m[0]); v[1] = (v[1] & ~m[1]) | (ld[1] & m[1]);
        v[2] = (v[2] & ~m[2]) | (ld[2] & m[2]); v[3] = (v[3] & ~m[3]) | (ld[3] & m[3]);
        return v;
    };
    // 16 bytes of a non-primary task for the block at `rel`: literal bytes placed at block position q, or a gather
    auto fetch = [&](uint64_t adj, int32_t rel, int32_t q) -> u32x4 {
        if (adj & ADJ_IMM) return place(adj & ADJ_LIT, q);
        return DW ? gather16_dw(adj + int64_t(rel)) : gather16(adj + int64_t(rel));
    };
    const uint16_t* const s_map = reinterpret_cast<const uint16_t*>(s_map32);
    const uint64_t dots16 = reinterpret_cast<uint64_t>(a.dots) + 32u;

    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, acc_lds = 0, acc_wait = 0, acc_rest = 0;
    if (DBG == 20) st0 = __builtin_amdgcn_s_memtime();
    for (uint32_t c = blockIdx.x; c < a.n_chunks; c += gridDim.x) {
        if (c != blockIdx.x) lds_barrier();            // LDS is reused by the next chunk
        const uint64_t tb = a.chunks[c].task_begin;
        const uint64_t dn = a.chunks[c].dst_n;
        const uint32_t n_hdr = uint32_t(dn >> 48) & CHUNK_N_MASK;
        if (filter == 2u && (dn & (CHUNK_LONG | CHUNK_DENSE | CHUNK_WAVE))) continue;   // long-run chunks belong to stitch4_kernel, dense ones to stitch_dense_kernel, wave chunks to stitchw_kernel
        const uint64_t dst = dn & ((1ull << 48) - 1);
        const uint32_t head = uint32_t(dst & 15ull);
        // a chunk table that points outside the descriptor array is refused, not followed
        const bool hdr_ok = n_hdr <= K && tb <= a.n_desc && n_hdr <= a.n_desc - tb;
        const uint32_t n = hdr_ok ? n_hdr : 0u;

        // ---- A: TPT consecutive descriptors per lane ----
        uint64_t adj[TPT];
        uint32_t len[TPT];
        uint32_t lsum = 0, lnz = 0;
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            adj[k] = dots16;
            len[k] = 0;
            const uint32_t i = tid * TPT + k;
            if (i < n) {
                const uint64_t d = a.desc[tb + i];
                len[k] = uint32_t(d >> 40) & ((1u << 22) - 1u);
                const uint32_t space = uint32_t(d >> 62);
                const uint64_t src = d & ((1ull << 40) - 1);
                if (space == SPACE_IMM) {
                    if (len[k] > IMM_MAX_BYTES) report(a.status, tb + i, STATUS_SRC_OOB);
                    else adj[k] = ADJ_IMM | src;
                } else {
                    const uint64_t limit = space == SPACE_PROTEOME ? a.src0_len : (space == SPACE_PAYLOAD ? a.src1_len : ~0ull);
                    if (src + len[k] > limit) {                  // never read out of bounds: task.rs would panic
                        report(a.status, tb + i, STATUS_SRC_OOB);
                    } else if (space != SPACE_FILL) {
                        adj[k] = reinterpret_cast<uint64_t>(space == SPACE_PROTEOME ? a.src0 : a.src1) + src;
                    }
                }
            }
            lsum += len[k];
            lnz += len[k] != 0u ? 1u : 0u;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s_map32[8u * tid + q] = 0u;
        if (DBG == 20) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); st1 = __builtin_amdgcn_s_memtime(); }
        const uint32_t incl = wave_incl_scan(lsum);
        const uint32_t nzincl = wave_incl_scan(lnz);
        if (lane == 63u) { s_w[0][wid] = incl; s_w[1][wid] = nzincl; }
        lds_barrier();

        // ---- B ----
        const uint32_t l0 = s_w[0][0], l1 = s_w[0][1], l2 = s_w[0][2], l3 = s_w[0][3];
        const uint32_t z0 = s_w[1][0], z1 = s_w[1][1], z2 = s_w[1][2], z3 = s_w[1][3];
        const uint32_t total = l0 + l1 + l2 + l3;
        const uint32_t nz = z0 + z1 + z2 + z3;
        uint32_t excl = incl - lsum + (wid > 0 ? l0 : 0u) + (wid > 1 ? l1 : 0u) + (wid > 2 ? l2 : 0u);
        uint32_t rank = nzincl - lnz + (wid > 0 ? z0 : 0u) + (wid > 1 ? z1 : 0u) + (wid > 2 ? z2 : 0u);
        const uint32_t nblk = total ? (head + total + 15u) >> 4 : 0u;
        const bool chunk_ok = hdr_ok && dst + total <= a.out_len && nblk <= 4096u && total <= DOTS_BYTES - 96u;
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            if (len[k] != 0u) {
                s_off[rank] = excl;
                s_adj[rank] = (adj[k] & ADJ_IMM) ? adj[k] : adj[k] - excl;
                if (rank >= 1u && chunk_ok) {
                    const uint32_t kmin = (excl + head + 15u) >> 4;    // first block starting at or after the task start
                    if (kmin < nblk) atomicAdd(&s_map32[kmin >> 1], 1u << (16u * (kmin & 1u)));
                }
                ++rank;
            }
            excl += len[k];
        }
        if (tid < 4u) { s_off[nz + tid] = total; s_adj[nz + tid] = dots16 - total; }   // sentinels past the last task
        lds_barrier();

        // ---- C: per-lane 16 two-byte counters -> in-lane prefix sums ----
        uint32_t y[8], pre[8];
        uint32_t tsum = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            y[q] = s_map32[8u * tid + q] * 0x00010001u;          // low half: first counter, high half: sum of both
            pre[q] = tsum;
            tsum += y[q] >> 16;
        }
        const uint32_t tincl = wave_incl_scan(tsum);
        if (lane == 63u) s_w[2][wid] = tincl;
        lds_barrier();

        // ---- D: add the lanes/waves before; ranks stay below 256*TPT < 65536 ----
        {
            const uint32_t m0 = s_w[2][0], m1 = s_w[2][1], m2 = s_w[2][2];
            const uint32_t mb = tincl - tsum + (wid > 0 ? m0 : 0u) + (wid > 1 ? m1 : 0u) + (wid > 2 ? m2 : 0u);
#pragma unroll
            for (int q = 0; q < 8; ++q) s_map32[8u * tid + q] = y[q] + (mb + pre[q]) * 0x00010001u;
        }
        lds_barrier();

        if (DBG == 20) st2 = __builtin_amdgcn_s_memtime();
        if (!chunk_ok) {                                      // never write out of bounds
            if (tid == 0) report(a.status, tb, STATUS_RES_OOB);
        } else {
            // ---- K2: wave `wid` takes the 64-block rows wid, wid+4, ... (a workgroup pass = 4 KiB of result) ----
            uint8_t* const out0 = a.out + (dst - head);
            const uint32_t nrow = (nblk + 63u) >> 6;
            for (uint32_t row = wid; row < (DBG == 4 ? 0u : nrow); row += 4u) {
                unsigned long long k0 = 0, k1 = 0, k2 = 0;
                if (DBG == 20) k0 = __builtin_amdgcn_s_memtime();
                const uint32_t b = (row << 6) + lane;
                const bool active = b < nblk;
                const int32_t rel = int32_t(b << 4) - int32_t(head);          // block start relative to dst
                const uint32_t hi = uint32_t(rel + 16) < total ? uint32_t(rel + 16) : total;
                uint32_t r = active ? uint32_t(s_map[b]) : 0u;                 // two-byte entries
                const uint32_t o0 = s_off[r], e0 = s_off[r + 1u], e1 = s_off[r + 2u], e2 = s_off[r + 3u];
                const uint64_t a0 = s_adj[r], a1 = s_adj[r + 1u], a2 = s_adj[r + 2u];
                // the sentinels make r+1, r+2 readable (dots) even when they are past the last task
                const bool need1 = DBG != 3 && active && e0 < hi, need2 = DBG != 3 && active && e1 < hi;
                const bool imm0 = DBG != 3 && active && (a0 & ADJ_IMM) != 0ull;
                u32x4 v;
                if (VAR == 1) {
                    // legacy: one byte-granular gather per overlapping task; lanes that need no second/third task stay masked off
                    u32x4 g1 = {0u, 0u, 0u, 0u}, g2 = {0u, 0u, 0u, 0u};
                    if (DBG == 20) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); k1 = __builtin_amdgcn_s_memtime(); }
                    v = g1;
                    if (DBG != 1) {
                        if (active) v = fetch(a0, rel, int32_t(o0) - rel);
                        if (need1) g1 = fetch(a1, rel, int32_t(e0) - rel);
                        if (need2) g2 = fetch(a2, rel, int32_t(e1) - rel);
                        if (DBG == 20) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); k2 = __builtin_amdgcn_s_memtime(); }
                    } else {
                        v = u32x4{uint32_t(a0), e0, r, hi};
                        g1 = u32x4{uint32_t(a1), e1, r, hi};
                        g2 = u32x4{uint32_t(a2), e2, r, hi};
                    }
                    v = tail(v, g1, uint32_t(int32_t(e0) - rel), need1);
                    v = tail(v, g2, uint32_t(int32_t(e1) - rel), need2);
                } else {
                    // tasks r+1 / r+2 that continue the primary stream (same source-minus-result offset) reuse its bytes
                    const bool same1 = need1 && !imm0 && a1 == a0;
                    const bool same2 = need2 && !imm0 && a2 == a0;
                    const uint64_t X = a0 + uint64_t(int64_t(rel));             // source address of the block's first byte
                    const uint32_t dlt = uint32_t(X) & 15u;
                    const uint64_t G = (active && !imm0) ? X - dlt : 0ull;      // its aligned 16-byte block (0: no primary load)
                    uint32_t pend = e0 < hi ? e0 : hi;                          // end of the bytes stream 0 supplies
                    if (same1) pend = e1 < hi ? e1 : hi;
                    if (same2) pend = e2 < hi ? e2 : hi;
                    const uint32_t pcount = uint32_t(int32_t(pend) - rel);
                    // does the next lane hold the following aligned block of the same stream?  (lane 63 and lanes
                    // next to an idle lane receive 0)
                    const uint64_t Gn = uint64_t(from_next_lane(0u, uint32_t(G))) | (uint64_t(from_next_lane(0u, uint32_t(G >> 32))) << 32);
                    const bool own_n = G != 0ull && dlt != 0u && pcount + dlt > 16u && Gn != G + 16ull;
                    u32x4 n_own = {0u, 0u, 0u, 0u}, g1 = {0u, 0u, 0u, 0u}, g2 = {0u, 0u, 0u, 0u};
                    v = n_own;
                    if (DBG == 20) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); k1 = __builtin_amdgcn_s_memtime(); }
                    if (DBG != 1) {
                        if (G != 0ull) v = load16_aligned(G);
                        if (own_n) n_own = load16_aligned(G + 16ull);
                        if (need1 && !same1) g1 = fetch(a1, rel, int32_t(e0) - rel);
                        if (need2 && !same2) g2 = fetch(a2, rel, int32_t(e1) - rel);
                        if (DBG == 20) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); k2 = __builtin_amdgcn_s_memtime(); }
                    } else {
                        v = u32x4{uint32_t(a0), e0, r, hi};
                        g1 = u32x4{uint32_t(a1), e1, r, hi};
                        g2 = u32x4{uint32_t(a2), e2, r, hi};
                    }
                    u32x4 nx;
                    nx[0] = from_next_lane(0u, v[0]); nx[1] = from_next_lane(0u, v[1]);
                    nx[2] = from_next_lane(0u, v[2]); nx[3] = from_next_lane(0u, v[3]);
                    if (own_n) nx = n_own;
                    const u32x4 p = imm0 ? place(a0 & ADJ_LIT, int32_t(o0) - rel) : funnel16(v, nx, dlt);
                    v = tail(p, g1, uint32_t(int32_t(e0) - rel), need1 && !same1);
                    v = tail(v, same2 ? p : g2, uint32_t(int32_t(e1) - rel), need2);
                }
                if (DBG != 3 && active && e2 < hi) {           // four or more tasks cut this block: three more per round,
                    uint32_t pos = e2;                         // their gathers in flight together
                    r += 3u;
                    while (pos < hi) {
                        const uint32_t o1 = s_off[r + 1u], o2 = s_off[r + 2u], o3 = s_off[r + 3u];
                        const uint64_t b0 = s_adj[r], b1 = s_adj[r + 1u], b2 = s_adj[r + 2u];
                        const bool n1 = o1 < hi, n2 = o2 < hi;
                        const u32x4 h0 = fetch(b0, rel, int32_t(pos) - rel);
                        u32x4 h1 = {0u, 0u, 0u, 0u}, h2 = {0u, 0u, 0u, 0u};
                        if (n1) h1 = fetch(b1, rel, int32_t(o1) - rel);
                        if (n2) h2 = fetch(b2, rel, int32_t(o2) - rel);
                        v = tail(v, h0, uint32_t(int32_t(pos) - rel), true);
                        v = tail(v, h1, uint32_t(int32_t(o1) - rel), n1);
                        v = tail(v, h2, uint32_t(int32_t(o2) - rel), n2);
                        pos = o3;                              // >= hi unless all three tasks ended inside the block
                        r += 3u;
                    }
                }
                if (active) {
                    uint8_t* o = out0 + (uint64_t(b) << 4);
                    if (DBG == 2) { if (v[0] == 0x12345678u && v[3] == 0x9abcdef0u) o[0] = 1; }
                    else if (rel >= 0 && uint32_t(rel) + 16u <= total) {
                        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(o));
                        else *reinterpret_cast<u32x4*>(o) = v;
                    } else {
                        // ragged first/last block of a chunk whose cut is not 16-byte aligned
                        const uint32_t ka = rel < 0 ? uint32_t(-rel) : 0u, kb = uint32_t(int32_t(hi) - rel);
#pragma unroll
                        for (uint32_t j = 0; j < 16u; ++j)
                            if (j >= ka && j < kb) o[j] = uint8_t(v[j >> 2] >> (8u * (j & 3u)));
                    }
                }
                if (DBG == 20) { const unsigned long long k3 = __builtin_amdgcn_s_memtime(); acc_lds += k1 - k0; acc_wait += k2 - k1; acc_rest += k3 - k2; }
            }
        }
        if (DBG == 20) {
            st3 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long st4 = __builtin_amdgcn_s_memtime();
            if (lane == 0u) {
                unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.out + ((a.out_len + 255ull) & ~255ull)) + (uint64_t(c) * 4u + wid) * 8u;
                dbg[0] = st0; dbg[1] = st1; dbg[2] = st2; dbg[3] = st3; dbg[4] = st4; dbg[5] = acc_lds; dbg[6] = acc_wait; dbg[7] = acc_rest;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// stitch_dense_kernel: the same contract as stitch_kernel for images of SHORT tasks (a few result bytes each: deep Task
// vectors, C5).  There a 16-byte result block is cut by three or four tasks and the per-block kernel pays a gather and a
// masked merge for each of them.  Here the work follows the tasks instead, one chunk per workgroup:
//   A  lane = 4 consecutive descriptors: decode, bounds-check, and at once the gather of every task's first 16 bytes (it does
//      not depend on the offsets); wave64 DPP scan of the lengths -> result offsets
//   B  every task ORs its first piece into an LDS image of the chunk's result (same 16-byte phase as the arena): the bytes
//      masked to the task's length, shifted to the destination's dword phase, up to five ds_or_b32 into the zeroed image --
//      tasks never overlap in the result, so the ORs of neighbouring tasks compose whatever their order and no byte-granular
//      LDS access is needed.  What a task has beyond 16 bytes goes on an LDS list of 16-byte pieces (ranked by a second scan);
//      the workgroup then takes the list with lane = piece, all gathers of a lane in flight together: one more memory
//      latency per chunk however long the tasks are.
//   C  the image leaves as aligned 16-byte non-temporal stores, 1 KiB per wave instruction; each block is zeroed as it goes.
// A chunk larger than the 12 KiB image is done in windows (B and C per window); the packer keeps the chunks of a dense image
// below it (1024 descriptors or 12 KiB, whichever comes first).
constexpr uint32_t DENSE_STAGE = 12288u;                   // bytes of the LDS image (CHUNK_BYTES_DENSE + 16)
constexpr uint32_t DENSE_PIECES = DENSE_STAGE / 16u + 8u;  // continuation pieces of one window

// up to n (1..16) bytes x, first byte lowest, OR-ed into the LDS image at byte offset o.  `lowmask[n]` = the low n bytes of a block
// (an LDS table: one ds_read_b128 instead of a dozen 64-bit shift / select instructions -- the kernel is VALU-bound).
__device__ __forceinline__ void dense_put(uint32_t* img, const u32x4* lowmask, uint32_t o, u32x4 x, uint32_t n)
{
    const u32x4 m = lowmask[n];
    x[0] &= m[0]; x[1] &= m[1]; x[2] &= m[2]; x[3] &= m[3];
    // five dwords from wb on; the data begin at byte bo = 1..4 of them (alignbyte shifts by 0..3 bytes only, so a dword-aligned
    // destination is taken as "shifted by four": dword 0 then holds nothing)
    const uint32_t wb = ((o + 3u) >> 2) - 1u, s2 = (0u - o) & 3u, end = o - 4u * wb + n;
    const uint32_t e0 = __builtin_amdgcn_alignbyte(x[0], 0u, s2);
    const uint32_t e1 = __builtin_amdgcn_alignbyte(x[1], x[0], s2);
    const uint32_t e2 = __builtin_amdgcn_alignbyte(x[2], x[1], s2);
    const uint32_t e3 = __builtin_amdgcn_alignbyte(x[3], x[2], s2);
    const uint32_t e4 = __builtin_amdgcn_alignbyte(0u, x[3], s2);
    if (s2 != 0u) atomicOr(&img[wb], e0);
    atomicOr(&img[wb + 1u], e1);
    if (end > 8u) atomicOr(&img[wb + 2u], e2);
    if (end > 12u) atomicOr(&img[wb + 3u], e3);
    if (end > 16u) atomicOr(&img[wb + 4u], e4);
}

// DBG != 0: timing-only ablations (results are wrong): 1 = no gathers, 2 = no stores, 3 = no LDS puts
// ONE: every chunk is a single window (the chunks a dense image flags; a larger one is refused) -- its own instance, because with
// the multi-window path in the same kernel the compiler keeps 95 VGPRs live instead of 58
// RIMG: the image is a rows image (sir_pack.hpp: CHUNK_CLIP on every chunk) -- the chunk starts on a 1 KiB row, skips the head of
// its first descriptor and clips its last one after its rows; an instance of its own (the host packer's images run the code they ran)
template <bool NT, bool DW, int DBG = 0, bool ONE = false, bool RIMG = false>
__global__ __launch_bounds__(256) void stitch_dense_kernel(const uint64_t* __restrict__ p_desc, const Chunk* __restrict__ p_chunks,
                                                              const uint8_t* __restrict__ p_src0, const uint8_t* __restrict__ p_src1,
                                                              uint8_t* __restrict__ p_out, unsigned long long* __restrict__ p_status,
                                                              const uint8_t* __restrict__ p_dots,
                                                              uint32_t n_chunks, uint64_t n_desc, uint64_t src0_len, uint64_t src1_len, uint64_t out_len,
                                                              uint32_t filter)
{
    struct { const uint64_t* desc; const Chunk* chunks; uint32_t n_chunks; uint64_t n_desc; const uint8_t* src0; uint64_t src0_len;
             const uint8_t* src1; uint64_t src1_len; uint8_t* out; uint64_t out_len; unsigned long long* status; const uint8_t* dots; }
        a{p_desc, p_chunks, n_chunks, n_desc, p_src0, src0_len, p_src1, src1_len, p_out, out_len, p_status, p_dots};
    constexpr int TPT = 4;
    constexpr uint32_t K = 256u * TPT;
    constexpr uint64_t OFF40 = (1ull << 40) - 1;
    __shared__ __attribute__((aligned(16))) uint32_t s_img[DENSE_STAGE / 4u + 8u];
    __shared__ uint64_t s_piece[DENSE_PIECES];              // space:2 | source offset:40 | (bytes - 1) << 42 | image offset << 46
    __shared__ uint32_t s_w[2][4];
    __shared__ u32x4 s_low[17];                             // s_low[j]: the low j bytes of a 16-byte block

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint32_t c = blockIdx.x;
    if (c >= a.n_chunks) return;
    if (tid < 17u) {
        u32x4 m;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) m[k] = tid >= 4u * k + 4u ? 0xFFFFFFFFu : (tid <= 4u * k ? 0u : (1u << (8u * (tid - 4u * k))) - 1u);
        s_low[tid] = m;
    }
    const uint64_t tb_raw = a.chunks[c].task_begin;
    const uint64_t dn = a.chunks[c].dst_n;
    if (dn & (CHUNK_LONG | CHUNK_WAVE)) return;            // long-run chunks belong to stitch4_kernel, wave chunks to stitchw_kernel
    if (filter == 3u && !(dn & CHUNK_DENSE)) return;       // (the other chunks of this image go to the per-block kernel)
    const uint64_t tb = RIMG ? tb_raw & TB_IDX_MASK : tb_raw;
    if (RIMG != ((dn & CHUNK_CLIP) != 0ull)) { if (tid == 0) report(a.status, tb, STATUS_RES_OOB); return; }      // (the launcher picked the wrong instance)
    const uint32_t hskip = RIMG ? uint32_t(tb_raw >> TB_IDX_BITS) & PIECE_MAX : 0u;
    const uint32_t tclip = RIMG ? uint32_t(tb_raw >> (TB_IDX_BITS + TB_SKIP_BITS)) : 0u;
    const bool fused = (dn & CHUNK_DENSE) != 0ull;         // the chunk may hold fused substitution descriptors
    const uint32_t n_hdr = uint32_t(dn >> 48) & CHUNK_N_MASK;
    const uint64_t dst = dn & ((1ull << 48) - 1);
    const uint32_t head = RIMG ? 0u : uint32_t(dst & 15ull);
    const bool hdr_ok = n_hdr <= K && tb <= a.n_desc && n_hdr <= a.n_desc - tb;
    const uint32_t n = hdr_ok ? n_hdr : 0u;

    // 16 bytes at offset `so` of a source space (descriptor layout: space in bits 40..41 here)
    auto piece = [&](uint64_t sw) -> u32x4 {
        const uint32_t space = uint32_t(sw >> 40) & 3u;
        const uint64_t so = sw & OFF40;
        if (space == SPACE_IMM) return u32x4{uint32_t(so), uint32_t(so >> 32), 0u, 0u};
        if (space == SPACE_FILL) return u32x4{0x2E2E2E2Eu, 0x2E2E2E2Eu, 0x2E2E2E2Eu, 0x2E2E2E2Eu};
        const uint64_t addr = reinterpret_cast<uint64_t>(space == SPACE_PROTEOME ? a.src0 : a.src1) + so;
        return DW ? gather16_dw(addr) : gather16(addr);
    };

    // ---- A: descriptors, and at once the first 16 bytes of every task (they do not depend on the offsets) ----
    // (later windows of a chunk larger than the image come back here with the descriptors in L2: nothing of a window stays in
    // registers while the next is worked on)
    u32x4 g[TPT];
    uint32_t len[TPT];
    uint64_t sw[TPT];                                       // space << 40 | source offset (the literal of an immediate task)
    uint32_t patch[TPT];                                    // fused substitution: bytes before the literal | literal << 12 | 1 << 20 [| position of a second literal << 21 | 1 << 28]
    auto load_tasks = [&](bool first_time) -> uint32_t {
        uint64_t d[TPT];
        uint32_t lsum = 0, bad = 0u;
#pragma unroll
        for (int k = 0; k < TPT; ++k) d[k] = tid * TPT + k < n ? a.desc[tb + tid * TPT + k] : 0ull;   // (0: an empty task)
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            // One branch-free decode for the three descriptor kinds (a wave runs every branch some lane takes: three separate paths cost
            // their sum).  Fused substitutions (sir_pack.hpp SNV3: bits 63..61 = 111; SNV5, two in a row: bits 63..60 = 1101) are ONE
            // reference run of len1 + 1 + len2 [+ 1 + len3] bytes of which one or two bytes are replaced afterwards.
            const uint32_t dlo = uint32_t(d[k]), dhi = uint32_t(d[k] >> 32);
            const bool is5 = (dhi >> 28) == 0xDu, is3 = (dhi >> 29) == 7u, fz = is5 || is3;
            const uint32_t f29 = __builtin_amdgcn_alignbit(dhi, dlo, 29);          // bits 29..60 of the descriptor
            const uint32_t l1 = f29 & (is5 ? 31u : 0xFFFu);
            const uint32_t l2 = is5 ? (f29 >> 5) & 31u : (f29 >> 12) & 0xFFFu;
            const uint32_t l3 = is5 ? (f29 >> 10) & 31u : 0u;
            const uint32_t b1 = (is5 ? f29 >> 15 : f29 >> 24) & 0xFFu, b2 = (f29 >> 23) & 0xFFu;
            uint32_t ln = fz ? l1 + 1u + l2 + (is5 ? 1u + l3 : 0u) : (dhi >> 8) & 0x3FFFFFu;
            // source bytes actually read: a fused run ends with its last non-empty copy (the literal may sit on the last residue)
            const uint32_t used = fz ? (l3 ? ln : (l2 ? l1 + 1u + l2 : l1)) : ln;
            const uint32_t space = fz ? SPACE_PROTEOME : dhi >> 30;
            const uint32_t so_lo = fz ? dlo & 0x1FFFFFFFu : dlo, so_hi = fz ? 0u : dhi & 0xFFu;
            const uint64_t so = (uint64_t(so_hi) << 32) | so_lo;
            const uint64_t limit = space == SPACE_PROTEOME ? a.src0_len : (space == SPACE_PAYLOAD ? a.src1_len : ~0ull);
            // never read out of bounds (task.rs would panic); a fused descriptor outside a dense image's chunk is refused too
            bool bd = so + used > limit || (space == SPACE_IMM && ln > IMM_MAX_BYTES) || (fz && !fused);
            // a rows image: the chunk's first descriptor begins in the chunk before -- its stream starts `hskip` bytes in -- and its last
            // one goes on into the next: `tclip` bytes less (a literal there is nobody's)
            const uint32_t hs = (RIMG && k == 0 && tid == 0u) ? hskip : 0u;
            const uint32_t tcl = (RIMG && tid * TPT + k + 1u == n) ? tclip : 0u;
            uint64_t so_h = so;
            uint32_t p1 = l1, p2 = l1 + 1u + l2;            // where the literals of a fused run sit
            bool lit1 = fz, lit2 = is5;
            if (RIMG) {
                if (hs + tcl != 0u && hs + tcl >= ln) bd = true;
                else {
                    ln -= hs + tcl;
                    if (k == 0) so_h = space == SPACE_IMM ? so >> (8u * hs) : (space == SPACE_FILL ? so : so + hs);
                    lit1 = lit1 && hs <= p1; lit2 = lit2 && hs <= p2;
                    p1 -= hs; p2 -= hs;
                    lit1 = lit1 && p1 < ln; lit2 = lit2 && p2 < ln;
                }
            }
            if (bd) { bad |= 1u << k; ln = 0u; }
            patch[k] = (fz && !bd) ? ((lit1 ? (p1 & 0xFFFu) | b1 << 12 | 1u << 20 : 0u) | (lit2 ? (p2 & 0x7Fu) << 21 | 1u << 28 : 0u)) : 0u;
            len[k] = ln;
            sw[k] = (uint64_t((is5 ? b2 << 16 : 0u) | space << 8 | uint32_t(so_h >> 32)) << 32) | uint32_t(so_h);   // (the second literal rides in bits 48..55)
            lsum += len[k];
        }
#pragma unroll
        for (int k = 0; k < TPT; ++k) g[k] = (len[k] && DBG != 1) ? piece(sw[k]) : u32x4{uint32_t(sw[k]), 0u, 0u, 0u};
        if (bad && first_time) report(a.status, tb + tid * TPT + uint32_t(__builtin_ctz(bad)), STATUS_SRC_OOB);   // task.rs would panic
        return lsum;
    };
    const uint32_t lsum = load_tasks(true);
    {   // zero the image: 3 x 16 B per lane, and the few dwords a put may touch past its end
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (uint32_t q = 0; q < DENSE_STAGE / 4096u; ++q) *reinterpret_cast<u32x4*>(&s_img[(q * 256u + tid) * 4u]) = z;
        if (tid < 2u) *reinterpret_cast<u32x4*>(&s_img[DENSE_STAGE / 4u + tid * 4u]) = z;
    }
    const uint32_t incl = wave_incl_scan(lsum);
    if (lane == 63u) s_w[0][wid] = incl;
    lds_barrier();
    const uint32_t l0 = s_w[0][0], l1 = s_w[0][1], l2 = s_w[0][2], l3 = s_w[0][3];
    const uint32_t total = l0 + l1 + l2 + l3;
    const uint32_t excl = incl - lsum + (wid > 0 ? l0 : 0u) + (wid > 1 ? l1 : 0u) + (wid > 2 ? l2 : 0u);
    if (!(hdr_ok && dst + total <= a.out_len)) {            // never write out of bounds
        if (tid == 0) report(a.status, tb, STATUS_RES_OOB);
        return;
    }
    const uint32_t span = head + total;                     // image bytes of the whole chunk (the image starts at the block-aligned dst - head)
    uint8_t* const out0 = a.out + (dst - head);
    const uint32_t off0 = head + excl;                      // image position of the lane's first task

    // `single`: the chunk is one window (every chunk of a dense image) -- nothing is clipped, and no immediate task has a second piece
    auto window = [&](const uint32_t w0_) {
        const uint32_t w0 = ONE ? 0u : w0_;
        const uint32_t w1 = ONE ? span : min(w0 + DENSE_STAGE, span);
        // ---- B: the first piece of every task; the pieces beyond it go on the list ----
        uint32_t off = off0, npc = 0;
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            const uint32_t b = ONE ? off : max(off, w0), e = ONE ? off + len[k] : min(off + len[k], w1);
            if (b < e) {
                uint32_t first = 0u;
                if (ONE || b == off) {                        // the task begins in this window: its first piece is in registers
                    first = min(e - b, 16u);
                    if (DBG != 3) dense_put(s_img, s_low, b - w0, g[k], first);
                    else if (g[k][0] == 0x12345678u) s_img[0] = 1u;
                }
                npc += (e - b - first + 15u) >> 4;
            }
            off += len[k];
        }
        const uint32_t pincl = wave_incl_scan(npc);
        if (lane == 63u) s_w[1][wid] = pincl;
        lds_barrier();
        const uint32_t p0 = s_w[1][0], p1 = s_w[1][1], p2 = s_w[1][2], p3 = s_w[1][3];
        const uint32_t n_pieces = p0 + p1 + p2 + p3;          // <= (w1 - w0) / 16 + 1 by construction
        if (n_pieces) {                                       // (uniform)
            uint32_t pi = pincl - npc + (wid > 0 ? p0 : 0u) + (wid > 1 ? p1 : 0u) + (wid > 2 ? p2 : 0u);
            off = off0;
#pragma unroll
            for (int k = 0; k < TPT; ++k) {
                const uint32_t b = ONE ? off : max(off, w0), e = ONE ? off + len[k] : min(off + len[k], w1);
                if (b < e) {
                    const uint32_t qs = b + ((ONE || b == off) ? min(e - b, 16u) : 0u);
                    const uint32_t space = uint32_t(sw[k] >> 40) & 3u;
                    if (qs < e) {                             // (not for most tasks)
                        // the wave runs this loop as often as its longest task has pieces: the record of a piece is the one before it
                        // plus a constant (source and image offset 16 bytes on), the last piece is shorter
                        const uint32_t skip = qs - off;       // bytes of the task before the first listed piece
                        const uint64_t so = (!ONE && space == SPACE_IMM) ? (sw[k] & OFF40) >> (8u * skip)             // (a literal cut by the window)
                                                               : (sw[k] & OFF40) + (space == SPACE_FILL ? 0ull : uint64_t(skip));
                        uint64_t rec = (sw[k] & (3ull << 40)) | so | (15ull << 42) | (uint64_t(qs - w0) << 46);
                        const uint64_t step = (space == SPACE_FILL ? 0ull : 16ull) | (16ull << 46);
                        uint32_t q = qs;
                        for (; q + 16u < e; q += 16u) { s_piece[pi++] = rec; rec += step; }
                        s_piece[pi++] = rec - (uint64_t(16u - (e - q)) << 42);
                    }
                }
                off += len[k];
            }
            lds_barrier();
            // ---- B': lane = piece, a lane's gathers in flight together ----
            for (uint32_t j0 = tid; j0 < n_pieces; j0 += 512u) {
                uint64_t r[2];
                u32x4 v[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint32_t j = j0 + 256u * uint32_t(u);
                    r[u] = j < n_pieces ? s_piece[j] : ~0ull;
                    v[u] = u32x4{0u, 0u, 0u, 0u};
                    if (r[u] != ~0ull && DBG != 1) v[u] = piece(r[u] & ((1ull << 42) - 1));
                }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    if (r[u] != ~0ull && DBG != 3) dense_put(s_img, s_low, uint32_t(r[u] >> 46), v[u], (uint32_t(r[u] >> 42) & 15u) + 1u);
            }
            lds_barrier();
        }
        if (fused) {                                          // (uniform) the literals of the fused substitutions replace the reference bytes under them
                                                              // (every put is in the image: the barrier after the scan, or the one that ends B')
            off = off0;
#pragma unroll
            for (int k = 0; k < TPT; ++k) {
                const uint32_t q = off + (patch[k] & 0xFFFu);
                if (((patch[k] >> 20) & 1u) && (ONE || (q >= w0 && q < w1))) reinterpret_cast<uint8_t*>(s_img)[q - w0] = uint8_t(patch[k] >> 12);
                const uint32_t q2 = off + ((patch[k] >> 21) & 0x7Fu);
                if ((patch[k] >> 28) && (ONE || (q2 >= w0 && q2 < w1))) reinterpret_cast<uint8_t*>(s_img)[q2 - w0] = uint8_t(sw[k] >> 48);
                off += len[k];
            }
            lds_barrier();
        }
        // ---- C: the window's blocks leave; the image is zero again behind them ----
        const uint32_t nblk = (w1 - w0 + 15u) >> 4;
        for (uint32_t b = tid; b < nblk; b += 256u) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(&s_img[b * 4u]);
            *reinterpret_cast<u32x4*>(&s_img[b * 4u]) = u32x4{0u, 0u, 0u, 0u};
            const uint32_t q = w0 + (b << 4);                 // image position of the block
            uint8_t* o = out0 + q;
            if (DBG == 2) { if (v[0] == 0x12345678u && v[3] == 0x9abcdef0u) o[0] = 1; }
            else if (q >= head && q + 16u <= span) {
                if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(o));
                else *reinterpret_cast<u32x4*>(o) = v;
            } else {                                          // ragged first/last block of a chunk whose cut is not 16-byte aligned
                const uint32_t ka = q < head ? head - q : 0u, kb = min(span - q, 16u);
#pragma unroll
                for (uint32_t j = 0; j < 16u; ++j)
                    if (j >= ka && j < kb) o[j] = uint8_t(v[j >> 2] >> (8u * (j & 3u)));
            }
        }
    };
    if (ONE) {
        if (span > DENSE_STAGE) { if (tid == 0) report(a.status, tb, STATUS_RES_OOB); return; }   // not a chunk of a dense image: refused, nothing written
        window(0u);
        return;
    }
    window(0u);
    for (uint32_t w0 = DENSE_STAGE; w0 < span; w0 += DENSE_STAGE) {   // (not for a dense chunk)
        lds_barrier();                                        // the image is reused
        (void)load_tasks(false);
        window(w0);
    }
}

// ---------------------------------------------------------------------------
// stitch4_kernel: the same contract as stitch_kernel, built around what limits it on MI355X (DESIGN.md section 4,
// profiles/r02_*): gfx950 counts loads and stores in ONE in-order counter (vmcnt), and under a saturated write stream a
// store is acknowledged thousands of cycles after it was issued -- so a load issued after a store waits for that store.
// The per-block kernel alternates gather and store per 1 KiB row and parks its waves ~90 % of the time.  Here
//   * EVERY load of a workgroup precedes its first store: descriptors, the gathers of the merge pass, then up to eight 1 KiB
//     rows of gathers per wave held in registers (4 VGPRs a row), and only then the stores, back to back;
//   * that is affordable because the bulk pass is lean: blocks cut by a task boundary are assembled once per TASK in a
//     compacted pass P (lane = task, every lane does merge work, byte masks from a 17-entry LDS table) and parked in an LDS
//     patch table; the bulk pass needs per 16-byte block one map read, one 16-byte task record, one byte-granular gather
//     of the covering task's stream (or the patch) -- about 13 VALU instructions per KiB row (the per-block kernel: ~100;
//     a SIMD issues one wave instruction in about four cycles, which makes ~100 per KiB the budget of a CU at the HBM
//     store ceiling, profiles/r02_copy_mix_ballast.json);
//   * set-up: ranks by ballot + mbcnt, one DPP scan, block map by packed 16-bit adds.
// Chunks of up to 256*TPT descriptors, TPT in {1, 2}, best with <= 32 KiB of result (2048 blocks = one round of eight rows per
// wave); larger chunks take a second round whose gathers wait for the first round's stores.  Dense images (TPT 4) stay on
// stitch_kernel.
// ---------------------------------------------------------------------------
struct __attribute__((aligned(16))) TaskRec {
    // Everything is kept in BLOCK SPACE: position p = (chunk's result offset & 15) + offset inside the chunk, so that 16-byte result
    // block b covers positions [16b, 16b + 16) and no signed arithmetic is needed.
    uint32_t adj_lo, adj_hi;   // source address minus the task's start position (immediate tasks: a readable dummy, their bytes sit in s_lit[rank])
    uint32_t end, off;         // positions [off & REC_OFF, end); REC_IMM in `off` marks an immediate task
};
constexpr uint32_t REC_IMM = 0x80000000u, REC_OFF = 0x7FFFFFFFu;
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint64_t rec_adj(const TaskRec& t) { return (uint64_t(t.adj_hi) << 32) | t.adj_lo; }

// bytes >= ja of v come from ld; mask = s_mask[ja] (ja = 16: nothing)
__device__ __forceinline__ u32x4 merge_tail(u32x4 v, u32x4 ld, u32x4 m)
{
    v[0] = (ld[0] & m[0]) | (v[0] & ~m[0]);
    v[1] = (ld[1] & m[1]) | (v[1] & ~m[1]);
    v[2] = (ld[2] & m[2]) | (v[2] & ~m[2]);
    v[3] = (ld[3] & m[3]) | (v[3] & ~m[3]);
    return v;
}

// 16 bytes of task `t` (rank r) for the block at position b16 (bytes before the task's start / after its end are whatever lies there)
__device__ __forceinline__ u32x4 rec_fetch(const TaskRec& t, const uint64_t* s_lit, uint32_t r, uint32_t b16)
{
    if (t.off & REC_IMM) return imm_block(s_lit[r], int32_t((t.off & REC_OFF) - b16));
    return gather16(rec_adj(t) + b16);
}

// The block at position b16 whose first byte lies in task r: that task's stream, overwritten from their start on by every task
// that begins before the block (or the chunk) ends.  The first three sources are fetched together; more take a loop.
__device__ __forceinline__ u32x4 assemble_block(const TaskRec* s_rec, const uint64_t* s_lit, const u32x4* s_mask, uint32_t r, uint32_t b16, uint32_t ptotal)
{
    const uint32_t hi = b16 + 16u < ptotal ? b16 + 16u : ptotal;
    const TaskRec t0 = s_rec[r], t1 = s_rec[r + 1u], t2 = s_rec[r + 2u];
    const bool need1 = t0.end < hi, need2 = need1 && t1.end < hi;
    u32x4 v = rec_fetch(t0, s_lit, r, b16), g1 = v, g2 = v;
    if (need1) g1 = rec_fetch(t1, s_lit, r + 1u, b16);
    if (need2) g2 = rec_fetch(t2, s_lit, r + 2u, b16);
    v = merge_tail(v, g1, s_mask[need1 ? (t1.off & REC_OFF) - b16 : 16u]);
    v = merge_tail(v, g2, s_mask[need2 ? (t2.off & REC_OFF) - b16 : 16u]);
    if (need2 && t2.end < hi) {                               // four or more tasks in this block
        r += 2u;
        TaskRec t = t2;
        while (t.end < hi) {
            t = s_rec[++r];
            const u32x4 g = rec_fetch(t, s_lit, r, b16);
            v = merge_tail(v, g, s_mask[(t.off & REC_OFF) - b16]);
        }
    }
    return v;
}

// assemble_block in two halves, so that the gathers of many blocks are in flight together: `issue` starts the fetches of
// the first three sources, `finish` merges them (and loops over a fourth and later source).
struct BlockFetch { u32x4 v, g1, g2; uint32_t ja1, ja2, next; };   // ja: position inside the block where source 1 / 2 starts (16: unused); next: rank to go on with (0: done)
__device__ __forceinline__ BlockFetch block_issue(const TaskRec* s_rec, const uint64_t* s_lit, uint32_t r, uint32_t b16, uint32_t ptotal)
{
    const uint32_t hi = b16 + 16u < ptotal ? b16 + 16u : ptotal;
    const TaskRec t0 = s_rec[r], t1 = s_rec[r + 1u], t2 = s_rec[r + 2u];
    const bool need1 = t0.end < hi, need2 = need1 && t1.end < hi;
    // the reference going on after a substituted residue is the same stream as before it: its bytes are already in `v`
    const bool same2 = need2 && !((t0.off | t2.off) & REC_IMM) && t2.adj_lo == t0.adj_lo && t2.adj_hi == t0.adj_hi;
    BlockFetch f;
    f.v = rec_fetch(t0, s_lit, r, b16); f.g1 = f.v; f.g2 = f.v;
    if (need1) f.g1 = rec_fetch(t1, s_lit, r + 1u, b16);
    if (need2 && !same2) f.g2 = rec_fetch(t2, s_lit, r + 2u, b16);
    f.ja1 = need1 ? (t1.off & REC_OFF) - b16 : 16u;
    f.ja2 = need2 ? (t2.off & REC_OFF) - b16 : 16u;
    f.next = (need2 && t2.end < hi) ? r + 2u : 0u;
    return f;
}
__device__ __forceinline__ u32x4 block_finish(const BlockFetch& f, const TaskRec* s_rec, const uint64_t* s_lit, const u32x4* s_mask, uint32_t b16, uint32_t ptotal)
{
    u32x4 v = merge_tail(f.v, f.g1, s_mask[f.ja1]);
    v = merge_tail(v, f.g2, s_mask[f.ja2]);
    if (f.next) {                                             // four or more tasks in this block
        const uint32_t hi = b16 + 16u < ptotal ? b16 + 16u : ptotal;
        uint32_t r = f.next;
        TaskRec t = s_rec[r];
        while (t.end < hi) {
            t = s_rec[++r];
            const u32x4 g = rec_fetch(t, s_lit, r, b16);
            v = merge_tail(v, g, s_mask[(t.off & REC_OFF) - b16]);
        }
    }
    return v;
}

// An edge block of a chunk whose cut is not 16-byte aligned: only the chunk's own bytes [head, ptotal) are written.
__device__ __forceinline__ void ragged_block(const TaskRec* s_rec, const uint64_t* s_lit, const u32x4* s_mask, uint32_t r, uint32_t b16, uint32_t head,
                                             uint32_t ptotal, uint8_t* op)
{
    const u32x4 o = assemble_block(s_rec, s_lit, s_mask, r, b16, ptotal);
    const uint32_t ka = b16 < head ? head - b16 : 0u, kb = (b16 + 16u < ptotal ? b16 + 16u : ptotal) - b16;
    for (uint32_t q = ka; q < kb; ++q) op[q] = uint8_t(o[q >> 2] >> (8u * (q & 3u)));
}

template <int TPT, bool NT, int DBG = 0, int ROWS = 8, uint32_t WIN = 0>
__global__ __launch_bounds__(256) void stitch4_kernel(const uint64_t* __restrict__ p_desc, const Chunk* __restrict__ p_chunks,
                                                      const uint8_t* __restrict__ p_src0, const uint8_t* __restrict__ p_src1,
                                                      uint8_t* __restrict__ p_out, unsigned long long* __restrict__ p_status,
                                                      const uint8_t* __restrict__ p_dots,
                                                      uint32_t n_chunks, uint64_t n_desc, uint64_t src0_len, uint64_t src1_len, uint64_t out_len,
                                                      uint32_t filter)
{
    constexpr uint32_t K = 256u * TPT;
    constexpr uint32_t R = ROWS;                            // rows per wave and pass in the bulk phase: 8 x 4 waves x 1 KiB = 32 KiB
    __shared__ __attribute__((aligned(16))) uint32_t s_map32[2048 + 8];     // 4096 two-byte block->rank entries
    __shared__ TaskRec s_rec[K + 8];
    __shared__ u32x4 s_patch[K + 8];
    __shared__ uint64_t s_lit[K + 8];                                        // bytes of immediate tasks, by rank
    __shared__ u32x4 s_mask[17];
    __shared__ __attribute__((aligned(16))) uint32_t s_wt[TPT][4];           // per wave and round: bytes (low 20 bits) | non-empty tasks
    __shared__ __attribute__((aligned(16))) uint32_t s_w[2][4];              // [0] block-map counts per wave, [1][0] bad-descriptor flag
    // WIN != 0 (variant 7, the "LDS-staged reference tile" of the design brief): the span of the proteome that the chunk's reference
    // copies read is brought into LDS with wave-contiguous 16-byte direct loads (global_load_lds_dwordx4: 1 KiB per wave instruction,
    // every line touched once, no VGPRs) and the bulk phase assembles its blocks from LDS (two aligned 16-byte reads + a funnel shift)
    // instead of one unaligned gather each.  Chunks whose span does not fit keep the gathers.
    __shared__ __attribute__((aligned(16))) uint8_t s_win[WIN ? WIN + 32u : 16u];
    __shared__ uint32_t s_span[2];                                           // [0] first, [1] one past the last 16-byte block of that span

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint16_t* const s_map = reinterpret_cast<const uint16_t*>(s_map32);
    const uint64_t dots16 = reinterpret_cast<uint64_t>(p_dots) + 32u;

    {                                                                       // one chunk per workgroup: a fresh wave has no store in flight
        const uint32_t c = blockIdx.x;
        if (c >= n_chunks) return;
        unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q7 = 0, q8 = 0;
        if (DBG == 20) q0 = __builtin_amdgcn_s_memtime();
        const uint64_t tb = p_chunks[c].task_begin, dn = p_chunks[c].dst_n;
        const uint32_t n_hdr = uint32_t(dn >> 48) & CHUNK_N_MASK;
        if (filter == 1u && !(dn & CHUNK_LONG)) return;    // the other chunks belong to the per-block stitch_kernel
        const uint64_t dst = dn & ((1ull << 48) - 1);
        const uint32_t head = uint32_t(dst) & 15u;
        // a chunk table that points outside the descriptor array is refused, not followed
        const bool hdr_ok = n_hdr <= K && tb <= n_desc && n_hdr <= n_desc - tb;
        const uint32_t n = hdr_ok ? n_hdr : 0u;
        uint64_t d[TPT];
#pragma unroll
        for (int k = 0; k < TPT; ++k) { const uint32_t i = tid + 256u * uint32_t(k); d[k] = i < n ? p_desc[tb + i] : 0ull; }
        if (tid < 17u) {                                        // s_mask[j]: bytes >= j of a block
            u32x4 m;
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) m[k] = tid <= 4u * k ? 0xFFFFFFFFu : (tid >= 4u * k + 4u ? 0u : 0xFFFFFFFFu << (8u * (tid - 4u * k)));
            s_mask[tid] = m;
        }
        if (tid == 0u) { s_w[1][0] = 0u; s_span[0] = ~0u; s_span[1] = 0u; }
        {
            u32x4 z = {0u, 0u, 0u, 0u};
            reinterpret_cast<u32x4*>(s_map32)[2u * tid] = z;
            reinterpret_cast<u32x4*>(s_map32)[2u * tid + 1u] = z;
        }
        lds_barrier();                                          // (the bad flag is cleared before anybody can raise it)

        // ---- A: lane `tid` decodes descriptors tid, tid + 256, ... (so ranks follow lane order within each round).  A descriptor
        //      is one task, or -- the fused substitution -- up to three: reference copy, one literal byte, reference copy. ----
        uint64_t adr[TPT];                                                  // source address of the (first) copy
        uint32_t part0[TPT], part2[TPT], lit_lo[TPT], lit_hi[TPT];         // bytes of the first / third task (the second is 1 byte when fused)
        bool imm[TPT], snv[TPT];
        uint32_t pk[TPT];                                                   // bytes (low 20 bits) | tasks
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            const uint32_t i = tid + 256u * uint32_t(k);
            const uint32_t dlo = uint32_t(d[k]), dhi = uint32_t(d[k] >> 32);
            const uint32_t space = dhi >> 30;
            snv[k] = i < n && (dhi >> 29) == 7u;
            uint32_t len = i < n ? (dhi >> 8) & 0x3FFFFFu : 0u;
            uint64_t src = (uint64_t(dhi & 0xFFu) << 32) | dlo;
            part2[k] = 0u;
            if (snv[k]) {                                                   // src 0..28, len1 29..40, len2 41..52, byte 53..60
                src = dlo & 0x1FFFFFFFu;
                len = (dlo >> 29) | ((dhi & 0x1FFu) << 3);
                part2[k] = (dhi >> 9) & 0xFFFu;
            }
            const uint32_t bytes = snv[k] ? len + 1u + part2[k] : len;
            uint64_t a = dots16;
            bool bad = bytes > CHUNK_BYTES;
            imm[k] = !snv[k] && space == SPACE_IMM;
            lit_lo[k] = snv[k] ? (dhi >> 21) & 0xFFu : dlo;
            lit_hi[k] = snv[k] ? 0u : dhi & 0xFFu;
            if (imm[k]) {
                bad = bad || len > IMM_MAX_BYTES;
            } else if (snv[k] || space != SPACE_FILL) {
                const bool ref = snv[k] || space == SPACE_PROTEOME;
                bad = bad || src + bytes > (ref ? src0_len : src1_len);      // never read out of bounds: task.rs would panic
                a = reinterpret_cast<uint64_t>(ref ? p_src0 : p_src1) + src;
                if (WIN && ref && !bad && bytes != 0u) {
                    atomicMin(&s_span[0], uint32_t(src >> 4));
                    atomicMax(&s_span[1], uint32_t((src + bytes + 15u) >> 4));
                }
            }
            if (bad && bytes != 0u) {                                       // reported, and the chunk is not executed
                report(p_status, tb + i, STATUS_SRC_OOB);
                atomicOr(&s_w[1][0], 1u);
                len = 0u; part2[k] = 0u; snv[k] = false;
            }
            adr[k] = a;
            part0[k] = len;
            const uint32_t total_k = snv[k] ? len + 1u + part2[k] : len;
            const uint32_t tasks_k = snv[k] ? (len ? 1u : 0u) + 1u + (part2[k] ? 1u : 0u) : (len ? 1u : 0u);
            pk[k] = total_k | (tasks_k << 20);
        }
        // result offsets and ranks among the non-empty tasks: ONE DPP scan of (bytes | tasks << 20).  With TPT > 1 the descriptors
        // of round k all precede those of round k + 1, so the rounds are scanned one after the other.
        uint32_t incl[TPT];
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            incl[k] = wave_incl_scan(pk[k]);
            if (lane == 63u) s_wt[k][wid] = incl[k];
        }
        if (DBG == 20) q1 = __builtin_amdgcn_s_memtime();
        lds_barrier();

        uint64_t win_lo = 0ull, win_hi = 0ull;                               // addresses the LDS window covers (none: empty)
        if (WIN) {
            const uint32_t b0 = s_span[0], b1 = s_span[1];
            if (b1 > b0 && (b1 - b0) * 16u <= WIN) {
                const uint8_t* g = p_src0 + (uint64_t(b0) << 4);
                win_lo = reinterpret_cast<uint64_t>(g); win_hi = win_lo + (uint64_t(b1 - b0) << 4);
                for (uint32_t off = tid * 16u; off < (b1 - b0) * 16u; off += 4096u)        // (whole waves: the LDS side is wave base + lane * 16)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + off),
                                                     (__attribute__((address_space(3))) void*)(s_win + (off & ~1023u)), 16, 0, 0);
            }
        }
        // ---- B: compact the non-empty tasks by rank; mark the first block starting inside or after each ----
        const bool chunk_bad = s_w[1][0] != 0u;
        uint32_t total_pk = 0u;                                             // packed totals of the chunk
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            const u32x4 wt = *reinterpret_cast<const u32x4*>(&s_wt[k][0]);
#pragma unroll
            for (uint32_t w = 0; w < 4u; ++w) total_pk += wt[w];
        }
        const uint32_t total = total_pk & 0xFFFFFu;
        const uint32_t ptotal = head + total;                               // end of the chunk in block space
        const uint32_t nz = total_pk >> 20;
        const uint32_t nblk = total ? (ptotal + 15u) >> 4 : 0u;
        const bool chunk_ok = hdr_ok && !chunk_bad && dst + total <= out_len && nblk <= 4096u && total <= CHUNK_BYTES && nz <= K;
        {
            uint32_t before = 0u;                                           // packed totals of the rounds before round k
#pragma unroll
            for (int k = 0; k < TPT; ++k) {
                const u32x4 wt = *reinterpret_cast<const u32x4*>(&s_wt[k][0]);
                const uint32_t base_pk = before + (wid > 0u ? wt[0] : 0u) + (wid > 1u ? wt[1] : 0u) + (wid > 2u ? wt[2] : 0u);
                before += wt[0] + wt[1] + wt[2] + wt[3];
                const uint32_t ex = incl[k] - pk[k] + base_pk;
                uint32_t pos = head + (ex & 0xFFFFFu);                      // the descriptor's start in block space
                uint32_t rank = ex >> 20;
                if (!chunk_ok) continue;
                // its tasks, in order: (part0 from adr) [one literal byte] [part2 from adr + part0 + 1]
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const uint32_t plen = q == 0 ? part0[k] : (q == 1 ? (snv[k] ? 1u : 0u) : part2[k]);
                    if (plen != 0u) {
                        const bool lit = q == 1 || (q == 0 && imm[k]);
                        const uint64_t a = (lit ? dots16 : (q == 2 ? adr[k] + part0[k] + 1u : adr[k])) - pos;
                        s_rec[rank] = TaskRec{uint32_t(a), uint32_t(a >> 32), pos + plen, lit ? (pos | REC_IMM) : pos};
                        if (lit) s_lit[rank] = (uint64_t(lit_hi[k]) << 32) | lit_lo[k];
                        const uint32_t kmin = (pos + 15u) >> 4;            // first block starting at or after the task start
                        if (rank >= 1u && kmin < nblk) atomicAdd(&s_map32[kmin >> 1], (kmin & 1u) ? 0x10000u : 1u);
                        ++rank;
                        pos += plen;
                    }
                }
            }
        }
        // sentinels past the last task (dots; their ends lie beyond every block of the chunk)
        if (tid < 4u && chunk_ok) s_rec[nz + tid] = TaskRec{uint32_t(dots16 - ptotal), uint32_t((dots16 - ptotal) >> 32), 0x40000000u, ptotal};
        lds_barrier();

        // ---- C/D: block map = inclusive prefix sum of the marks, 16 two-byte counters per lane, packed 16-bit adds ----
        const u32x4 x0 = reinterpret_cast<const u32x4*>(s_map32)[2u * tid], x1 = reinterpret_cast<const u32x4*>(s_map32)[2u * tid + 1u];
        uint32_t y[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) { y[q] = x0[q] + (x0[q] << 16); y[4 + q] = x1[q] + (x1[q] << 16); }   // (c0, c0 + c1)
        uint32_t tsum = 0u;
#pragma unroll
        for (int q = 0; q < 8; ++q) tsum += y[q] >> 16;
        const uint32_t tincl = wave_incl_scan(tsum);
        if (lane == 63u) s_w[0][wid] = tincl;

        // ---- P (the map is not needed): lane = task.  Task t owns the block it starts in when it is the first task to start
        //      there (at a non-zero offset): it assembles that block from task t-1 (which covers the block's first byte) on; its
        //      gathers fly while the map is finished. ----
        if (DBG == 20) q2 = __builtin_amdgcn_s_memtime();
        BlockFetch pf[TPT];
        uint32_t pb16[TPT];
        bool owner[TPT];
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            const uint32_t t = tid + 256u * uint32_t(k);
            owner[k] = false;
            pb16[k] = 0u;
            if (chunk_ok && t >= 1u && t < nz) {
                const uint32_t s = s_rec[t].off & REC_OFF, b16 = s & ~15u;
                owner[k] = s != b16 && (b16 == 0u ? t == 1u : (s_rec[t - 1u].off & REC_OFF) <= b16);
                pb16[k] = b16;
                if (owner[k]) pf[k] = block_issue(s_rec, s_lit, t - 1u, b16, ptotal);
            }
        }
        lds_barrier();
        {
            const u32x4 mt = *reinterpret_cast<const u32x4*>(&s_w[0][0]);
            uint32_t run = tincl - tsum + (wid > 0u ? mt[0] : 0u) + (wid > 1u ? mt[1] : 0u) + (wid > 2u ? mt[2] : 0u);
            run |= run << 16;                                               // the running rank in both halves
            u32x4 o0, o1;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const u16x2 yy = __builtin_bit_cast(u16x2, y[q]);
                const u16x2 rr = __builtin_bit_cast(u16x2, run);
                const uint32_t o = __builtin_bit_cast(uint32_t, u16x2(yy + rr));
                if (q < 4) o0[q] = o; else o1[q - 4] = o;
                run = __builtin_bit_cast(uint32_t, u16x2(rr + yy.yy));      // += c0 + c1 in both halves
            }
            reinterpret_cast<u32x4*>(s_map32)[2u * tid] = o0;
            reinterpret_cast<u32x4*>(s_map32)[2u * tid + 1u] = o1;
        }
        // P, second half: merge and park the cut blocks
#pragma unroll
        for (int k = 0; k < TPT; ++k)
            if (owner[k]) s_patch[tid + 256u * uint32_t(k)] = block_finish(pf[k], s_rec, s_lit, s_mask, pb16[k], ptotal);
        if (DBG == 20) q3 = __builtin_amdgcn_s_memtime();
        if (WIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this wave's direct-to-LDS loads have landed (lds_barrier waits for LDS operations only)
        lds_barrier();
        if (DBG == 20) q4 = __builtin_amdgcn_s_memtime();

        if (!chunk_ok) {                                      // never write out of bounds
            if (tid == 0u) report(p_status, tb, STATUS_RES_OOB);
        } else if (DBG != 4) {
            // ---- bulk: per pass wave `wid` takes the 64-block rows wid, wid+4, ..., wid+28 of the pass's 32: R gathers, then R stores ----
            uint8_t* const out0 = p_out + (dst - head);                     // 16-byte aligned
            const uint32_t npass = (nblk + 256u * R - 1u) / (256u * R);
#pragma unroll 1
            for (uint32_t pass = 0; pass < npass; ++pass) {
                // position of the wave's first row in this pass (readfirstlane: a scalar, and opaque enough that eight sets of per-row
                // induction registers are not hoisted)
                const uint32_t pbase = __builtin_amdgcn_readfirstlane(pass * (4096u * R) + (wid << 10));
                const uint32_t p0 = pbase + (lane << 4);
                // look-ups first (LDS only), then the R gathers back to back: issued one by one, each would queue separately behind
                // the other waves' stores in the CU's memory pipeline
                uint64_t X[R];
#pragma unroll
                for (uint32_t j = 0; j < R; ++j) {
                    const uint32_t b16 = p0 + 4096u * j;
                    const uint32_t r = b16 < ptotal ? uint32_t(s_map[b16 >> 4]) : nz;   // idle lanes look at a sentinel (dots)
                    const TaskRec tr = s_rec[r];
                    // a task ends inside the block: the block the next task parked is taken instead (address 0 | patch index)
                    X[j] = tr.end <= b16 + 15u ? uint64_t(r + 1u) : rec_adj(tr) + b16;
                }
                u32x4 v[R];
#pragma unroll
                for (uint32_t j = 0; j < R; ++j) {
                    // (if / else, not a select: gather and patch land in the same registers, lanes disjoint)
                    if (uint32_t(X[j] >> 32) == 0u) v[j] = s_patch[uint32_t(X[j])];
                    else if (WIN && X[j] >= win_lo && X[j] + 16u <= win_hi + 16u && X[j] < win_hi) {
                        const uint32_t o = uint32_t(X[j] - win_lo);            // byte offset in the LDS window
                        const u32x4* w16 = reinterpret_cast<const u32x4*>(s_win);
                        v[j] = funnel16(w16[o >> 4], w16[(o >> 4) + 1u], o & 15u);
                    }
                    else v[j] = (DBG == 1) ? u32x4{uint32_t(X[j]), 0u, 0u, 0u} : gather16(X[j]);
                }
                if (DBG == 20) q7 = __builtin_amdgcn_s_memtime();
                if (DBG == 20) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); q8 = __builtin_amdgcn_s_memtime(); }
#pragma unroll
                for (uint32_t j = 0; j < R; ++j) {
                    const uint32_t b16 = p0 + 4096u * j;
                    if (b16 >= head && b16 + 16u <= ptotal) {               // whole blocks of the chunk; ragged edge blocks are written below
                        uint8_t* op = out0 + b16;
                        if (DBG == 2) { if (v[j][0] == 0x12345678u && v[j][3] == 0x9abcdef0u) op[0] = 1; }
                        else if (NT) __builtin_nontemporal_store(v[j], reinterpret_cast<u32x4*>(op));
                        else *reinterpret_cast<u32x4*>(op) = v[j];
                    }
                }
            }
            // ragged first / last block of a chunk whose cut is not 16-byte aligned (rare): one lane each, byte stores; these loads
            // do queue behind the two waves' stores
            if (DBG != 2 && (tid == 0u || tid == 64u)) {
                const uint32_t b16 = tid == 0u ? 0u : (nblk - 1u) << 4;
                if ((b16 < head || b16 + 16u > ptotal) && (tid == 0u || b16 != 0u))
                    ragged_block(s_rec, s_lit, s_mask, uint32_t(s_map[b16 >> 4]), b16, head, ptotal, out0 + b16);
            }
        }
        if (DBG == 20) {
            q5 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long q6 = __builtin_amdgcn_s_memtime();
            if (lane == 0u) {
                unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p_out + ((out_len + 255ull) & ~255ull)) + (uint64_t(c) * 4u + wid) * 8u;
                dbg[0] = q0; dbg[1] = q1; dbg[2] = q2; dbg[3] = q3; dbg[4] = q4; dbg[5] = q5; dbg[6] = q6; dbg[7] = ((q7 - q4) << 32) | ((q8 - q7) & 0xFFFFFFFFull);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// ordered_kernel: Task vectors whose result ranges overlap or go backwards.
// The CPU engine runs tasks one after the other (gir.rs:233), so a later task
// overwrites an earlier one; reproduce that order with one workgroup and a
// barrier per task.  `esize` scales element offsets to bytes (4 for Rust chars).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ordered_kernel(OrderedArgs a)
{
    for (uint64_t i = 0; i < a.n_tasks; ++i) {
        const uint64_t len = a.length[i] * a.esize;
        const uint64_t src = a.start_pos[i] * a.esize;
        const uint64_t dst = a.start_pos_res[i] * a.esize;
        const uint8_t* s = (a.code[i] == 0 ? a.ref : a.alt) + src;    // task.rs:42-49: any non-zero code reads the alt tape
        uint8_t* o = a.res + dst;
        for (uint64_t k = threadIdx.x; k < len; k += blockDim.x) o[k] = s[k];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// validate_kernel (DEBUG_GPU): rows are the SoA arrays of gir.rs:283-299.
// status word = min over failing rows of (row << 8 | reason).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void validate_kernel(ValidateArgs a)
{
    const uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    uint32_t reason = 0;
    if (i < a.n_tasks) {
        const uint32_t code = a.code[i];
        const uint64_t sp = a.start_pos[i], ln = a.length[i], sr = a.start_pos_res[i];
        const uint64_t n_src = code == 0 ? a.n_ref : a.n_alt;
        if (code > 1u) reason = STATUS_BAD_CODE;                                   // haplotype_instruction.rs:154
        else if (sr + ln > a.n_res || sr + ln < sr) reason = STATUS_RES_OOB;       // task.rs:43/47 (destination slice)
        else if (sp + ln > n_src || sp + ln < sp) reason = STATUS_SRC_OOB;         // task.rs:43/47 (source slice)
        else if (i >= 1 && sr != a.start_pos_res[i - 1] + a.length[i - 1]) reason = STATUS_NOT_CONTIGUOUS;  // gir.rs:208
    }
    const unsigned long long bad = __ballot(reason != 0);
    if (bad) {
        const uint32_t first = uint32_t(__ffsll((long long)bad)) - 1u;
        if ((threadIdx.x & 63u) == first) report(a.status, i, reason);
    }
}

// ---------------------------------------------------------------------------
// digest_kernel: digest[h] = sum_i (byte_i + 1) * 2^(8 * (i mod 8)) * splitmix64(i div 8), i relative to the haplotype start
// (mod 2^64) -- word by word: sum_k splitmix64(k) * (little-endian word k + 0x01..01 over the bytes that exist).  The
// definition is part of the C ABI contract (include/vcf2prot_hip.h: v2p_batch_digests) so any checker can recompute it.  Round 5: until then one splitmix64 per BYTE and a 15-step binary search per 16-byte
// block -- 60.9 ms for the north star's 36 GB, eight times an execute.  Now a wave walks 64 KiB of the arena, knows the
// haplotype it is in (one scalar search when it leaves it), and a lane's 16 bytes are the pieces of two or three words:
// two or three multipliers, whichever way the haplotype's first byte sits against the arena's 16-byte blocks.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

constexpr uint32_t DIGEST_ITER = 64;          // 1 KiB steps of a wave: it walks 64 KiB
constexpr uint64_t DIGEST_ONES = 0x0101010101010101ull;

// bytes [p, pe) of the arena, one by one, starting in haplotype h (the general form: a block that is cut by a haplotype boundary or
// by the arena's end, or an arena that is not 16-byte aligned)
__device__ __forceinline__ void digest_bytes(const DigestArgs& a, uint64_t p, uint64_t pe, uint64_t h)
{
    uint64_t sum = 0;
    for (uint64_t q = p; q < pe; ++q) {
        if (a.hap_begin[h + 1] <= q) {
            if (sum) atomicAdd(reinterpret_cast<unsigned long long*>(&a.digest[h]), (unsigned long long)sum);
            sum = 0;
            while (a.hap_begin[h + 1] <= q) ++h;
        }
        const uint64_t i = q - a.hap_begin[h];
        sum += ((uint64_t(a.out[q]) + 1ull) << (8u * (uint32_t(i) & 7u))) * mix64(i >> 3);
    }
    if (sum) atomicAdd(reinterpret_cast<unsigned long long*>(&a.digest[h]), (unsigned long long)sum);
}

__global__ __launch_bounds__(256) void digest_kernel(DigestArgs a)
{
    const uint64_t total = a.hap_begin[a.n_haps];
    const uint64_t nblk = (total + 15) >> 4;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = uint64_t(blockIdx.x) * 4u + (threadIdx.x >> 6);
    const bool aligned = (reinterpret_cast<uint64_t>(a.out) & 15u) == 0u;
    uint64_t h_cur = 0, hb = 0, he = 0;                   // (wave-uniform) the haplotype the wave is in: bytes [hb, he) of the arena
    bool known = false;
    uint64_t acc = 0;
    auto flush = [&]() {
        if (!known) return;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0u && acc) atomicAdd(reinterpret_cast<unsigned long long*>(&a.digest[h_cur]), (unsigned long long)acc);
        acc = 0;
    };
    for (uint32_t it = 0; it < DIGEST_ITER; ++it) {
        const uint64_t bw = (wave * DIGEST_ITER + it) * 64u;            // the wave's first 16-byte block of this step
        if (bw >= nblk) break;
        const uint64_t pw = bw << 4, pe_w = pw + 1024u < total ? pw + 1024u : total;
        if (!known || pw < hb || pe_w > he) {
            flush();
            // haplotype of byte pw: the last h with hap_begin[h] <= pw (empty haplotypes share an offset) -- pw is wave-uniform: scalar loads
            uint64_t lo = 0, hi = a.n_haps;
            while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (a.hap_begin[mid] <= pw) lo = mid; else hi = mid; }
            h_cur = lo; hb = a.hap_begin[lo]; he = a.hap_begin[lo + 1];
            known = pe_w <= he;
        }
        const uint64_t p = pw + (uint64_t(lane) << 4);
        if (!known) {
            // a haplotype ends inside this KiB: every lane finds its own
            if (p < total) {
                uint64_t lo = h_cur, hi = a.n_haps;
                while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (a.hap_begin[mid] <= p) lo = mid; else hi = mid; }
                digest_bytes(a, p, p + 16u < total ? p + 16u : total, lo);
            }
            continue;
        }
        if (aligned && p + 16u <= pe_w) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(a.out + p);
            const uint64_t lo = (uint64_t(v[1]) << 32) | v[0], hi = (uint64_t(v[3]) << 32) | v[2];
            const uint64_t r = p - hb, k0 = r >> 3;
            const uint32_t sh = 8u * (uint32_t(r) & 7u);                 // (wave-uniform: the lanes' blocks are 16 bytes apart)
            if (sh == 0u) acc += (lo + DIGEST_ONES) * mix64(k0) + (hi + DIGEST_ONES) * mix64(k0 + 1u);
            else {
                const uint32_t rs = 64u - sh;
                const uint64_t mA = ~0ull >> sh;                         // the block's first 8 - s bytes close word k0
                const uint64_t A = ((lo & mA) + (DIGEST_ONES & mA)) << sh;
                const uint64_t B = ((lo >> rs) | (hi << sh)) + DIGEST_ONES;
                const uint64_t C = (hi >> rs) + (DIGEST_ONES >> rs);     // its last s bytes open word k0 + 2
                acc += A * mix64(k0) + B * mix64(k0 + 1u) + C * mix64(k0 + 2u);
            }
        } else if (p < pe_w) digest_bytes(a, p, p + 16u < pe_w ? p + 16u : pe_w, h_cur);
    }
    flush();
}

// ---- launchers (host) -------------------------------------------------------
static inline uint32_t grid_for(uint64_t work_items, uint32_t cap)
{
    return uint32_t(work_items < cap ? (work_items ? work_items : 1) : cap);
}

// One buffer of '.' per device: '.'-fill descriptors gather from it like any other source.
static const uint8_t* device_dots(hipError_t* err)
{
    static std::mutex mu;
    static uint8_t* bufs[64] = {};
    int dev = 0;
    *err = hipGetDevice(&dev);
    if (*err != hipSuccess || dev < 0 || dev >= 64) { if (*err == hipSuccess) *err = hipErrorInvalidDevice; return nullptr; }
    std::lock_guard<std::mutex> lk(mu);
    if (!bufs[dev]) {
        uint8_t* p = nullptr;
        *err = hipMalloc(reinterpret_cast<void**>(&p), DOTS_BYTES);
        if (*err != hipSuccess) return nullptr;
        *err = hipMemset(p, '.', DOTS_BYTES);
        if (*err == hipSuccess) *err = hipDeviceSynchronize();
        if (*err != hipSuccess) { (void)hipFree(p); return nullptr; }
        bufs[dev] = p;
    }
    return bufs[dev];
}

// ---- phases ------------------------------------------------------------------
// A launch that streams more descriptor bytes from HBM than about 1/32 of what it stores runs at 3.5 TB/s where the bare copy runs at
// 5.9: reads that miss every cache, mixed into a saturated store stream, cost the memory system far more than their bytes
// (tools/wave_copy_bench.py: the cliff sits between 256 and 320 descriptor bytes per 8 KiB of result; hiding their LATENCY -- resident
// waves loading a chunk ahead -- recovers nothing).  So the two never mix: the chunk table is cut into phases of V2P_PHASE_BYTES of
// image (chunk records + descriptors), and before a phase's stitch kernels run, touch_image_kernel reads that phase's records and
// descriptor lines -- a read-only kernel at read speed, which leaves them in the memory-side cache (256 MB; the result stores are
// non-temporal and do not displace them) -- and the stitch kernels then find them there.  The copy benchmark: 3.47 -> 6.04 TB/s with
// 64 MB phases, the touch kernels' time included.
// (PHASE_BYTES_DEFAULT / PHASE_BYTES_RICH / PHASE_MIN_CHUNKS and image_is_rich(): sir_pack.hpp, next to the chunk order's rule)

__global__ __launch_bounds__(256) void touch_image_kernel(const uint64_t* __restrict__ desc, const Chunk* __restrict__ chunks, uint32_t n_chunks, uint64_t n_desc,
                                                          const uint8_t* __restrict__ payload, uint64_t payload_len, uint64_t* __restrict__ stage)
{
    // workgroup b runs on XCD b % 8: its four waves read chunks of residue b % 8
    touch_chunks(desc, chunks, touch_wave_first(blockIdx.x & 7u, (blockIdx.x >> 3) * 4u + (threadIdx.x >> 6)), n_chunks, n_desc, payload, payload_len, threadIdx.x & 63u, stage);
}

static hipError_t launch_stitch_range(const StitchArgs& args, hipStream_t stream, int nontemporal, uint32_t max_blocks);

// (experiments, libv2p_bench.so only: V2P_PHASE_GAP_US -- one wave idling that long between two phases; V2P_PHASE_SYNC -- the host waits
// for every phase.  Neither changed the step time.)
#ifdef V2P_BENCH_VARIANTS
__global__ void idle_kernel(uint32_t us)
{
    const uint64_t t0 = __builtin_readcyclecounter();       // (s_memtime: 100 MHz)
    while (__builtin_readcyclecounter() - t0 < uint64_t(us) * 100u) __builtin_amdgcn_s_sleep(64);
}
#endif

// chunks per phase and whether the read-ahead rides on the stitch launches: launch_stitch()'s rule, also asked by the caller that sizes the
// staging buffers (0: one launch, no phases)
static uint64_t phase_plan(const StitchArgs& a, int nontemporal, uint32_t max_blocks, bool staged, bool* ride_out)
{
    *ride_out = false;
    if (a.n_chunks == 0) return 0;
    const uint64_t img_desc = a.img_desc ? a.img_desc : a.n_desc, img_bytes = a.img_bytes ? a.img_bytes : a.out_len;
    const bool rich = image_is_rich(img_desc, img_bytes);
    uint64_t phase_bytes = rich ? PHASE_BYTES_RICH : PHASE_BYTES_DEFAULT;
    // STAGED descriptors are read from a buffer the read-ahead has just written, not from the image's lines it pulled in: a rich image
    // takes larger phases then -- while the reference is small (its slice next to the phase in every XCD's L2).  One shot of C3 whole (1.8 MB
    // of proteome): 11.95 / 11.89 / 11.82 / 11.67 / 11.80 / 11.82 ms with 28 / 36 / 40 / 44 / 48 / 56 MB, 14.7 with 64; C4 whole (56 MB):
    // 9.36 / 9.62 / 10.29 / 11.54 with 28 / 36 / 40 / 44 (profiles/r05_staged_phase_sweep.json)
    if (staged && rich && a.src0_len <= PHASE_STAGED_SMALL_REF) phase_bytes = PHASE_BYTES_STAGED;
    if (a.opt_phase_bytes == ~0ull) phase_bytes = 0;
    else if (a.opt_phase_bytes != 0) phase_bytes = a.opt_phase_bytes;
    const bool streams = (nontemporal & 4) != 0 || (nontemporal & 16) == 0;
    const uint32_t min_chunks = a.opt_phase_min_chunks ? a.opt_phase_min_chunks : PHASE_MIN_CHUNKS;
    if (max_blocks != 0 || phase_bytes == 0 || !streams || a.n_chunks < min_chunks) return 0;
    const double per_chunk = 16.0 + 8.0 * double(img_desc) / double(a.n_chunks);
    uint64_t per = uint64_t(double(phase_bytes) / per_chunk);
    const uint64_t per_min = a.opt_phase_min_chunks ? 8u : 4096u;
    per = per < per_min ? per_min : (per & ~7ull);
    const bool no_touch = (a.opt_touch & 1u) != 0, own_touch = (a.opt_touch & 2u) != 0;
    *ride_out = !own_touch && !no_touch && (nontemporal & 4) != 0 && (nontemporal & 48) == 48 && !(nontemporal & 2);
    return per;
}

uint32_t stitch_stage_chunks(const StitchArgs& args, int nontemporal)
{
    bool ride = false;
    const uint64_t per = phase_plan(args, nontemporal, 0, true, &ride);
    const int wsel = (nontemporal >> 28) & 3;
    if (!ride || !(nontemporal & 8) || wsel != 0 || (args.opt_touch & 4u) != 0 || args.opt_dual || per == 0 || per > 0x7FFFFFFFull) return 0;
    return uint32_t(per < args.n_chunks ? per : ((uint64_t(args.n_chunks) + 7u) & ~7ull));
}

hipError_t launch_stitch(const StitchArgs& args, hipStream_t stream, int nontemporal, uint32_t max_blocks)
{
    if (args.n_chunks == 0) return hipSuccess;
    StitchArgs a = args;
    hipError_t err = hipSuccess;
    a.dots = device_dots(&err);
    if (!a.dots) return err;
    // 64 MB of image per phase; 28 MB where the image is more than 1/33 of the result it describes: about 3.5 MB per XCD, most of which
    // its 4 MB L2 holds next to the proteome windows it works on.  With the chunk table dealt inside blocks of the arena (sir_pack.hpp)
    // C3 whole runs 7.44 / 7.42 / 7.57 / 7.63 ms with 24 / 28 / 32 / 36 MB phases, 8.2 with 12, 8.1 with 48 (with ONE order for the whole
    // table the best was 20 MB: 8.85 ms; 10.3 with 32, 12.0 with 64); C4 whole 6.26 / 6.16 / 7.09 with 20 / 32 / 48; C2, whose image is
    // 1/45 of its result: 3.31 / 3.24 / 3.21 with 16 / 32 / 64 the other way.
    a.rows = (nontemporal & 8) != 0;                                             // (bit 3: a rows image, sir_pack.hpp: CHUNK_CLIP on every chunk)
    const uint64_t img_desc = a.img_desc ? a.img_desc : a.n_desc, img_bytes = a.img_bytes ? a.img_bytes : a.out_len;
    const bool rich = image_is_rich(img_desc, img_bytes);                        // C2: 2.2 %, C4: 3.7 %, C3: 5 %
    // (phase size, store policy and threshold are launch options -- v2p_launch_opts, v2p_set_launch_opts -- for A/B runs and tests:
    // nothing here reads the environment)
    // ("sc1 nt" row stores: by default images with a thin descriptor stream get them -- C2 3.14 -> 3.06 ms (-2.6 %), C4 +0.9 %, C3 +6 %)
    a.store_sc1 = a.opt_store_sc1 >= 0 ? uint32_t(a.opt_store_sc1 != 0) : uint32_t(!rich);
    // (the kernels of per-block and dense images are bound by their instruction stream, not by memory: phases only cost them --
    // C3 per-block 2.00 -> 2.19 ms, C5 dense 0.53 -> 0.71; wave and long-run images gain: C2 3.26 -> 2.65 ms)
    // a pure wave image: the read-ahead of phase k + 1 rides on the trailing workgroups of phase k's launch (they are dispatched while
    // its last chunks drain) instead of a kernel of its own between the two; only phase 0 has a touch kernel (opt_touch 2: A/B)
    bool ride = false;
    const uint64_t per = phase_plan(a, nontemporal, max_blocks, args.stage != nullptr && a.rows, &ride);
    if (per == 0) return launch_stitch_range(a, stream, nontemporal, max_blocks);
    const bool no_touch = (a.opt_touch & 1u) != 0;
    // (V2P_PHASE_ONE_LAUNCH, A/B: ONE launch for all phases, the read-ahead workgroups of phase g + 1 placed in the grid before the
    // stitch workgroups of phase g (stitch_wave.hip) -- no kernel boundary, no tail, no launch gap between two phases.  Measured:
    // C2 3.16 against 3.21 ms, but C3 1.98 against 1.85 and C3 whole 11.1 against 9.35: the boundary is what keeps two phases'
    // images from sharing the L2 and the read-ahead from running into the stores.  Not the default.)
    const int wsel = (nontemporal >> 28) & 3;
    if (ride && wsel == 0 && (a.opt_touch & 4u) != 0 && per <= 0x7FFFFFFFull) {
        const uint32_t n0 = uint32_t(args.n_chunks < per ? args.n_chunks : per);
        hipLaunchKernelGGL(touch_image_kernel, dim3(8u * ((touch_waves_per_xcd(n0) + 3u) / 4u)), dim3(256), 0, stream, a.desc, a.chunks, n0, a.n_desc, a.src1, a.src1_len, nullptr);
        a.phase_chunks = uint32_t(per);
        return launch_stitch_range(a, stream, nontemporal, 0);
    }
    if (ride && wsel == 0 && a.opt_dual && a.aux_stream && a.ev_fork && a.ev_join && args.n_chunks >= 4u * per) {
        // DUAL: pieces of half a phase alternate between two streams.  A stream's kernel boundary (the drain of its last waves -- a wave
        // lives about 17 us of a 126 us phase -- and the ramp of the next kernel's first ones) then falls into the other stream's
        // mid-flight, which takes the freed wave slots; in flight at any time: two half-phases' image, as much as one phase before.  The
        // read-ahead rides as before, on the trailing workgroups of the piece two back (the one before it on the same stream).
        const uint64_t half = (per / 2u) & ~7ull;
        std::vector<std::pair<uint64_t, uint32_t>> pieces;                            // (first chunk, chunks)
        for (uint64_t c0 = 0; c0 < args.n_chunks; ) {
            uint64_t nc = pieces.size() == 1u ? (half / 2u) & ~7ull : half;           // (the second stream's first piece is half as long: the stagger)
            if (nc == 0u) nc = 8u;
            if (nc > args.n_chunks - c0) nc = args.n_chunks - c0;
            pieces.emplace_back(c0, uint32_t(nc));
            c0 += nc;
        }
        err = hipEventRecord(a.ev_fork, stream);
        if (err == hipSuccess) err = hipStreamWaitEvent(a.aux_stream, a.ev_fork, 0);
        if (err != hipSuccess) return err;
        for (size_t i = 0; i < pieces.size(); ++i) {
            hipStream_t s = (i & 1u) ? a.aux_stream : stream;
            a.chunks = args.chunks + pieces[i].first;
            a.n_chunks = pieces[i].second;
            a.next_chunks = nullptr; a.n_next = 0;
            if (i + 2u < pieces.size()) { a.next_chunks = args.chunks + pieces[i + 2u].first; a.n_next = pieces[i + 2u].second; }
            if (i < 2u) hipLaunchKernelGGL(touch_image_kernel, dim3(8u * ((touch_waves_per_xcd(a.n_chunks) + 3u) / 4u)), dim3(256), 0, s, a.desc, a.chunks, a.n_chunks, a.n_desc, a.src1, a.src1_len, nullptr);
            err = launch_stitch_range(a, s, nontemporal, 0);
            if (err != hipSuccess) return err;
        }
        err = hipEventRecord(a.ev_join, a.aux_stream);
        if (err == hipSuccess) err = hipStreamWaitEvent(stream, a.ev_join, 0);
        return err != hipSuccess ? err : hipGetLastError();
    }
    // STAGED descriptors: the ride form of a rows image with the caller's two buffers (stitch_kernels.h)
    const bool staged = ride && a.rows && wsel == 0 && args.stage != nullptr && per <= args.stage_chunks;
    uint32_t phase = 0;
    for (uint64_t c0 = 0; c0 < args.n_chunks; c0 += per, ++phase) {
        const uint32_t nc = uint32_t(args.n_chunks - c0 < per ? args.n_chunks - c0 : per);
        a.chunks = args.chunks + c0;
        a.n_chunks = nc;
        a.next_chunks = nullptr; a.n_next = 0;
        if (ride && c0 + per < args.n_chunks) {
            a.next_chunks = args.chunks + c0 + per;
            a.n_next = uint32_t(args.n_chunks - (c0 + per) < per ? args.n_chunks - (c0 + per) : per);
        }
        a.stage_cur = nullptr; a.stage_next = nullptr;
        if (staged) {
            a.stage_cur = args.stage + uint64_t(phase & 1u) * args.stage_chunks * STAGE_SLOTS;
            a.stage_next = args.stage + uint64_t((phase + 1u) & 1u) * args.stage_chunks * STAGE_SLOTS;
        }
        if (!no_touch && (!ride || c0 == 0)) hipLaunchKernelGGL(touch_image_kernel, dim3(8u * ((touch_waves_per_xcd(nc) + 3u) / 4u)), dim3(256), 0, stream, a.desc, a.chunks, nc, a.n_desc, a.src1, a.src1_len,
                                                                 staged ? const_cast<uint64_t*>(a.stage_cur) : nullptr);
        err = launch_stitch_range(a, stream, nontemporal, 0);
        if (err != hipSuccess) return err;
#ifdef V2P_BENCH_VARIANTS
        if (const char* e = getenv("V2P_PHASE_GAP_US")) hipLaunchKernelGGL(idle_kernel, dim3(1), dim3(64), 0, stream, uint32_t(atoi(e)));
        if (getenv("V2P_PHASE_SYNC")) (void)hipStreamSynchronize(stream);
#endif
    }
    return hipGetLastError();
}

static hipError_t launch_stitch_range(const StitchArgs& a, hipStream_t stream, int nontemporal, uint32_t max_blocks)
{
    hipError_t err = hipSuccess;
    // `nontemporal` bit 0: nt result stores; bits 4 / 5: no long-run / no per-block chunk in the image; bits 6..7: tasks per lane
    // of the largest long-run chunk; bits 8..11: tasks per lane of the largest per-block chunk; bits 12..15: variant (0 = default, 1 / 2 =
    // the per-block kernel with byte-granular / aligned gathers for every chunk -- images without fused descriptors only, A/B runs;
    // 3 = the per-block kernel also where the dense kernel would be picked; 4..6: stitch4 with 1 / 2 / 4 rows per round; 7 / 11 =
    // stitch4 with the chunk's reference span staged in a 36 / 20 KiB LDS window; 8 / 9 = the dense kernel for every per-block
    // chunk, with byte-granular / dword-aligned gathers); bits 16..23: timing-only ablation; bits 24..30: KiB of idle LDS (experiments)
    const int nt = nontemporal & 1;
    const int var = (nontemporal >> 12) & 0xF;
    const int dbg = (nontemporal >> 16) & 0xFF;
#ifndef V2P_BENCH_VARIANTS
    // timing-only ablations (wrong results), the kernels' A/B variants, idle LDS and waves-per-group selectors are not in this
    // library (libv2p_bench.so has them): only the routing-only variants 3 and 8 are
    if (dbg || (var != 0 && var != 3 && var != 8) || (nontemporal >> 24) != 0) return hipErrorInvalidValue;
#endif
    const uint32_t grid = grid_for(a.n_chunks, max_blocks ? max_blocks : 0x7FFFFFFFu);
    int tpt = (nontemporal >> 8) & 0xF;                 // chunks hold <= 256*tpt tasks
    if (tpt == 0) tpt = STITCH_TASKS_PER_LANE;
    const uint32_t lds_pad = uint32_t((nontemporal >> 24) & 0x7F) * 1024u;
#define V2P_KARGS a.desc, a.chunks, a.src0, a.src1, a.out, a.status, a.dots, a.n_chunks, a.n_desc, a.src0_len, a.src1_len, a.out_len
#define V2P_L(TT, NTT, VV, DD, FF) hipLaunchKernelGGL((stitch_kernel<TT, NTT, VV, DD>), dim3(grid), dim3(256), 0, stream, V2P_KARGS, uint32_t(FF))
    // (the timing-only ablations of the kernels -- DBG != 0: results are wrong -- exist only in the V2P_BENCH_VARIANTS build of this
    // file, libv2p_bench.so; the engine library instantiates DBG = 0 alone and refuses the bits)
#ifdef V2P_BENCH_VARIANTS
#define V2P_LAUNCH_V(TT, VV, FF) do { \
        if (dbg == 1) V2P_L(TT, true, VV, 1, FF); \
        else if (dbg == 2) V2P_L(TT, true, VV, 2, FF); \
        else if (dbg == 3) V2P_L(TT, true, VV, 3, FF); \
        else if (dbg == 4) V2P_L(TT, true, VV, 4, FF); \
        else if (dbg == 20) V2P_L(TT, true, VV, 20, FF); \
        else V2P_L(TT, true, VV, 0, FF); } while (0)
#else
#define V2P_LAUNCH_V(TT, VV, FF) V2P_L(TT, true, VV, 0, FF)
#endif
#define V2P_L4(TT, NTT, DD, RR, FF) hipLaunchKernelGGL((stitch4_kernel<TT, NTT, DD, RR>), dim3(grid), dim3(256), lds_pad, stream, V2P_KARGS, uint32_t(FF))
#ifdef V2P_BENCH_VARIANTS
#define V2P_LAUNCH(TT, FF) do { \
        if (!nt) V2P_L(TT, false, 0, 0, FF); \
        else if (var == 1) V2P_LAUNCH_V(TT, 1, FF); \
        else V2P_LAUNCH_V(TT, 0, FF); } while (0)
#define V2P_L3(TT, NTT, DD, FF) do { \
        if (var == 4) V2P_L4(TT, NTT, DD, 1, FF); \
        else if (var == 7) hipLaunchKernelGGL((stitch4_kernel<TT, NTT, DD, 8, 36864u>), dim3(grid), dim3(256), lds_pad, stream, V2P_KARGS, uint32_t(FF)); \
        else if (var == 11) hipLaunchKernelGGL((stitch4_kernel<TT, NTT, DD, 8, 20480u>), dim3(grid), dim3(256), lds_pad, stream, V2P_KARGS, uint32_t(FF)); \
        else if (var == 5) V2P_L4(TT, NTT, DD, 2, FF); \
        else if (var == 6) V2P_L4(TT, NTT, DD, 4, FF); \
        else V2P_L4(TT, NTT, DD, 8, FF); } while (0)
#else
#define V2P_LAUNCH(TT, FF) do { if (!nt) V2P_L(TT, false, 0, 0, FF); else V2P_LAUNCH_V(TT, 0, FF); } while (0)
#define V2P_L3(TT, NTT, DD, FF) V2P_L4(TT, NTT, DD, 8, FF)
#endif
#ifdef V2P_BENCH_VARIANTS
#define V2P_LAUNCH3(TT, FF) do { \
        if (!nt) V2P_L3(TT, false, 0, FF); \
        else if (dbg == 1) V2P_L3(TT, true, 1, FF); \
        else if (dbg == 2) V2P_L3(TT, true, 2, FF); \
        else if (dbg == 4) V2P_L3(TT, true, 4, FF); \
        else if (dbg == 20) V2P_L3(TT, true, 20, FF); \
        else V2P_L3(TT, true, 0, FF); } while (0)
#else
#define V2P_LAUNCH3(TT, FF) do { if (!nt) V2P_L3(TT, false, 0, FF); else V2P_L3(TT, true, 0, FF); } while (0)
#endif
    // which kernels run: bits 4 / 5 of `nontemporal` say the image has no long-run / no per-block chunk (hints: a workgroup
    // that finds a chunk of the other kind returns at once)
    const bool dense = (var == 8 || var == 9 || (tpt > 2 && var == 0)) && max_blocks == 0;    // 9: dword-aligned gathers
    const bool per_block_only = var == 1 || var == 2 || var == 3 || (dbg == 3 && !dense) || max_blocks != 0;   // A/B: every chunk on the per-block kernel (unfused images)
    // a capped grid (max_blocks) is the per-block kernel's persistent loop: an image with long-run, dense or wave chunks (fused
    // descriptors, which that kernel does not know) is refused instead of being reported chunk by chunk as "source out of bounds"
    if (max_blocks != 0 && ((nontemporal & (2 | 4)) || !(nontemporal & 16))) return hipErrorInvalidValue;
    const int tpt_long = (nontemporal >> 6) & 3;        // bits 6..7: tasks per lane of the largest long-run chunk (0 = 2)
    // dense images (chunks of more than 512 short tasks) go to stitch_dense_kernel; variant 3 keeps them on the per-block kernel and
    // variant 8 sends every per-block chunk of any image there (A/B runs)
#define V2P_LDD(NTT, DWW, DD, FF) do { if ((FF) == 3 && a.rows) hipLaunchKernelGGL((stitch_dense_kernel<NTT, false, 0, true, true>), dim3(a.n_chunks), dim3(256), 0, stream, V2P_KARGS, uint32_t(FF)); \
        else if ((FF) == 3) hipLaunchKernelGGL((stitch_dense_kernel<NTT, DWW, DD, true>), dim3(a.n_chunks), dim3(256), 0, stream, V2P_KARGS, uint32_t(FF)); \
        else hipLaunchKernelGGL((stitch_dense_kernel<NTT, DWW, DD, false>), dim3(a.n_chunks), dim3(256), 0, stream, V2P_KARGS, uint32_t(FF)); } while (0)
#ifdef V2P_BENCH_VARIANTS
#define V2P_LD(NTT, FF) do { if (var == 9) V2P_LDD(NTT, true, 0, FF); else if (dbg == 1) V2P_LDD(NTT, false, 1, FF); else if (dbg == 2) V2P_LDD(NTT, false, 2, FF); \
        else if (dbg == 3) V2P_LDD(NTT, false, 3, FF); else V2P_LDD(NTT, false, 0, FF); } while (0)
#else
#define V2P_LD(NTT, FF) V2P_LDD(NTT, false, 0, FF)
#endif
    // chunks flagged for stitchw_kernel (bit 2): one wave per chunk; bits 28..29: waves per workgroup (0 = 1, 1 = 2, 2 = 4; A/B runs)
    if ((nontemporal & 4) && !max_blocks) {
        const int wsel = (nontemporal >> 28) & 3;
        err = launch_stitch_wave(a, stream, nt != 0, wsel == 2 ? 4 : (wsel == 1 ? 2 : 1));
        if (err != hipSuccess) return err;
        if ((nontemporal & 48) == 48 && !(nontemporal & 2)) return hipSuccess;      // a pure wave image: no other kernel has work
    }
    if (per_block_only) {
        switch (tpt) { case 1: V2P_LAUNCH(1, 0); break; case 2: V2P_LAUNCH(2, 0); break; default: V2P_LAUNCH(4, 0); break; }
    } else {
        // chunks flagged dense (bit 1) always go to the dense kernel; the plain ones follow `dense`
        const bool plain = !(nontemporal & 32), flagged = (nontemporal & 2) != 0, plain_to_dense = plain && dense;
        if (flagged || plain_to_dense) { if (nt) V2P_LD(true, plain_to_dense ? 2 : 3); else V2P_LD(false, plain_to_dense ? 2 : 3); }
        if (plain && !plain_to_dense) switch (tpt) { case 1: V2P_LAUNCH(1, 2); break; case 2: V2P_LAUNCH(2, 2); break; default: V2P_LAUNCH(4, 2); break; }
        if (!(nontemporal & 16)) { if (tpt_long == 1) V2P_LAUNCH3(1, 1); else V2P_LAUNCH3(2, 1); }
    }
#undef V2P_LD
#undef V2P_LDD
#undef V2P_LAUNCH3
#undef V2P_L3
#undef V2P_L4
#undef V2P_LAUNCH
#undef V2P_LAUNCH_V
#undef V2P_L
#undef V2P_KARGS
    return hipGetLastError();
}

hipError_t launch_ordered(const OrderedArgs& a, hipStream_t stream)
{
    if (a.n_tasks == 0) return hipSuccess;
    hipLaunchKernelGGL(ordered_kernel, dim3(1), dim3(1024), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_validate(const ValidateArgs& a, hipStream_t stream)
{
    if (a.n_tasks == 0) return hipSuccess;
    const uint64_t blocks = (a.n_tasks + 255) / 256;
    hipLaunchKernelGGL(validate_kernel, dim3(uint32_t(blocks)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_digest(const DigestArgs& a, uint64_t out_bytes, hipStream_t stream)
{
    if (a.n_haps == 0 || out_bytes == 0) return hipSuccess;
    const uint64_t nblk = (out_bytes + 15) / 16, per_group = 4ull * 64u * DIGEST_ITER;       // four waves of 64 KiB each
    const uint64_t groups = (nblk + per_group - 1) / per_group;
    if (groups > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(digest_kernel, dim3(uint32_t(groups)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// HIP loads a translation unit's code object when one of its kernels is first used -- 17 ms for the stitch kernels, paid by the first
// execute of a process (C3 whole: 26 ms instead of 8.4).  v2p_init launches one empty kernel per unit instead.
__global__ void code_object_loader_a() {}   // (a kernel of its own, so that profiles of the real ones hold no empty launches)
hipError_t preload_stitch_kernels(hipStream_t stream)
{
    hipError_t err = hipSuccess;
    (void)device_dots(&err);                                       // (the '.' buffer of the device: an allocation, a fill and a wait)
    hipLaunchKernelGGL(code_object_loader_a, dim3(1), dim3(64), 0, stream);
    return err != hipSuccess ? err : hipGetLastError();
}

}  // namespace v2p
