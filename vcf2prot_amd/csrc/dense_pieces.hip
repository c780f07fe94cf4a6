// dense_pieces.hip -- PIECE images (dense_pieces.h): the re-execution form of a dense rows image, and its executor.
//
// stitch_dense_kernel is bound by its vector instructions, exactly (570 M for C5, x 4 cycles / 1 024 SIMDs = its
// duration), and most of them are bookkeeping per descriptor, not byte moving: 62 to decode one of three descriptor kinds, two
// workgroup scans for positions, a list of continuation pieces for what a descriptor has beyond sixteen bytes, the substituted
// residues of fused descriptors in a pass of their own -- 225 per descriptor slot against 25 for the put itself.  All of it is the
// same every time the image is executed.  So an image that is executed AGAIN is first re-written, once, as PIECES: <= 16 result bytes
// of one source each, with their position inside the chunk and at most one substituted residue -- what the kernel derives, stored.
//   pieces_build_kernel  one workgroup per chunk, lane = 4 descriptors: the dense kernel's decode (all kinds, head skip / tail clip of a
//                        rows image, the bounds Task::execute would panic on), a scan for positions, every descriptor cut into pieces;
//                        pass 0 counts, pass 1 writes (a scan of the chunks' counts in between)
//   stitch_pieces_kernel one workgroup per chunk, lane = piece: record, gather (an immediate or a '.' fill needs none), the residue,
//                        the put into the chunk's LDS image (the dense kernel's dense_put); then the image leaves as aligned 16-byte
//                        non-temporal stores.  No decode, no scan, one barrier between the puts and the stores.
// Byte / index work only (no MFMA).  8 bytes of image per piece: C5 1.35 pieces per descriptor.
// Measured (C5, 20 000 haplotypes; tools/pieces_probe.py, profiles/r05_pieces.txt): 0.73 ms against the dense kernel's 0.83-0.85, half its
// vector instructions (255 M against 465 M) -- the piece kernel is no longer bound by them (58 % busy) but by the memory system: its
// record stream (1.1 GB, cold reads) runs into its own 1.6 GB of result stores, DESIGN.md section 3's effect, at 3.8 TB/s together.
// (Phases with the next phase's records read ahead on trailing workgroups, what wave images gain a third from, cost this one: 1.03 ms
// with 256 MB phases, 1.24 with 16 MB, against 0.77 in one launch -- a read-ahead workgroup per chunk is as many workgroups again.)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "dense_pieces.h"
#include "stitch_kernels.h"
#include "stitch_device.hpp"

namespace v2p {

namespace {

constexpr uint32_t PIECES_STAGE = 12288u;                  // bytes of the LDS image (= stitch_dense_kernel's: a dense chunk is <= twelve rows)
constexpr uint32_t STATUS_PIECES_REFUSED = 9;              // a chunk / descriptor the dense kernel itself would refuse, or one that does not fit the form

struct Dec {
    uint32_t space; uint64_t src; uint32_t len;
    bool lit1, lit2; uint32_t p1, p2, b1, b2;
    bool bad;
};

// one descriptor of a dense rows image (stitch_dense_kernel's load_tasks, RIMG): plain / '.' fill / immediate / fused substitution (SNV3) /
// two in a row (SNV5); hs, tcl: head skip of the chunk's first descriptor, tail clip of its last
__device__ __forceinline__ Dec decode_desc(uint64_t d, uint32_t hs, uint32_t tcl, uint64_t src0_len, uint64_t src1_len)
{
    Dec r;
    const uint32_t dlo = uint32_t(d), dhi = uint32_t(d >> 32);
    const bool is5 = (dhi >> 28) == 0xDu, is3 = (dhi >> 29) == 7u, fz = is5 || is3;
    const uint32_t f29 = uint32_t(d >> 29);
    const uint32_t l1 = f29 & (is5 ? 31u : 0xFFFu);
    const uint32_t l2 = is5 ? (f29 >> 5) & 31u : (f29 >> 12) & 0xFFFu;
    const uint32_t l3 = is5 ? (f29 >> 10) & 31u : 0u;
    r.b1 = (is5 ? f29 >> 15 : f29 >> 24) & 0xFFu; r.b2 = (f29 >> 23) & 0xFFu;
    uint32_t ln = fz ? l1 + 1u + l2 + (is5 ? 1u + l3 : 0u) : (dhi >> 8) & 0x3FFFFFu;
    const uint32_t used = fz ? (l3 ? ln : (l2 ? l1 + 1u + l2 : l1)) : ln;      // source bytes actually read (the literal may sit on the run's last residue)
    r.space = fz ? SPACE_PROTEOME : dhi >> 30;
    uint64_t so = fz ? uint64_t(dlo & 0x1FFFFFFFu) : ((uint64_t(dhi & 0xFFu) << 32) | dlo);
    const uint64_t limit = r.space == SPACE_PROTEOME ? src0_len : (r.space == SPACE_PAYLOAD ? src1_len : ~0ull);
    r.bad = so + used > limit || (r.space == SPACE_IMM && ln > IMM_MAX_BYTES);
    r.p1 = l1; r.p2 = l1 + 1u + l2; r.lit1 = fz; r.lit2 = is5;
    if (hs + tcl != 0u) {
        if (hs + tcl >= ln) r.bad = true;
        else {
            ln -= hs + tcl;
            if (hs) so = r.space == SPACE_IMM ? so >> (8u * hs) : (r.space == SPACE_FILL ? so : so + hs);
            r.lit1 = r.lit1 && hs <= r.p1; r.lit2 = r.lit2 && hs <= r.p2;
            r.p1 -= hs; r.p2 -= hs;
        }
    }
    r.lit1 = r.lit1 && r.p1 < ln; r.lit2 = r.lit2 && r.p2 < ln;
    r.src = so; r.len = r.bad ? 0u : ln;
    return r;
}

__device__ __forceinline__ uint64_t pack_piece(uint32_t space, uint64_t src, uint32_t dst, uint32_t len, bool lit, uint32_t litpos, uint32_t byte)
{
    // bits: 0..30 src | 31..32 space | 33..46 dst | 47..50 len - 1 | 51..54 litpos | 55 has | 56..63 byte
    if (space == SPACE_IMM)          // up to five bytes: bits 0..30 and 51..63 hold them
        return (src & PIECE_SRC_MAX) | (uint64_t(SPACE_IMM) << 31) | (uint64_t(dst) << 33) | (uint64_t(len - 1u) << 47) | ((src >> 31) << 51);
    return (src & PIECE_SRC_MAX) | (uint64_t(space) << 31) | (uint64_t(dst) << 33) | (uint64_t(len - 1u) << 47) | (uint64_t(litpos & 15u) << 51) |
           (uint64_t(lit ? 1u : 0u) << 55) | (uint64_t(byte & 0xFFu) << 56);
}

// the pieces of one descriptor: ranges of <= 32 bytes from its first byte on, a range holding at most ONE substituted residue (two inside
// one range: it ends in front of the second); EMIT = false: how many
template <bool EMIT>
__device__ __forceinline__ uint32_t cut_pieces(const Dec& r, uint32_t off, uint64_t* out)
{
    uint32_t n = 0;
    for (uint32_t a = 0; a < r.len; ) {
        uint32_t b = a + PIECE_BYTES < r.len ? a + PIECE_BYTES : r.len;
        const bool in1 = r.lit1 && r.p1 >= a && r.p1 < b;
        bool in2 = r.lit2 && r.p2 >= a && r.p2 < b;
        if (in1 && in2) { b = r.p2; in2 = false; }
        if (EMIT) {
            const bool lit = in1 || in2;
            const uint32_t lp = in1 ? r.p1 - a : r.p2 - a, lb = in1 ? r.b1 : r.b2;
            const uint64_t src = r.space == SPACE_IMM ? r.src >> (8u * a) : (r.space == SPACE_FILL ? 0ull : r.src + a);
            out[n] = pack_piece(r.space, src, off + a, b - a, lit, lp, lb);
        }
        ++n;
        a = b;
    }
    return n;
}

// exclusive prefix over the workgroup's 256 lanes (four waves) + the total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_part, uint32_t& total)
{
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v);
    if (lane == 63u) s_part[wid] = incl;
    __syncthreads();
    const uint32_t w0 = s_part[0], w1 = s_part[1], w2 = s_part[2], w3 = s_part[3];
    total = w0 + w1 + w2 + w3;
    const uint32_t before = (wid > 0 ? w0 : 0u) + (wid > 1 ? w1 : 0u) + (wid > 2 ? w2 : 0u);
    __syncthreads();
    return before + incl - v;
}

template <int PASS>
__global__ __launch_bounds__(256) void pieces_build_kernel(PieceBuildArgs a)
{
    __shared__ uint32_t s_part[4];
    const uint32_t c = blockIdx.x, tid = threadIdx.x;
    const Chunk ch = a.chunks[c];
    const uint64_t tb = ch.task_begin & TB_IDX_MASK, dn = ch.dst_n;
    const uint32_t hskip = chunk_head_skip(ch.task_begin), tclip = chunk_tail_clip(ch.task_begin);
    const uint32_t n = uint32_t(dn >> 48) & CHUNK_N_MASK;
    const uint64_t dst = dn & DST_MASK;
    // only the chunks of a dense rows image are taken: anything else leaves the image to the kernels it was built for
    bool refused = !(dn & CHUNK_DENSE) || !(dn & CHUNK_CLIP) || (dn & (CHUNK_LONG | CHUNK_WAVE)) != 0ull || n > CHUNK_TASKS_DEEP || tb > a.n_desc || n > a.n_desc - tb || (dst & 1023u) != 0u;
    Dec r[4];
    uint32_t lsum = 0, npc = 0;
    bool bad = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t i = tid * 4u + uint32_t(k);
        const uint64_t d = (!refused && i < n) ? a.desc[tb + i] : 0ull;              // (0: an empty proteome copy)
        r[k] = decode_desc(d, i == 0u ? hskip : 0u, i + 1u == n ? tclip : 0u, a.src0_len, a.src1_len);
        if (i >= n) { r[k].len = 0u; r[k].bad = false; }
        bad = bad || r[k].bad || ((r[k].space == SPACE_PROTEOME || r[k].space == SPACE_PAYLOAD) && r[k].src + r[k].len > PIECE_SRC_MAX);
        lsum += r[k].len;
    }
    uint32_t total = 0;
    const uint32_t off0 = block_excl_scan(lsum, s_part, total);
    refused = refused || total > PIECES_STAGE || dst + total > a.out_len;
    if (PASS == 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) npc += cut_pieces<false>(r[k], 0u, nullptr);
        uint32_t n_pieces = 0;
        (void)block_excl_scan(npc, s_part, n_pieces);
        if (bad) atomicMin(a.status, (unsigned long long)(((tb + tid * 4u) << 8) | STATUS_PIECES_REFUSED));
        if (tid == 0) {
            if (refused || n_pieces > PIECE_CHUNK_MAX) atomicMin(a.status, (unsigned long long)((uint64_t(c) << 8) | STATUS_PIECES_REFUSED));
            a.count[c] = n_pieces;
        }
        return;
    }
    // PASS 1 (the image passed pass 0): positions of the thread's pieces, then the pieces
#pragma unroll
    for (int k = 0; k < 4; ++k) npc += cut_pieces<false>(r[k], 0u, nullptr);
    uint32_t n_pieces = 0;
    const uint32_t p0 = block_excl_scan(npc, s_part, n_pieces);
    const uint64_t base = a.base[c];
    uint64_t* out = a.pieces + base + p0;
    uint32_t off = off0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        out += cut_pieces<true>(r[k], off, out);
        off += r[k].len;
    }
    if (tid == 0) a.chunks2[c] = Chunk{base | (uint64_t(total) << 40), dst | (uint64_t(n_pieces) << 48) | CHUNK_DENSE | CHUNK_CLIP};
}

// up to n (1..16) bytes x, first byte lowest, OR-ed into the LDS image at byte offset o (stitch_kernels.hip: dense_put)
__device__ __forceinline__ void piece_put(uint32_t* img, const u32x4* lowmask, uint32_t o, u32x4 x, uint32_t n)
{
    const u32x4 m = lowmask[n];
    x[0] &= m[0]; x[1] &= m[1]; x[2] &= m[2]; x[3] &= m[3];
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

template <bool NT>
__global__ __launch_bounds__(256) void stitch_pieces_kernel(PieceExecArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_img[PIECES_STAGE / 4u + 8u];
    __shared__ u32x4 s_low[17];                             // s_low[j]: the low j bytes of a 16-byte block
    const uint32_t tid = threadIdx.x, c = blockIdx.x;
    const Chunk ch = a.chunks2[c];
    const uint64_t p0 = ch.task_begin & ((1ull << 40) - 1ull);
    const uint32_t span = uint32_t(ch.task_begin >> 40) & 0x3FFFu, n = uint32_t(ch.dst_n >> 48) & 0x7FFu;
    const uint64_t dst = ch.dst_n & DST_MASK;
    // EVERY round's records are requested before anything else (a chunk holds up to 2 047 pieces: eight per lane; C5: ~1 070, five): they
    // are cold reads between the result stores, and requested a round or two ahead each of them is waited for (0.78-0.80 ms against 0.73)
    constexpr uint32_t R = 8;
    uint64_t rec[R];
#pragma unroll
    for (uint32_t j = 0; j < R; ++j) { const uint32_t k = tid + 256u * j; rec[j] = k < n ? a.pieces[p0 + k] : 0ull; }
    if (tid < 17u) {
        u32x4 m;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) m[k] = tid >= 4u * k + 4u ? 0xFFFFFFFFu : (tid <= 4u * k ? 0u : (1u << (8u * (tid - 4u * k))) - 1u);
        s_low[tid] = m;
    }
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (uint32_t q = 0; q < PIECES_STAGE / 4096u; ++q) *reinterpret_cast<u32x4*>(&s_img[(q * 256u + tid) * 4u]) = z;
        if (tid < 2u) *reinterpret_cast<u32x4*>(&s_img[PIECES_STAGE / 4u + tid * 4u]) = z;
    }
    if (span > PIECES_STAGE || dst + span > a.out_len) return;                  // (never written by pieces_build_kernel)
    lds_barrier();
    // round j's put runs behind round j + 1's gather
    auto fetch = [&](uint64_t r) -> u32x4 {
        const uint32_t lo = uint32_t(r);
        const uint32_t space = uint32_t(r >> 31) & 3u;
        const bool mem = space == SPACE_PROTEOME || space == SPACE_PAYLOAD;
        return gather16(reinterpret_cast<uint64_t>(space == SPACE_PAYLOAD ? a.src1 : a.src0) + (mem ? (lo & 0x7FFFFFFFu) : 0u));   // (fills, immediates, idle lanes: the proteome's first bytes, dropped)
    };
    u32x4 g = fetch(rec[0]);
#pragma unroll
    for (uint32_t j = 0; j < R; ++j) {
        if (256u * j >= n) break;                                               // (uniform)
        u32x4 g0 = g;
        if (j + 1u < R && 256u * (j + 1u) < n) g = fetch(rec[j + 1u < R ? j + 1u : j]);
        const uint32_t k = tid + 256u * j;
        const uint64_t cur = rec[j];
        const uint32_t lo = uint32_t(cur), hi = uint32_t(cur >> 32);
        const uint32_t space = uint32_t(cur >> 31) & 3u;
        const uint32_t o = (hi >> 1) & 0x3FFFu, len = ((hi >> 15) & 15u) + 1u;
        if (space == SPACE_FILL) g0 = u32x4{0x2E2E2E2Eu, 0x2E2E2E2Eu, 0x2E2E2E2Eu, 0x2E2E2E2Eu};
        if (space == SPACE_IMM) {
            const uint64_t v = uint64_t(lo & 0x7FFFFFFFu) | ((cur >> 51) << 31);
            g0 = u32x4{uint32_t(v), uint32_t(v >> 32), 0u, 0u};
        } else if ((hi >> 23) & 1u) {                                            // the substituted residue of a fused descriptor
            const uint32_t q = (hi >> 19) & 15u, byte = hi >> 24;
            const uint32_t sh = 8u * (q & 3u), m = 0xFFu << sh, bv = byte << sh, w = q >> 2;
            g0[0] = w == 0u ? (g0[0] & ~m) | bv : g0[0];
            g0[1] = w == 1u ? (g0[1] & ~m) | bv : g0[1];
            g0[2] = w == 2u ? (g0[2] & ~m) | bv : g0[2];
            g0[3] = w == 3u ? (g0[3] & ~m) | bv : g0[3];
        }
        if (k < n) piece_put(s_img, s_low, o, g0, len);
    }
    lds_barrier();
    // the image leaves: whole 16-byte blocks as aligned stores (a rows chunk starts on a 1 KiB row), a ragged last block byte by byte
    uint8_t* const out0 = a.out + dst;
    const uint32_t nblk = (span + 15u) >> 4;
    for (uint32_t b = tid; b < nblk; b += 256u) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(&s_img[b * 4u]);
        uint8_t* o = out0 + (b << 4);
        if ((b << 4) + 16u <= span) {
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(o));
            else *reinterpret_cast<u32x4*>(o) = v;
        } else {
            const uint32_t kb = span - (b << 4);
            for (uint32_t j = 0; j < kb; ++j) o[j] = uint8_t(v[j >> 2] >> (8u * (j & 3u)));
        }
    }
}

// ---- TILE images: one workgroup per tile of the parse (dense_pieces.h) ----
constexpr uint32_t TILES_STAGE = 16384u;                   // bytes of the LDS image: TILE_SPAN_MAX + the 15 bytes a range may start inside its first block

// workgroup b of a grid of G -> the j-th tile of the contiguous range XCD b % 8 owns (build_rows.hip: xcd_contiguous): neighbouring tiles
// -- which share the 16-byte block and the cache line their ranges meet in -- are written through one L2
__device__ __forceinline__ uint32_t tiles_xcd_contiguous(uint32_t b, uint32_t G)
{
    const uint32_t x = b & 7u, j = b >> 3, q = G >> 3, r = G & 7u;
    return x * q + (x < r ? x : r) + j;
}

template <bool NT>
__global__ __launch_bounds__(256) void stitch_tiles_kernel(TileExecArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_img[TILES_STAGE / 4u + 8u];
    __shared__ u32x4 s_low[17];                             // s_low[j]: the low j bytes of a 16-byte block
    const uint32_t tid = threadIdx.x;
    const uint64_t c = a.tile0 + tiles_xcd_contiguous(blockIdx.x, gridDim.x);
    if (*a.status != ~0ull) return;                         // (the parse refused the stream, or reported what the reference would panic on)
    const uint32_t n = a.tile_count[c];
    const uint64_t dst = a.tile_res_base[c], end = a.tile_res_base[c + 1u];
    const uint32_t span = uint32_t(end - dst), lead = uint32_t(dst) & 15u;
    const uint64_t p0 = c * a.tile_slots;
    constexpr uint32_t R = TILE_SLOTS_MAX / 256u;
    uint64_t rec[R];
#pragma unroll
    for (uint32_t j = 0; j < R; ++j) { const uint32_t k = tid + 256u * j; rec[j] = k < n ? a.pieces[p0 + k] : 0ull; }
    if (tid < 17u) {
        u32x4 m;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) m[k] = tid >= 4u * k + 4u ? 0xFFFFFFFFu : (tid <= 4u * k ? 0u : (1u << (8u * (tid - 4u * k))) - 1u);
        s_low[tid] = m;
    }
    if (end <= dst || end - dst > TILE_SPAN_MAX || n > a.tile_slots || end > a.out_len) return;       // (an empty tile; the others: never built by the parse)
    const uint32_t nblk = (lead + span + 15u) >> 4;
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
        for (uint32_t b = tid; b < nblk + 2u; b += 256u) *reinterpret_cast<u32x4*>(&s_img[b * 4u]) = z;
    }
    lds_barrier();
    auto fetch = [&](uint64_t r) -> u32x4 {
        const uint32_t lo = uint32_t(r);
        const uint32_t space = uint32_t(r >> 31) & 3u;
        const bool mem = space == SPACE_PROTEOME || space == SPACE_PAYLOAD;
        return gather16(reinterpret_cast<uint64_t>(space == SPACE_PAYLOAD ? a.src1 : a.src0) + (mem ? (lo & 0x7FFFFFFFu) : 0u));
    };
    u32x4 g = fetch(rec[0]);
#pragma unroll
    for (uint32_t j = 0; j < R; ++j) {
        if (256u * j >= n) break;                                               // (uniform)
        u32x4 g0 = g;
        if (j + 1u < R && 256u * (j + 1u) < n) g = fetch(rec[j + 1u < R ? j + 1u : j]);
        const uint32_t k = tid + 256u * j;
        const uint64_t cur = rec[j];
        const uint32_t lo = uint32_t(cur), hi = uint32_t(cur >> 32);
        const uint32_t space = uint32_t(cur >> 31) & 3u;
        const uint32_t o = ((hi >> 1) & 0x3FFFu) + lead, len = ((hi >> 15) & 15u) + 1u;
        if (space == SPACE_FILL) g0 = u32x4{0x2E2E2E2Eu, 0x2E2E2E2Eu, 0x2E2E2E2Eu, 0x2E2E2E2Eu};
        if (space == SPACE_IMM) {
            const uint64_t v = uint64_t(lo & 0x7FFFFFFFu) | ((cur >> 51) << 31);
            g0 = u32x4{uint32_t(v), uint32_t(v >> 32), 0u, 0u};
        } else if ((hi >> 23) & 1u) {                                            // the substituted residue of a fused run
            const uint32_t q = (hi >> 19) & 15u, byte = hi >> 24;
            const uint32_t sh = 8u * (q & 3u), m = 0xFFu << sh, bv = byte << sh, w = q >> 2;
            g0[0] = w == 0u ? (g0[0] & ~m) | bv : g0[0];
            g0[1] = w == 1u ? (g0[1] & ~m) | bv : g0[1];
            g0[2] = w == 2u ? (g0[2] & ~m) | bv : g0[2];
            g0[3] = w == 3u ? (g0[3] & ~m) | bv : g0[3];
        }
        if (k < n && o + len <= lead + span) piece_put(s_img, s_low, o, g0, len);
    }
    lds_barrier();
    // the image leaves: whole 16-byte blocks as aligned stores; the first and the last block of the range, shared with the neighbouring
    // tiles, byte by byte where they are ragged
    uint8_t* const out0 = a.out + (dst - lead);
    for (uint32_t b = tid; b < nblk; b += 256u) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(&s_img[b * 4u]);
        uint8_t* o = out0 + (b << 4);
        const uint32_t lo_b = b == 0u ? lead : 0u, hi_b = (b << 4) + 16u <= lead + span ? 16u : lead + span - (b << 4);
        if (lo_b == 0u && hi_b == 16u) {
            if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(o));
            else *reinterpret_cast<u32x4*>(o) = v;
        } else {
            for (uint32_t j = lo_b; j < hi_b; ++j) o[j] = uint8_t(v[j >> 2] >> (8u * (j & 3u)));
        }
    }
}

__global__ void code_object_loader_e() {}

}  // namespace

hipError_t launch_pieces_build(const PieceBuildArgs& a, int pass, hipStream_t stream)
{
    if (a.n_chunks == 0) return hipSuccess;
    if (pass == 0) hipLaunchKernelGGL(pieces_build_kernel<0>, dim3(a.n_chunks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(pieces_build_kernel<1>, dim3(a.n_chunks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_stitch_pieces(const PieceExecArgs& a, hipStream_t stream, bool nontemporal)
{
    if (a.n_chunks == 0) return hipSuccess;
    if (nontemporal) hipLaunchKernelGGL(stitch_pieces_kernel<true>, dim3(a.n_chunks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(stitch_pieces_kernel<false>, dim3(a.n_chunks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_stitch_tiles(const TileExecArgs& a, hipStream_t stream, bool nontemporal)
{
    if (a.n_tiles == 0) return hipSuccess;
    if (a.n_tiles > 0x7FFFFFFFull || a.tile_slots == 0u || a.tile_slots > TILE_SLOTS_MAX) return hipErrorInvalidValue;
    if (nontemporal) hipLaunchKernelGGL(stitch_tiles_kernel<true>, dim3(uint32_t(a.n_tiles)), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(stitch_tiles_kernel<false>, dim3(uint32_t(a.n_tiles)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t preload_dense_pieces(hipStream_t stream)
{
    hipLaunchKernelGGL(code_object_loader_e, dim3(1), dim3(64), 0, stream);
    return hipGetLastError();
}

}  // namespace v2p
