// stitch_device.hpp -- device helpers shared by the gfx950 stitch kernels (stitch_kernels.hip, stitch_wave.hip, bench_kernels.hip):
// byte-granular 16-byte gathers, DPP lane exchange and wave64 scans, literal placement, the device status word.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "stitch_kernels.h"

namespace v2p {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) unaligned16 { u32x4 v; };

// 16 bytes from an arbitrary (unaligned) global address held as an integer.  The explicit global
// address space matters: a pointer rebuilt from an integer would otherwise be a FLAT access
// (counts on vmcnt AND lgkmcnt, returns out of order).
__device__ __forceinline__ u32x4 gather16(uint64_t addr)
{
    typedef const __attribute__((address_space(1))) unaligned16* gptr;
    return reinterpret_cast<gptr>(addr)->v;
}

// Same 16 bytes through dword-aligned loads (x4 + x1) and four v_alignbyte.
struct __attribute__((packed, aligned(4))) dwaligned16 { u32x4 v; };
__device__ __forceinline__ u32x4 gather16_dw(uint64_t addr)
{
    typedef const __attribute__((address_space(1))) dwaligned16* gptr;
    typedef const __attribute__((address_space(1))) uint32_t* dptr;
    const uint64_t base = addr & ~3ull;
    const uint32_t sh = uint32_t(addr) & 3u;
    const u32x4 v = reinterpret_cast<gptr>(base)->v;
    const uint32_t e = *reinterpret_cast<dptr>(base + 16u);
    u32x4 o;
    o.x = __builtin_amdgcn_alignbyte(v.y, v.x, sh);
    o.y = __builtin_amdgcn_alignbyte(v.z, v.y, sh);
    o.z = __builtin_amdgcn_alignbyte(v.w, v.z, sh);
    o.w = __builtin_amdgcn_alignbyte(e, v.w, sh);
    return o;
}

// 16 bytes from a 16-byte aligned global address: a wave whose lanes read consecutive blocks touches every
// 128-byte line exactly once (the byte-granular gather16 touches every line from two neighbouring lane quads).
__device__ __forceinline__ u32x4 load16_aligned(uint64_t addr)
{
    typedef const __attribute__((address_space(1))) u32x4* gptr;
    return *reinterpret_cast<gptr>(addr);
}

// DPP wave_shl:1 -- lane i receives lane i+1's value (lane 63 keeps `old`).
__device__ __forceinline__ uint32_t from_next_lane(uint32_t old, uint32_t x)
{
    return uint32_t(__builtin_amdgcn_update_dpp(int(old), int(x), 0x130, 0xf, 0xf, false));
}

// bytes d .. d+15 of the 32 bytes {v, n}, d in 0..15 (two select stages pick the dwords, v_alignbyte the bytes)
__device__ __forceinline__ u32x4 funnel16(u32x4 v, u32x4 n, uint32_t d)
{
    const bool s8 = (d & 8u) != 0u, s4 = (d & 4u) != 0u;
    const uint32_t t0 = s8 ? v[2] : v[0], t1 = s8 ? v[3] : v[1], t2 = s8 ? n[0] : v[2],
                   t3 = s8 ? n[1] : v[3], t4 = s8 ? n[2] : n[0], t5 = s8 ? n[3] : n[1];
    const uint32_t u0 = s4 ? t1 : t0, u1 = s4 ? t2 : t1, u2 = s4 ? t3 : t2, u3 = s4 ? t4 : t3, u4 = s4 ? t5 : t4;
    const uint32_t r = d & 3u;
    u32x4 o;
    o.x = __builtin_amdgcn_alignbyte(u1, u0, r);
    o.y = __builtin_amdgcn_alignbyte(u2, u1, r);
    o.z = __builtin_amdgcn_alignbyte(u3, u2, r);
    o.w = __builtin_amdgcn_alignbyte(u4, u3, r);
    return o;
}

// An immediate descriptor's literal bytes (<= 5, first byte lowest) placed at byte position q of a
// 16-byte block, q in -4..15 (negative: the task began in the previous block).
__device__ __forceinline__ u32x4 imm_block(uint64_t lit, int32_t q)
{
    const uint32_t sh = 8u * (uint32_t(q) & 7u);
    const uint64_t x = lit << sh;
    const uint64_t y = sh ? lit >> (64u - sh) : 0ull;
    uint64_t lo, hi;
    if (q < 0) { lo = lit >> (8u * uint32_t(-q)); hi = 0ull; }
    else if (q < 8) { lo = x; hi = y; }
    else { lo = 0ull; hi = x; }
    return u32x4{uint32_t(lo), uint32_t(lo >> 32), uint32_t(hi), uint32_t(hi >> 32)};
}

constexpr uint64_t ADJ_IMM = 1ull << 63;       // s_adj entry of an immediate task: flag | literal bytes
constexpr uint64_t ADJ_LIT = (1ull << 40) - 1;

// ---- wave64 inclusive add-scan with DPP (row_shr 1/2/4/8, row_bcast 15/31) ----
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x)
{
    x += __builtin_amdgcn_update_dpp(0u, x, 0x111, 0xf, 0xf, false);  // row_shr:1
    x += __builtin_amdgcn_update_dpp(0u, x, 0x112, 0xf, 0xf, false);  // row_shr:2
    x += __builtin_amdgcn_update_dpp(0u, x, 0x114, 0xf, 0xf, false);  // row_shr:4
    x += __builtin_amdgcn_update_dpp(0u, x, 0x118, 0xf, 0xf, false);  // row_shr:8
    x += __builtin_amdgcn_update_dpp(0u, x, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
    x += __builtin_amdgcn_update_dpp(0u, x, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
    return x;
}

__device__ __forceinline__ void report(unsigned long long* status, uint64_t index, uint32_t reason)
{
    atomicMin(status, (unsigned long long)((index << 8) | reason));
}

// Workgroup barrier that orders LDS only.  __syncthreads() also waits for vmcnt(0), i.e. for
// every result store the wave has in flight; nothing in these kernels re-reads its own stores.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}


// ---- reading a phase's image ahead of its stitch kernels (launch_stitch: phases) ----
// first chunk of read-ahead wave q of the workgroups that run on XCD r (workgroup index % 8): residue r, 4 chunks 8 apart per wave
__device__ __forceinline__ uint32_t touch_wave_first(uint32_t r, uint32_t q) { return r + 8u * 4u * q; }
__host__ __device__ __forceinline__ uint32_t touch_waves_per_xcd(uint32_t n_chunks) { return ((n_chunks + 7u) / 8u + 3u) / 4u; }
constexpr uint32_t STAGE_SLOTS = 64;               // descriptor slots per chunk of a staging buffer (= CHUNK_TASKS_WAVE)
constexpr uint32_t TOUCH_CHUNKS_PER_WAVE = 4;      // records, then first descriptors, then payload bytes of four chunks in flight per wave
// One wave: chunks [c0, c0 + 4) of `chunks` -- their records, every descriptor (one per lane and round; its line comes in) and a
// payload descriptor's first and last source byte (frameshift tails, long insertions: first touched by the stitch kernel they would
// be cold reads between its stores too -- C3: 2.36 -> 1.93 ms).  The loads are the point; nothing is kept.
// The chunks are c0, c0 + 8, c0 + 16, c0 + 24: chunk c is stitched by a workgroup of XCD c % 8 (launch order), and the wave reading
// ahead runs on that XCD too (touch_wave_first), so what it reads lands in the L2 the stitch wave will look in, not only in the
// memory-side cache behind it.
// stage != nullptr (round 5, STAGED descriptors): the wave also COPIES each chunk's descriptors into row (chunk index) of `stage`, 64
// slots per chunk -- the stitch wave of chunk c then loads its descriptors from stage[64 c + lane], an address it knows before its
// chunk record has arrived (one round trip less in its chain), from lines this XCD has just written; the image's own array -- dense,
// or the tiles' slots of a padded image (sir_pack.hpp) -- is only ever read here, as a stream.
__device__ __forceinline__ void touch_chunks(const uint64_t* __restrict__ desc, const Chunk* __restrict__ chunks, uint32_t c0, uint32_t n_chunks, uint64_t n_desc,
                                             const uint8_t* __restrict__ payload, uint64_t payload_len, uint32_t lane, uint64_t* __restrict__ stage = nullptr)
{
    if (c0 >= n_chunks) return;
    Chunk ch[TOUCH_CHUNKS_PER_WAVE];
#pragma unroll
    for (uint32_t k = 0; k < TOUCH_CHUNKS_PER_WAVE; ++k) ch[k] = chunks[c0 + 8u * k < n_chunks ? c0 + 8u * k : c0];
    uint64_t d[TOUCH_CHUNKS_PER_WAVE];
#pragma unroll
    for (uint32_t k = 0; k < TOUCH_CHUNKS_PER_WAVE; ++k) {
        const uint64_t n = (ch[k].dst_n >> 48) & uint64_t(CHUNK_N_MASK), tb = ch[k].task_begin & TB_IDX_MASK;
        const uint32_t n1 = (ch[k].dst_n & CHUNK_CLIP) ? uint32_t(ch[k].dst_n & CHUNK_N1_MASK) : 0u;      // (a padded rows image: sir_pack.hpp)
        const uint64_t i = chunk_desc_slot(tb, n1, lane);
        d[k] = (lane < n && i < n_desc) ? desc[i] : (uint64_t(SPACE_FILL) << 62);
    }
    if (stage) {
#pragma unroll
        for (uint32_t k = 0; k < TOUCH_CHUNKS_PER_WAVE; ++k)
            if (c0 + 8u * k < n_chunks) stage[uint64_t(c0 + 8u * k) * STAGE_SLOTS + lane] = d[k];
    }
    auto payload_lines = [&](uint64_t dd) {
        const uint64_t src = dd & SRC_MASK, len = (dd >> 40) & LEN_MASK;
        if ((dd >> 62) == SPACE_PAYLOAD && len != 0u && src < payload_len && len <= payload_len - src) {
            const uint32_t b0 = payload[src], b1 = payload[src + len - 1u];
            asm volatile("" :: "v"(b0), "v"(b1));
        }
    };
#pragma unroll
    for (uint32_t k = 0; k < TOUCH_CHUNKS_PER_WAVE; ++k) payload_lines(d[k]);
    for (uint32_t k = 0; k < TOUCH_CHUNKS_PER_WAVE; ++k) {           // chunks of more than 64 descriptors (per-block, dense images)
        const uint64_t n = (ch[k].dst_n >> 48) & uint64_t(CHUNK_N_MASK);
        const uint64_t tb = ch[k].task_begin & TB_IDX_MASK;
        for (uint64_t i = tb + 64u + lane; i < tb + n && i < n_desc; i += 64u) payload_lines(desc[i]);
    }
#pragma unroll
    for (uint32_t k = 0; k < TOUCH_CHUNKS_PER_WAVE; ++k) asm volatile("" :: "v"(d[k]));
}

}  // namespace v2p
