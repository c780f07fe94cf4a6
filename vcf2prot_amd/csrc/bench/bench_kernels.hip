// bench_kernels.hip -- micro-benchmarks of the stitch kernels' data movement (development tools: tools/copy_bench.py, copy_mix.py,
// hbm_ceiling.py, gather_ceiling.py).  NOT part of libvcf2prot_hip.so and not declared in include/: built into
// vcf2prot_amd/lib/libv2p_bench.so together with a V2P_BENCH_VARIANTS build of the engine (timing-only ablations of the kernels).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../stitch_kernels.h"
#include "../stitch_device.hpp"
#include "v2p_bench.h"

namespace v2p {

// ---------------------------------------------------------------------------
// fill_kernel: 16-byte streaming stores; used to measure the write ceiling the
// stitch kernel is compared against (profiles/).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fill_kernel(uint8_t* out, uint64_t n16, uint32_t word, int nt, uint32_t span16)
{
    const u32x4 v = {word, word, word, word};
    u32x4* o = reinterpret_cast<u32x4*>(out);
    if (span16 == 0) {     // grid-stride
        for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += uint64_t(gridDim.x) * blockDim.x) {
            if (nt) __builtin_nontemporal_store(v, o + i); else o[i] = v;
        }
    } else {               // one workgroup per contiguous span (the stitch kernel's store pattern)
        const uint64_t b = uint64_t(blockIdx.x) * span16;
        const uint64_t e = b + span16 < n16 ? b + span16 : n16;
        for (uint64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
            if (nt) __builtin_nontemporal_store(v, o + i); else o[i] = v;
        }
    }
}

// gather_bench_kernel: 16-byte-per-lane gathers from an L2-resident window at a chosen misalignment
// (address path / L1 ceiling the stitch kernel's gathers are compared against).
__global__ __launch_bounds__(256) void gather_bench_kernel(const uint8_t* __restrict__ src, uint64_t window, uint32_t misalign,
                                                           uint32_t iters, uint32_t* __restrict__ sink)
{
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    u32x4 acc = {0u, 0u, 0u, 0u};
    uint64_t off = (uint64_t(wave) * 4096ull) % window;
    for (uint32_t i = 0; i < iters; i += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t o = (off + uint64_t(u) * 1024ull + lane * 16u) % (window - 64u);
            const u32x4 v = gather16(reinterpret_cast<uint64_t>(src) + o + misalign);
            acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3];
        }
        off = (off + 4096ull * 17ull) % window;
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[wave] = acc[0];
}

// copy_bench_kernel: the stitch kernel's data movement with all bookkeeping removed -- one workgroup per 32 KiB span of
// `out`, 1 KiB per wave and pass, source = a cache-resident window (slice b % 8 for workgroup b, as the XCD-aware chunk
// order arranges) read at byte misalignment `shift`, streamed out with non-temporal 16-byte stores.  Ceiling the K2 phase is
// compared against, per load flavour:
//   0 byte-granular dwordx4 gather   1 aligned dwordx4 + next lane's block (DPP wave_shl:1) + funnel shift
//   2 two aligned dwordx4 + funnel   3 dword-aligned x4 + x1 + v_alignbyte   4 stores only   5 loads only (mode 1 loads)
template <int MODE>
__global__ __launch_bounds__(256) void copy_bench_kernel(const uint8_t* __restrict__ src, uint64_t window, uint32_t shift_delay,
                                                         uint8_t* __restrict__ out, uint64_t n16, uint32_t* __restrict__ sink)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t shift = shift_delay & 15u, delay = shift_delay >> 4;     // delay: cycles every wave idles before its first load
    const uint64_t slice = (window / 8u) & ~4095ull;
    const uint64_t span = 2048u;                                            // 16-byte blocks per workgroup
    const uint64_t b0 = uint64_t(blockIdx.x) * span;
    const uint64_t e = b0 + span < n16 ? b0 + span : n16;
    const uint64_t base = reinterpret_cast<uint64_t>(src) + 64u + (blockIdx.x & 7u) * slice
                        + ((uint64_t(blockIdx.x >> 3) * span * 16u) % (slice - span * 16u - 64u));
    if (delay) {                                                            // stands for the stitch kernel's per-chunk set-up
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < delay) __builtin_amdgcn_s_sleep(8);
    }
    uint64_t extra = 0;
    if (MODE == 8 || MODE == 9) {
        // the stitch kernel's descriptor traffic: 8 bytes per lane streamed from HBM (read once, never cached), used before the
        // first store; mode 9 first reads a 16-byte header whose value the descriptor address depends on (two dependent misses)
        const uint64_t* dsc = reinterpret_cast<const uint64_t*>(out + ((n16 * 16u + 4095u) & ~4095ull));
        uint64_t off = uint64_t(blockIdx.x) * 256u + threadIdx.x;
        if (MODE == 9) {
            const uint64_t* hdr = dsc + ((n16 + 2047u) / 2048u) * 256u + 2u * blockIdx.x;
            off += (hdr[0] | hdr[1]) & 1u;                                  // header words are zero
        }
        extra = dsc[off] & 15u;                                             // descriptor words are zero too
    }
    u32x4 acc = {0u, 0u, 0u, 0u};
    if (MODE == 6 || MODE == 7) {
        // two aligned loads per block; 6: load -> store per pass, 7: the next pass's loads are issued before this pass's store,
        // so the wait for them (vmcnt(1)) leaves the store in flight
        if (b0 + threadIdx.x >= e) return;
        uint64_t b = b0 + threadIdx.x;
        uint64_t X = base + ((b - b0) << 4) + shift;
        uint32_t d = uint32_t(X) & 15u;
        u32x4 lo = load16_aligned(X - d), hi = load16_aligned(X - d + 16u);
        if (MODE == 7) {
            u32x4 v = funnel16(lo, hi, d);                                  // peeled first pass
            uint64_t bn = b + 256u;
            const bool more = bn < e;
            if (more) { X = base + ((bn - b0) << 4) + shift; d = uint32_t(X) & 15u; lo = load16_aligned(X - d); hi = load16_aligned(X - d + 16u); }
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out) + b);
            if (!more) return;
            b = bn;
        }
#pragma unroll 1
        for (;;) {
            const u32x4 v = funnel16(lo, hi, d);
            const uint64_t bn = b + 256u;
            const bool more = bn < e;
            if (MODE == 7 && more) { X = base + ((bn - b0) << 4) + shift; d = uint32_t(X) & 15u; lo = load16_aligned(X - d); hi = load16_aligned(X - d + 16u); }
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out) + b);
            if (!more) break;
            if (MODE == 6) { X = base + ((bn - b0) << 4) + shift; d = uint32_t(X) & 15u; lo = load16_aligned(X - d); hi = load16_aligned(X - d + 16u); }
            b = bn;
        }
        return;
    }
#pragma unroll 1
    for (uint64_t b = b0 + threadIdx.x; b < e; b += 256u) {
        const uint64_t X = base + ((b - b0) << 4) + shift + extra;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (MODE == 0 || MODE == 8 || MODE == 9) v = gather16(X);
        else if (MODE == 3) v = gather16_dw(X);
        else if (MODE == 2) {
            const uint32_t d = uint32_t(X) & 15u;
            const u32x4 lo = load16_aligned(X - d), hi = load16_aligned(X - d + 16u);
            v = funnel16(lo, hi, d);
        } else if (MODE == 1 || MODE == 5) {
            const uint32_t d = uint32_t(X) & 15u;
            const u32x4 lo = load16_aligned(X - d);
            u32x4 own = {0u, 0u, 0u, 0u};
            if (lane == 63u && d != 0u) own = load16_aligned(X - d + 16u);
            u32x4 nx;
            nx[0] = from_next_lane(0u, lo[0]); nx[1] = from_next_lane(0u, lo[1]);
            nx[2] = from_next_lane(0u, lo[2]); nx[3] = from_next_lane(0u, lo[3]);
            if (lane == 63u) nx = own;
            v = funnel16(lo, nx, d);
        } else v = u32x4{uint32_t(b), shift, 0u, 0u};
        if (MODE == 5) { acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3]; }
        else __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out) + b);
    }
    if (MODE == 5 && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[blockIdx.x] = acc[0];
}

// copy_prefetch_kernel: persistent workgroups (grid = resident set); span k of a workgroup uses a "descriptor" (8 B per lane,
// streamed from HBM) that was requested DEPTH spans earlier.  BYTES: descriptor bytes per lane actually loaded (8, 4 or 0).
template <int DEPTH, int BYTES>
__global__ __launch_bounds__(256) void copy_prefetch_kernel(const uint8_t* __restrict__ src, uint64_t window, uint32_t shift,
                                                            uint8_t* __restrict__ out, uint64_t n16)
{
    const uint64_t slice = (window / 8u) & ~4095ull;
    const uint64_t span = 2048u;
    const uint64_t n_span = (n16 + span - 1) / span;
    const uint8_t* dsc = out + ((n16 * 16u + 4095u) & ~4095ull);
    uint64_t ring[DEPTH];
    auto fetch = [&](uint64_t sp) -> uint64_t {
        if (sp >= n_span || BYTES == 0) return 0ull;
        if (BYTES == 8) return reinterpret_cast<const uint64_t*>(dsc)[sp * 256u + threadIdx.x];
        return uint64_t(reinterpret_cast<const uint32_t*>(dsc)[sp * 256u + threadIdx.x]);
    };
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) ring[k] = fetch(uint64_t(blockIdx.x) + uint64_t(k) * gridDim.x);
    for (uint64_t sp = blockIdx.x; sp < n_span; sp += uint64_t(gridDim.x) * DEPTH) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            const uint64_t cur = sp + uint64_t(k) * gridDim.x;
            if (cur >= n_span) break;
            const uint64_t extra = ring[k] & 15u;                                  // descriptor words are zero
            ring[k] = fetch(cur + uint64_t(DEPTH) * gridDim.x);                    // request the descriptor DEPTH spans ahead
            const uint64_t b0 = cur * span;
            const uint64_t e = b0 + span < n16 ? b0 + span : n16;
            const uint64_t base = reinterpret_cast<uint64_t>(src) + 64u + (cur & 7u) * slice + (((cur >> 3) * span * 16u) % (slice - span * 16u - 64u));
#pragma unroll 1
            for (uint64_t b = b0 + threadIdx.x; b < e; b += 256u) {
                const u32x4 v = gather16(base + ((b - b0) << 4) + shift + extra);
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out) + b);
            }
        }
    }
}

// copy_mix_kernel: non-persistent copy (one workgroup per 32 KiB span) whose workgroups additionally stream "descriptor"
// bytes from HBM; everything about that read stream is a run-time parameter:
//   p.x bytes per lane (4, 8, 16)   p.y every p.y-th workgroup reads (1 = all)   p.z bytes between two workgroups' pieces
//   p.w bit 0: non-temporal loads; bit 1: lanes 0..63 only; bit 2: the read is issued after the first pass's store instead of before
__global__ __launch_bounds__(256) void copy_mix_kernel(const uint8_t* __restrict__ src, uint64_t window, uint32_t shift,
                                                       uint8_t* __restrict__ out, uint64_t n16, const uint8_t* __restrict__ dsc, uint4 p)
{
    const uint64_t slice = (window / 8u) & ~4095ull;
    const uint64_t span = 2048u;
    uint64_t sp = blockIdx.x;
    if (p.w & 8u) {
        // the stitch kernel's XCD-aware order: workgroup 8j + x takes the j-th span of "proteome slice" x; a "haplotype" is
        // 240 spans (7.5 MiB), a slice 30 of them
        const uint64_t x = sp & 7u, j = sp >> 3;
        const uint64_t perm = (j / 30u) * 240u + x * 30u + (j % 30u);
        if (perm < (n16 + span - 1) / span && ((n16 + span - 1) / span) % 240u == 0u) sp = perm;
    }
    const uint64_t b0 = sp * span;
    const uint64_t e = b0 + span < n16 ? b0 + span : n16;
    const uint64_t base = reinterpret_cast<uint64_t>(src) + 64u + (blockIdx.x & 7u) * slice
                        + ((uint64_t(blockIdx.x >> 3) * span * 16u) % (slice - span * 16u - 64u));
    uint64_t extra = 0;
    const uint32_t n_lanes = (p.y >> 16) ? (p.y >> 16) : 256u;               // p.y bits 16..: lanes that read (0 = all)
    p.y &= 0xFFFFu;
    const bool reader = p.x != 0u && (blockIdx.x % p.y) == 0u && (!(p.w & 2u) || threadIdx.x < 64u) && threadIdx.x < n_lanes;
    auto rd = [&]() {
        const uint8_t* q = dsc + uint64_t(blockIdx.x / p.y) * (p.z & 0xFFFFu) + uint64_t(threadIdx.x) * p.x;
        uint32_t x;
        if (p.x == 4u) x = (p.w & 1u) ? __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(q)) : *reinterpret_cast<const uint32_t*>(q);
        else if (p.x == 8u) { const uint64_t y = (p.w & 1u) ? __builtin_nontemporal_load(reinterpret_cast<const uint64_t*>(q)) : *reinterpret_cast<const uint64_t*>(q); x = uint32_t(y) | uint32_t(y >> 32); }
        else { const u32x4 y = (p.w & 1u) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(q)) : *reinterpret_cast<const u32x4*>(q); x = y[0] | y[1] | y[2] | y[3]; }
        return uint64_t(x & 15u);
    };
    if (reader && !(p.w & 4u)) extra = rd();
    bool first = true;
    // ballast that stands for the stitch kernel's bookkeeping: p.w bits 8..15 = VALU instructions (x8) per pass and wave,
    // bits 16..23 = LDS round trips per pass, bits 24..27 = workgroup barriers before the first pass
    const uint32_t n_valu = ((p.w >> 8) & 0xFFu), n_lds = (p.w >> 16) & 0xFFu, n_bar = (p.w >> 24) & 0xFu;
    __shared__ uint32_t s_ballast[1024];
    s_ballast[threadIdx.x] = threadIdx.x; s_ballast[threadIdx.x + 256u] = threadIdx.x + 1u;
    s_ballast[threadIdx.x + 512u] = threadIdx.x + 2u; s_ballast[threadIdx.x + 768u] = threadIdx.x + 3u;
    for (uint32_t k = 0; k < n_bar; ++k) __syncthreads();
    uint32_t acc = threadIdx.x;
    if (p.w & 16u) {
        // the stitch4 order: all eight rows of a wave gathered first, then stored back to back
        u32x4 v[8];
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) {
            const uint64_t b = b0 + threadIdx.x + 256u * j;
            v[j] = gather16(base + ((b - b0) << 4) + shift + extra);
        }
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) {
            const uint64_t b = b0 + threadIdx.x + 256u * j;
            if (b < e) __builtin_nontemporal_store(v[j], reinterpret_cast<u32x4*>(out) + b);
        }
        return;
    }
#pragma unroll 1
    for (uint64_t b = b0 + threadIdx.x; b < e; b += 256u) {
#pragma unroll 1
        for (uint32_t k = 0; k < n_lds; ++k) acc = s_ballast[acc & 1023u] + uint32_t(b);          // dependent LDS reads
#pragma unroll 1
        for (uint32_t k = 0; k < n_valu; ++k) {
            acc = acc * 3u + 1u; acc ^= acc >> 3; acc += uint32_t(b); acc = (acc << 1) | (acc >> 31);
            acc ^= 0x9E3779B9u; acc += acc >> 5; acc ^= acc << 7; acc += 11u;
        }
        u32x4 v = gather16(base + ((b - b0) << 4) + shift + extra + (acc == 0xFFFFFFFFu ? 1u : 0u));
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out) + b);
        if (first && reader && (p.w & 4u)) extra = rd();
        first = false;
    }
}

hipError_t launch_copy_mix(const uint8_t* src, uint64_t window, uint32_t shift, uint8_t* out, uint64_t bytes, const uint8_t* dsc,
                           uint32_t bytes_per_lane, uint32_t every, uint32_t stride, uint32_t flags, hipStream_t stream)
{
    const uint64_t n16 = bytes / 16;
    if (n16 == 0 || window < (1u << 20) || (every & 0xFFFFu) == 0) return hipErrorInvalidValue;
    const uint32_t grid = uint32_t((n16 + 2047) / 2048);
    const uint32_t lds_pad = (flags >> 8) * 1024u;                   // flags bits 8..: KiB of (unused) dynamic LDS per workgroup, to cap the workgroups per CU
    // stride bits 16..: ballast (VALU x8 per pass: 8 bits, LDS round trips per pass: 8 bits, barriers: 4 bits)
    hipLaunchKernelGGL(copy_mix_kernel, dim3(grid), dim3(256), lds_pad, stream, src, window, shift, out, n16, dsc, make_uint4(bytes_per_lane, every, stride, (flags & 0xFFu) | (stride >> 16 << 8)));
    return hipGetLastError();
}

hipError_t launch_copy_prefetch(const uint8_t* src, uint64_t window, uint32_t shift, uint8_t* out, uint64_t bytes, int depth, int desc_bytes,
                                uint32_t grid, hipStream_t stream)
{
    const uint64_t n16 = bytes / 16;
    if (n16 == 0 || window < (1u << 20)) return hipErrorInvalidValue;
#define V2P_CP(D, B) hipLaunchKernelGGL((copy_prefetch_kernel<D, B>), dim3(grid), dim3(256), 0, stream, src, window, shift, out, n16)
    if (desc_bytes == 0) V2P_CP(1, 0);
    else if (desc_bytes == 4) { if (depth <= 1) V2P_CP(1, 4); else V2P_CP(4, 4); }
    else if (depth <= 1) V2P_CP(1, 8);
    else if (depth == 2) V2P_CP(2, 8);
    else if (depth <= 4) V2P_CP(4, 8);
    else V2P_CP(8, 8);
#undef V2P_CP
    return hipGetLastError();
}

hipError_t launch_copy_bench(const uint8_t* src, uint64_t window, uint32_t shift, uint8_t* out, uint64_t bytes, int mode,
                             uint32_t* sink, hipStream_t stream)
{
    const uint64_t n16 = bytes / 16;
    if (n16 == 0 || window < (1u << 20)) return hipErrorInvalidValue;
    const uint32_t grid = uint32_t((n16 + 2047) / 2048);
#define V2P_CB(M) hipLaunchKernelGGL((copy_bench_kernel<M>), dim3(grid), dim3(256), 0, stream, src, window, shift, out, n16, sink)
    switch (mode) {
        case 0: V2P_CB(0); break;
        case 1: V2P_CB(1); break;
        case 2: V2P_CB(2); break;
        case 3: V2P_CB(3); break;
        case 4: V2P_CB(4); break;
        case 5: V2P_CB(5); break;
        case 6: V2P_CB(6); break;
        case 7: V2P_CB(7); break;
        case 8: V2P_CB(8); break;
        default: V2P_CB(9); break;
    }
#undef V2P_CB
    return hipGetLastError();
}

hipError_t launch_gather_bench(const uint8_t* src, uint64_t window, uint32_t misalign, uint32_t iters, uint32_t blocks,
                               uint32_t* sink, hipStream_t stream)
{
    hipLaunchKernelGGL(gather_bench_kernel, dim3(blocks), dim3(256), 0, stream, src, window, misalign, iters, sink);
    return hipGetLastError();
}

static inline uint32_t grid_for(uint64_t work_items, uint32_t cap)
{
    return uint32_t(work_items < cap ? (work_items ? work_items : 1) : cap);
}

hipError_t launch_fill(uint8_t* out, uint64_t bytes, uint32_t word, int nontemporal, hipStream_t stream)
{
    const uint64_t n16 = bytes / 16;
    if (n16 == 0) return hipSuccess;
    const uint32_t span16 = uint32_t(nontemporal >> 8);          // bits 8..: 16-byte blocks per workgroup (0 = grid-stride)
    const int nt = nontemporal & 1;
    if (span16) hipLaunchKernelGGL(fill_kernel, dim3(uint32_t((n16 + span16 - 1) / span16)), dim3(256), 0, stream, out, n16, word, nt, span16);
    else hipLaunchKernelGGL(fill_kernel, dim3(grid_for((n16 + 255) / 256, 256u * 8u)), dim3(256), 0, stream, out, n16, word, nt, 0u);
    return hipGetLastError();
}


}  // namespace v2p

extern "C" {

int v2p_gather_bench_launch(void* hip_stream, const uint8_t* d_src, uint64_t window, uint32_t misalign, uint32_t iters,
                            uint32_t blocks, uint32_t* d_sink)
{
    return v2p::launch_gather_bench(d_src, window, misalign, iters, blocks, d_sink, reinterpret_cast<hipStream_t>(hip_stream)) == hipSuccess ? 0 : -2;
}

int v2p_copy_bench_launch(void* hip_stream, const uint8_t* d_src, uint64_t window, uint32_t shift, uint8_t* d_out, uint64_t bytes,
                          int mode, uint32_t* d_sink)
{
    return v2p::launch_copy_bench(d_src, window, shift, d_out, bytes, mode, d_sink, reinterpret_cast<hipStream_t>(hip_stream)) == hipSuccess ? 0 : -2;
}

int v2p_copy_prefetch_launch(void* hip_stream, const uint8_t* d_src, uint64_t window, uint32_t shift, uint8_t* d_out, uint64_t bytes,
                             int depth, int desc_bytes, uint32_t grid)
{
    return v2p::launch_copy_prefetch(d_src, window, shift, d_out, bytes, depth, desc_bytes, grid, reinterpret_cast<hipStream_t>(hip_stream)) == hipSuccess ? 0 : -2;
}

int v2p_copy_mix_launch(void* hip_stream, const uint8_t* d_src, uint64_t window, uint32_t shift, uint8_t* d_out, uint64_t bytes,
                        const uint8_t* d_desc, uint32_t bytes_per_lane, uint32_t every, uint32_t stride, uint32_t flags)
{
    return v2p::launch_copy_mix(d_src, window, shift, d_out, bytes, d_desc, bytes_per_lane, every, stride, flags, reinterpret_cast<hipStream_t>(hip_stream)) == hipSuccess ? 0 : -2;
}

int v2p_fill_launch(void* hip_stream, uint8_t* d_out, uint64_t bytes, uint32_t word, int nontemporal)
{
    return v2p::launch_fill(d_out, bytes, word, nontemporal, reinterpret_cast<hipStream_t>(hip_stream)) == hipSuccess ? 0 : -2;
}

}  // extern "C"
