// wave_copy_bench.hip -- micro-benchmark of stitchw_kernel's data movement (development tool, NOT part of libvcf2prot_hip.so):
// one wave per 8 KiB of result, eight byte-granular dwordx4 gathers back to back, eight range-checked non-temporal buffer stores,
// with the source pattern, the number of extra "patch phase" gathers, a streamed descriptor read and the workgroup shape as
// run-time / template parameters.  Built by tools/wave_copy_bench.py into build_ab/libv2p_wavebench.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) unaligned16 { u32x4 v; };
__device__ __forceinline__ u32x4 gather16(uint64_t addr)
{
    typedef const __attribute__((address_space(1))) unaligned16* gptr;
    return reinterpret_cast<gptr>(addr)->v;
}

struct Params {
    const uint8_t* src; uint64_t window;      // cache-resident source window (bytes), 8 slices
    uint8_t* out; uint64_t n_chunks;          // 8 KiB per chunk
    const uint64_t* dsc;                      // descriptor stream (512 B per chunk), may be null
    uint32_t pattern;                         // 0: contiguous 8 KiB per chunk (C2), 1: 20 runs of 416 B at random places of the slice (C3), 2: one line for everybody
    uint32_t shift;                           // byte misalignment of the reads
    uint32_t n_p;                             // extra gathers before the copy (0..4): 64 scattered 16-byte reads each
    uint32_t run_blocks;                      // pattern 1: 16-byte blocks per run (26 = 416 B)
    uint32_t aligned;                         // 1: loads at 16-byte aligned addresses (shift ignored)
    uint32_t dsc_lanes;                       // lanes that read 8 bytes of the descriptor stream (64 = 512 B per chunk)
    uint32_t dsc_mod;                         // != 0: chunk c reads the descriptors of chunk c % dsc_mod (a cache-resident table)
    uint32_t aux;                             // cache policy bits of the result stores (2 = nt, 0 = plain, 16 = sc1 ...)
    uint32_t prefetch;                        // != 0: every wave also touches the descriptor lines of chunk c + prefetch (pulls them into its L2)
};

template <int WPG>
__global__ __launch_bounds__(64 * WPG, 8) void wave_copy_kernel(Params p)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wid = WPG == 1 ? 0u : uint32_t(__builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6)));
    const uint64_t c = uint64_t(blockIdx.x) * WPG + wid;
    if (c >= p.n_chunks) return;
    const uint64_t slice = (p.window / 8u) & ~4095ull;
    const uint64_t sbase = reinterpret_cast<uint64_t>(p.src) + 64u + (c & 7u) * slice;
    uint32_t extra = 0;
    uint32_t spf = 0u;
    if (p.dsc && (p.prefetch >> 31)) {                                              // scalar-cache prefetch: one dword per 64-byte line, used at the very end
        const uint64_t cn = c + (p.prefetch & 0x7FFFFFFFu);
        if (cn < p.n_chunks) {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(p.dsc + (p.dsc_mod ? cn % p.dsc_mod : cn) * 64u);
            for (uint32_t k = 0; k < p.dsc_lanes; k += 8u) spf |= q[2u * k];
        }
    } else
    if (p.dsc && p.prefetch && lane < p.dsc_lanes && (lane & 7u) == 0u) {           // one lane per 64-byte line
        const uint64_t cn = c + p.prefetch;
        if (cn < p.n_chunks) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(extra) : "v"(p.dsc + (p.dsc_mod ? cn % p.dsc_mod : cn) * 64u + lane) : "memory");
        extra &= 0u;
    }
    if (p.dsc && lane < p.dsc_lanes) extra |= uint32_t(p.dsc[(p.dsc_mod ? c % p.dsc_mod : c) * 64u + lane] & 15ull);   // (descriptor words are zero)
    extra = uint32_t(__builtin_amdgcn_readfirstlane(int(extra)));
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (uint32_t k = 0; k < p.n_p; ++k) {
        const uint32_t h = (uint32_t(c) * 2654435761u + (lane + 64u * k) * 40503u) >> 7;
        const u32x4 g = gather16(sbase + (h % uint32_t(slice - 64u)) + extra);
        acc[0] ^= g[0]; acc[1] ^= g[1]; acc[2] ^= g[2]; acc[3] ^= g[3];
    }
    uint64_t X[8];
#pragma unroll
    for (uint32_t j = 0; j < 8u; ++j) {
        const uint32_t b = j * 64u + lane;
        uint64_t a;
        if (p.pattern == 0u) a = sbase + ((c >> 3) * 8192ull) % (slice - 8192ull - 64ull) + b * 16u + p.shift;
        else if (p.pattern == 1u) {
            const uint32_t run = b / p.run_blocks, off = (b % p.run_blocks) * 16u;
            const uint32_t h = (uint32_t(c >> 3) * 2654435761u + run * 2246822519u) >> 5;
            a = sbase + (h % uint32_t(slice - 1024u)) + off;
            if (p.aligned) a &= ~15ull;
        } else a = sbase;
        X[j] = a + extra + ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u ? 1u : 0u);
    }
    u32x4 v[8];
#pragma unroll
    for (uint32_t j = 0; j < 8u; ++j) v[j] = gather16(X[j]);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out + c * 8192ull, 0, 8192, 0x00020000);
    if (spf == 0x12345678u && lane == 0u) p.out[c * 8192ull] = 1;
    if (p.aux == 0u) {
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rsrc, int(lane * 16u + j * 1024u), 0, 0);
    } else if (p.aux == 16u) {
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rsrc, int(lane * 16u + j * 1024u), 0, 16);
    } else if (p.aux == 18u) {
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rsrc, int(lane * 16u + j * 1024u), 0, 18);
    } else {
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rsrc, int(lane * 16u + j * 1024u), 0, 2);
    }
}


// ---- the dependent chain of the real kernel (chunk record -> descriptors -> gathers -> stores) and two ways of taking the cold
// misses off it: (remap) every XCD walks a contiguous eighth of the chunk table, so 8 consecutive waves of an XCD share a record
// line; (persist) resident waves loop over chunks and load the NEXT chunk's record (two ahead) and descriptors (one ahead) before
// the stores of the current one -- vmcnt retires in order, so those loads complete under the gathers and stores.
struct ChainParams {
    const uint8_t* src; uint64_t window; uint8_t* out; uint64_t n_chunks;
    const uint64_t* dsc;          // 512-byte slot per chunk
    const uint64_t* rec;          // 16 bytes per chunk: {slot index, 0}
    uint32_t pattern, shift, run_blocks, dsc_lanes;
    uint32_t remap;               // 1: block b takes chunk (b & 7) * ceil(n/8) + (b >> 3)
    uint32_t per_xcd;
};

template <int PATTERN>
__device__ __forceinline__ void chain_addresses(const ChainParams& p, uint64_t c, uint32_t lane, uint32_t extra, uint64_t (&X)[8])
{
    const uint32_t slice = uint32_t(p.window / 8u) & ~4095u;
    const uint64_t sbase = reinterpret_cast<uint64_t>(p.src) + 64u + (c & 7u) * uint64_t(slice);
#pragma unroll
    for (uint32_t j = 0; j < 8u; ++j) {
        const uint32_t b = j * 64u + lane;
        uint32_t a;
        if (PATTERN == 0) a = (uint32_t(c >> 3) * 8192u) % (slice - 8192u - 64u) + b * 16u + p.shift;
        else {
            const uint32_t run = b / 26u, off = (b % 26u) * 16u;
            const uint32_t h = (uint32_t(c >> 3) * 2654435761u + run * 2246822519u) >> 5;
            a = (h % (slice - 1024u)) + off;
        }
        X[j] = sbase + a + extra;
    }
}

template <int PATTERN>
__global__ __launch_bounds__(64, 8) void wave_chain_kernel(ChainParams p)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t b = blockIdx.x;
    const uint64_t c = p.remap ? (b & 7u) * uint64_t(p.per_xcd) + (b >> 3) : b;
    if (c >= p.n_chunks) return;
    const uint64_t slot = p.rec[2u * c];
    uint32_t extra = uint32_t(p.dsc[slot * 64u + (lane < p.dsc_lanes ? lane : 0u)] & 15ull);
    extra = uint32_t(__builtin_amdgcn_readfirstlane(int(extra)));
    uint64_t X[8];
    chain_addresses<PATTERN>(p, c, lane, extra, X);
    u32x4 v[8];
#pragma unroll
    for (uint32_t j = 0; j < 8u; ++j) v[j] = gather16(X[j]);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out + c * 8192ull, 0, 8192, 0x00020000);
#pragma unroll
    for (uint32_t j = 0; j < 8u; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rsrc, int(lane * 16u + j * 1024u), 0, 2);
}

template <int PATTERN>
__global__ __launch_bounds__(64, 8) void wave_chain_persist_kernel(ChainParams p)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t S = gridDim.x;
    uint64_t c = blockIdx.x;
    if (c >= p.n_chunks) return;
    const uint32_t dl = lane < p.dsc_lanes ? lane : 0u;
    uint64_t r1 = p.rec[2u * (c + S < p.n_chunks ? c + S : c)];          // record of the next chunk
    uint64_t d = p.dsc[p.rec[2u * c] * 64u + dl];                         // descriptors of this one
    asm volatile("" : "+v"(d), "+v"(r1));       // nothing in flight at the loop's entry: the back edge's state (next descriptors, stores) sets the waits
    for (;;) {
        const uint64_t c2 = c + 2u * S < p.n_chunks ? c + 2u * S : c;
        const uint64_t r2 = __hip_atomic_load(p.rec + 2u * c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // in flight until the next iteration's middle
        uint32_t extra = uint32_t(__builtin_amdgcn_readfirstlane(int(uint32_t(d & 15ull))));
        uint64_t X[8];
        chain_addresses<PATTERN>(p, c, lane, extra, X);
        u32x4 v[8];
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) v[j] = gather16(X[j]);
        // next chunk's descriptors: retire under the stores (a relaxed atomic load: the compiler neither sinks it to its use in the
        // next iteration nor moves the stores above it)
        const uint64_t dn = __hip_atomic_load(p.dsc + r1 * 64u + dl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(p.out + c * 8192ull, 0, 8192, 0x00020000);
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) __builtin_amdgcn_raw_buffer_store_b128(v[j], rsrc, int(lane * 16u + j * 1024u), 0, 2);
        c += S;
        if (c >= p.n_chunks) break;
        d = dn; r1 = r2;
    }
}

extern "C" __attribute__((visibility("default"))) int v2p_bench_wave_chain(void* stream, const uint8_t* src, uint64_t window, uint8_t* out, uint64_t bytes, const uint64_t* dsc,
                                    const uint64_t* rec, uint32_t pattern, uint32_t dsc_lanes, uint32_t remap, uint32_t persist_waves)
{
    ChainParams p{src, window, out, bytes / 8192u, dsc, rec, pattern, 5u, 26u, dsc_lanes, remap, 0u};
    if (!p.n_chunks || window < (1u << 20) || !dsc || !rec) return -1;
    p.per_xcd = uint32_t((p.n_chunks + 7) / 8);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -3;
    (void)hipGetLastError();
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(persist_waves ? persist_waves : (remap ? p.per_xcd * 8u : uint32_t(p.n_chunks)));
    if (persist_waves && pattern == 0u) hipLaunchKernelGGL(wave_chain_persist_kernel<0>, grid, dim3(64), 0, s, p);
    else if (persist_waves) hipLaunchKernelGGL(wave_chain_persist_kernel<1>, grid, dim3(64), 0, s, p);
    else if (pattern == 0u) hipLaunchKernelGGL(wave_chain_kernel<0>, grid, dim3(64), 0, s, p);
    else hipLaunchKernelGGL(wave_chain_kernel<1>, grid, dim3(64), 0, s, p);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -100 - int(e);
}

// reads `bytes` from `p` (16 bytes per lane, grid-stride) and keeps nothing: pulls a range into the memory-side cache
template <bool NT>
__global__ __launch_bounds__(256) void touch_kernel(const u32x4* __restrict__ p, uint64_t n16, uint32_t* sink)
{
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (uint64_t i = uint64_t(blockIdx.x) * 256u + threadIdx.x; i < n16; i += uint64_t(gridDim.x) * 256u) {
        const u32x4 v = NT ? __builtin_nontemporal_load(p + i) : p[i];
        acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1u;
}

// the copy in `phases` sub-launches, each preceded by a kernel that reads its descriptors (dsc_lanes * 8 bytes of every 512-byte slot
// are what the copy reads; the touch reads whole slots) -- descriptor reads then never mix with the result stores
extern "C" __attribute__((visibility("default"))) int v2p_bench_wave_copy_phased(void* stream, const uint8_t* src, uint64_t window, uint8_t* out, uint64_t bytes, const uint64_t* dsc,
                                          uint32_t pattern, uint32_t dsc_lanes, uint32_t aux, uint32_t phases, uint32_t* sink, int touch)
{
    const uint64_t n_chunks = bytes / 8192u;
    if (!n_chunks || !phases || !dsc) return -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -3;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const uint64_t per = ((n_chunks + phases - 1) / phases + 7) & ~7ull;
    for (uint64_t c0 = 0; c0 < n_chunks; c0 += per) {
        const uint64_t n = n_chunks - c0 < per ? n_chunks - c0 : per;
        if (touch == 1) hipLaunchKernelGGL(touch_kernel<true>, dim3(2048), dim3(256), 0, s, reinterpret_cast<const u32x4*>(dsc + c0 * 64u), n * 32u, sink);
        if (touch == 2) hipLaunchKernelGGL(touch_kernel<false>, dim3(2048), dim3(256), 0, s, reinterpret_cast<const u32x4*>(dsc + c0 * 64u), n * 32u, sink);
        Params p{src, window, out + c0 * 8192ull, n, dsc + c0 * 64u, pattern, 5u, 0u, 26u, 0u, dsc_lanes, 0u, aux, 0u};
        hipLaunchKernelGGL(wave_copy_kernel<1>, dim3(uint32_t(n)), dim3(64), 0, s, p);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -100 - int(e);
}

extern "C" __attribute__((visibility("default"))) int v2p_bench_wave_copy(void* stream, const uint8_t* src, uint64_t window, uint8_t* out, uint64_t bytes, const uint64_t* dsc,
                                   uint32_t pattern, uint32_t shift, uint32_t n_p, uint32_t run_blocks, uint32_t aligned, int wpg,
                                   uint32_t dsc_lanes, uint32_t dsc_mod, uint32_t aux, uint32_t prefetch)
{
    Params p{src, window, out, bytes / 8192u, dsc, pattern, shift, n_p, run_blocks ? run_blocks : 26u, aligned, dsc_lanes, dsc_mod, aux, prefetch};
    if (!p.n_chunks || window < (1u << 20)) return -1;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -3;
    (void)hipGetLastError();
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (wpg == 4) hipLaunchKernelGGL(wave_copy_kernel<4>, dim3(uint32_t((p.n_chunks + 3) / 4)), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(wave_copy_kernel<1>, dim3(uint32_t(p.n_chunks)), dim3(64), 0, s, p);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -100 - int(e);
}
