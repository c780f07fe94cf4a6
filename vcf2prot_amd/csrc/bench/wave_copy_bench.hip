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

// reads `bytes` from `p` (16 bytes per lane, grid-stride) and keeps nothing: pulls a range into the memory-side cache
__global__ __launch_bounds__(256) void touch_kernel(const u32x4* __restrict__ p, uint64_t n16, uint32_t* sink)
{
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (uint64_t i = uint64_t(blockIdx.x) * 256u + threadIdx.x; i < n16; i += uint64_t(gridDim.x) * 256u) {
        const u32x4 v = __builtin_nontemporal_load(p + i);
        acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1u;
}

// the copy in `phases` sub-launches, each preceded by a kernel that reads its descriptors (dsc_lanes * 8 bytes of every 512-byte slot
// are what the copy reads; the touch reads whole slots) -- descriptor reads then never mix with the result stores
extern "C" int v2p_bench_wave_copy_phased(void* stream, const uint8_t* src, uint64_t window, uint8_t* out, uint64_t bytes, const uint64_t* dsc,
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
        if (touch) hipLaunchKernelGGL(touch_kernel, dim3(2048), dim3(256), 0, s, reinterpret_cast<const u32x4*>(dsc + c0 * 64u), n * 32u, sink);
        Params p{src, window, out + c0 * 8192ull, n, dsc + c0 * 64u, pattern, 5u, 0u, 26u, 0u, dsc_lanes, 0u, aux, 0u};
        hipLaunchKernelGGL(wave_copy_kernel<1>, dim3(uint32_t(n)), dim3(64), 0, s, p);
    }
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : -100 - int(e);
}

extern "C" int v2p_bench_wave_copy(void* stream, const uint8_t* src, uint64_t window, uint8_t* out, uint64_t bytes, const uint64_t* dsc,
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
