// stitch_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the SIR executor.
//
// Replaces the reference's step-6 loop
//     for task in g_rep { res[dst..dst+len] = (code==0 ? ref : alt)[src..src+len] }
// (task.rs:38-50, gir.rs:230-234) for a whole batch of haplotypes per launch.
// Integer/byte work only: HBM-bound gather/scatter, no MFMA.
//
// Kernels
//   stitch_kernel   K2 (+K0 fused, +in-chunk K1): one 256-lane workgroup per chunk of
//                   <=256 descriptors / <=64 KiB of result.  Lanes load one 8-byte
//                   descriptor each (coalesced), a wave64 DPP prefix scan + a 4-entry
//                   LDS carry turn lengths into result offsets, then every lane owns
//                   16-byte aligned result blocks: it finds the covering task by a
//                   branch-free search of the offsets in LDS, gathers 16 source bytes
//                   with one unaligned dwordx4 load per overlapping task and merges them
//                   with v_bfi byte masks, and writes one aligned dwordx4 store.  Result
//                   stores are therefore full 16-byte, fully coalesced (1 KiB per wave
//                   instruction) whatever the source alignments are.
//   ordered_kernel  reference-order execution for non-canonical Task vectors
//                   (overlapping / descending result ranges): one workgroup, tasks in
//                   order, barrier between tasks => "later task wins" as on the CPU.
//   validate_kernel DEBUG_GPU: first row that breaks code / bounds / the contiguity
//                   predicate of gir.rs:208-226 (wave ballot + one atomicMin per wave).
//   digest_kernel   per-haplotype position-sensitive checksum of the result arena.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "stitch_kernels.h"

namespace v2p {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((packed, aligned(1))) unaligned16 { u32x4 v; };

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

__device__ __forceinline__ uint32_t byte_mask_below(uint32_t k)  // bytes [0,k) of a dword, k in 0..4
{
    return k >= 4u ? 0xFFFFFFFFu : ((1u << (8u * k)) - 1u);
}

// v[j] = ld[j] for bytes j in [a,b) of the 16-byte block
__device__ __forceinline__ u32x4 merge_bytes(u32x4 v, u32x4 ld, uint32_t a, uint32_t b)
{
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        uint32_t lo = a > 4u * w ? a - 4u * w : 0u;
        uint32_t hi = b > 4u * w ? b - 4u * w : 0u;
        uint32_t m = byte_mask_below(hi) & ~byte_mask_below(lo);
        v[w] = (v[w] & ~m) | (ld[w] & m);
    }
    return v;
}

__device__ __forceinline__ void report(unsigned long long* status, uint64_t index, uint32_t reason)
{
    atomicMin(status, (unsigned long long)((index << 8) | reason));
}

template <bool NT>
__global__ __launch_bounds__(256) void stitch_kernel(StitchArgs a)
{
    __shared__ uint64_t s_desc[256];
    __shared__ uint32_t s_off[257];
    __shared__ uint32_t s_wsum[4];

    const uint32_t tid = threadIdx.x, lane = tid & 63u, wid = tid >> 6;
    const uint8_t* const base0 = a.src0;
    const uint8_t* const base1 = a.src1;
    const u32x4 dots = {0x2E2E2E2Eu, 0x2E2E2E2Eu, 0x2E2E2E2Eu, 0x2E2E2E2Eu};

    for (uint32_t c = blockIdx.x; c < a.n_chunks; c += gridDim.x) {
        const uint64_t tb = a.chunks[c].task_begin;
        const uint64_t dn = a.chunks[c].dst_n;
        const uint32_t n = uint32_t(dn >> 48);
        const uint64_t dst = dn & ((1ull << 48) - 1);

        // ---- K1: descriptors -> result offsets (one descriptor per lane) ----
        uint64_t d = uint64_t(SPACE_FILL) << 62;
        uint32_t len = 0;
        if (tid < n) {
            d = a.desc[tb + tid];
            len = uint32_t(d >> 40) & ((1u << 22) - 1u);
            const uint32_t space = uint32_t(d >> 62);
            const uint64_t src = d & ((1ull << 40) - 1);
            const uint64_t limit = space == SPACE_PROTEOME ? a.src0_len : (space == SPACE_PAYLOAD ? a.src1_len : ~0ull);
            if (space == 3u || src + len > limit) {          // never read out of bounds: task.rs would panic
                report(a.status, tb + tid, STATUS_SRC_OOB);
                d = (d & ~(3ull << 62)) | (uint64_t(SPACE_FILL) << 62);
            }
        }
        const uint32_t incl = wave_incl_scan(len);
        if (lane == 63u) s_wsum[wid] = incl;
        __syncthreads();
        const uint32_t w0 = s_wsum[0], w1 = s_wsum[1], w2 = s_wsum[2], w3 = s_wsum[3];
        const uint32_t wave_base = (wid > 0 ? w0 : 0u) + (wid > 1 ? w1 : 0u) + (wid > 2 ? w2 : 0u);
        const uint32_t excl = incl - len + wave_base;
        const uint32_t total = w0 + w1 + w2 + w3;
        s_desc[tid] = d;
        s_off[tid] = excl;
        if (tid == 255u) s_off[256] = total;
        __syncthreads();

        if (dst + total > a.out_len) {                        // never write out of bounds
            if (tid == 0) report(a.status, tb, STATUS_RES_OOB);
        } else {
            // ---- K2: every lane owns 16-byte aligned blocks of the chunk's result range ----
            const uint32_t head = uint32_t(dst & 15ull);
            const uint32_t nblk = (head + total + 15u) >> 4;
            uint8_t* const out0 = a.out + (dst - head);
            for (uint32_t b = tid; b < nblk; b += 256u) {
                const int32_t rel = int32_t(b << 4) - int32_t(head);          // block start relative to dst
                const uint32_t lo = rel < 0 ? 0u : uint32_t(rel);
                const uint32_t hi = uint32_t(rel + 16) < total ? uint32_t(rel + 16) : total;
                // covering task: largest i with s_off[i] <= lo (zero-length tasks share an offset
                // with their successor, so the last of equal offsets is the non-empty one)
                uint32_t ti = 0;
#pragma unroll
                for (uint32_t step = 128u; step >= 1u; step >>= 1)
                    if (s_off[ti + step] <= lo) ti += step;

                u32x4 v = {0u, 0u, 0u, 0u};
                uint32_t pos = lo;
                while (pos < hi) {
                    const uint32_t t_off = s_off[ti], t_end = s_off[ti + 1];
                    const uint32_t seg_end = t_end < hi ? t_end : hi;
                    if (seg_end > pos) {
                        const uint64_t dd = s_desc[ti];
                        const uint32_t space = uint32_t(dd >> 62);
                        u32x4 ld = dots;
                        if (space != SPACE_FILL) {
                            // byte j of this load is result byte rel + j
                            const uint8_t* p = (space == SPACE_PROTEOME ? base0 : base1)
                                             + (dd & ((1ull << 40) - 1)) + (int64_t(rel) - int64_t(t_off));
                            ld = reinterpret_cast<const unaligned16*>(p)->v;
                        }
                        const uint32_t ja = uint32_t(int32_t(pos) - rel), jb = uint32_t(int32_t(seg_end) - rel);
                        if (ja == 0u && jb == 16u) v = ld;
                        else v = merge_bytes(v, ld, ja, jb);
                        pos = seg_end;
                    }
                    ++ti;
                }
                uint8_t* o = out0 + (uint64_t(b) << 4);
                if (rel >= 0 && uint32_t(rel) + 16u <= total) {
                    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(o));
                    else *reinterpret_cast<u32x4*>(o) = v;
                } else {
                    // ragged first/last block of the chunk: neighbours own the other bytes
                    const uint32_t ja = uint32_t(int32_t(lo) - rel), jb = uint32_t(int32_t(hi) - rel);
#pragma unroll
                    for (uint32_t j = 0; j < 16u; ++j)
                        if (j >= ja && j < jb) o[j] = uint8_t(v[j >> 2] >> (8u * (j & 3u)));
                }
            }
        }
        __syncthreads();   // LDS is reused by the next chunk
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
// digest_kernel: digest[h] = sum_i (byte_i + 1) * splitmix64(i), i relative to
// the haplotype start.  The definition is part of the C ABI contract (see
// include/vcf2prot_hip.h: v2p_batch_digests) so any checker can recompute it.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(256) void digest_kernel(DigestArgs a)
{
    const uint64_t total = a.hap_begin[a.n_haps];
    const uint64_t nblk = (total + 15) >> 4;
    for (uint64_t b = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; ; b += uint64_t(gridDim.x) * blockDim.x) {
        const bool active = b < nblk;
        if (!__any(active)) break;
        uint64_t sum = 0, h = 0;
        bool single = false;
        if (active) {
            const uint64_t p = b << 4;
            // haplotype of byte p: last h with hap_begin[h] <= p (empty haplotypes share an offset)
            uint64_t lo = 0, hi = a.n_haps;
            while (hi - lo > 1) { const uint64_t mid = (lo + hi) >> 1; if (a.hap_begin[mid] <= p) lo = mid; else hi = mid; }
            h = lo;
            const uint64_t pe = p + 16 < total ? p + 16 : total;
            single = a.hap_begin[h + 1] >= pe;
            if (single) {
                const uint64_t hb = a.hap_begin[h];
                for (uint64_t q = p; q < pe; ++q) sum += (uint64_t(a.out[q]) + 1ull) * mix64(q - hb);
            } else {
                for (uint64_t q = p; q < pe; ++q) {
                    while (a.hap_begin[h + 1] <= q) ++h;
                    atomicAdd(reinterpret_cast<unsigned long long*>(&a.digest[h]),
                              (unsigned long long)((uint64_t(a.out[q]) + 1ull) * mix64(q - a.hap_begin[h])));
                }
            }
        }
        // one atomic per wave when the whole wave sits in one haplotype
        const uint64_t h0 = __shfl(h, 0);
        const bool uniform = __all(!active || (single && h == h0)) && __shfl(active && single, 0);
        if (uniform) {
            uint64_t s = active ? sum : 0;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
            if ((threadIdx.x & 63u) == 0) atomicAdd(reinterpret_cast<unsigned long long*>(&a.digest[h0]), (unsigned long long)s);
        } else if (active && single) {
            atomicAdd(reinterpret_cast<unsigned long long*>(&a.digest[h]), (unsigned long long)sum);
        }
    }
}

// ---------------------------------------------------------------------------
// fill_kernel: 16-byte streaming stores; used to measure the write ceiling the
// stitch kernel is compared against (profiles/).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fill_kernel(uint8_t* out, uint64_t n16, uint32_t word, int nt)
{
    const u32x4 v = {word, word, word, word};
    u32x4* o = reinterpret_cast<u32x4*>(out);
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n16; i += uint64_t(gridDim.x) * blockDim.x) {
        if (nt) __builtin_nontemporal_store(v, o + i); else o[i] = v;
    }
}

// ---- launchers (host) -------------------------------------------------------
static inline uint32_t grid_for(uint64_t work_items, uint32_t cap)
{
    return uint32_t(work_items < cap ? (work_items ? work_items : 1) : cap);
}

hipError_t launch_stitch(const StitchArgs& a, hipStream_t stream, int nontemporal, uint32_t max_blocks)
{
    if (a.n_chunks == 0) return hipSuccess;
    const uint32_t grid = grid_for(a.n_chunks, max_blocks ? max_blocks : 0x7FFFFFFFu);
    if (nontemporal) hipLaunchKernelGGL(stitch_kernel<true>, dim3(grid), dim3(256), 0, stream, a);
    else             hipLaunchKernelGGL(stitch_kernel<false>, dim3(grid), dim3(256), 0, stream, a);
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
    const uint64_t nblk = (out_bytes + 15) / 16;
    hipLaunchKernelGGL(digest_kernel, dim3(grid_for((nblk + 255) / 256, 256u * 32u)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_fill(uint8_t* out, uint64_t bytes, uint32_t word, int nontemporal, hipStream_t stream)
{
    const uint64_t n16 = bytes / 16;
    if (n16 == 0) return hipSuccess;
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for((n16 + 255) / 256, 256u * 8u)), dim3(256), 0, stream, out, n16, word, nontemporal);
    return hipGetLastError();
}

}  // namespace v2p
