// build_kernels.hip -- the device image of a batch, built ON the device.
//
// Input: per-transcript GIRs as TranscriptInstruction::get_g_rep returns them (transcript_instructions.rs:335-427), un-rebased,
// concatenated over the transcripts of every haplotype in result order (v2p_txstream, include/vcf2prot_hip.h).
// What the reference does on the host in HaplotypeInstruction::get_g_rep (haplotype_instruction.rs:94-133) -- three running sums
// (ref_counter, alt_counter, res_counter), update_task (:140-158) -- and what sir_pack.hpp's ImageBuilder does after it become:
//   scan        tx_res_base   = exclusive prefix sum of the transcripts' result lengths (res_counter; haplotypes lie back to back,
//                               so hap_out_begin[h] = tx_res_base[first transcript of h]); ref_counter disappears (reference tasks
//                               read the resident proteome at tx_proteome_off), alt_counter is tx_alt_begin (given)
//   count       one lane per transcript walks its tasks, checks what update_task / Task::execute would panic on, and counts the
//               descriptors it will emit: a '.' fill for every gap, the task itself, each cut wherever it crosses a multiple
//               of the grid (chunk k = result bytes [k*W, (k+1)*W): membership is a pure function of result offsets)
//   scan        desc_base = exclusive prefix sum of those counts
//   emit        the same walk writes the descriptors; the piece that starts exactly on a grid line records its index as the
//               first descriptor of that chunk
//   chunks      one lane per chunk: descriptor count, routing flag, proteome slice of its first reference read
//   xcd order   stable counting sort of the chunk table into the launch order of sir_pack.hpp's order_chunks_for_xcds
// Integer/index work only; every kernel streams its arrays once (HBM-bound, tiny next to the stitch kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "build_kernels.h"

namespace v2p {

__device__ __forceinline__ void breport(unsigned long long* status, uint64_t index, uint32_t reason)
{
    atomicMin(status, (unsigned long long)((index << 8) | reason));
}

// ---- exclusive scan u32 -> u64, three passes over 1024-element tiles -------------------------------------------------------
constexpr uint32_t SCAN_TILE = 1024;

__global__ __launch_bounds__(256) void scan_tile_sums(const uint32_t* __restrict__ in, uint64_t n, uint64_t* __restrict__ tile_sum)
{
    __shared__ uint64_t s[4];
    const uint64_t base = uint64_t(blockIdx.x) * SCAN_TILE;
    uint64_t v = 0;
    for (uint32_t k = 0; k < 4; ++k) { const uint64_t i = base + threadIdx.x * 4u + k; if (i < n) v += in[i]; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63u) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// one workgroup: tile_sum[i] <- exclusive prefix; tile_sum[n_tiles] <- total
__global__ __launch_bounds__(1024) void scan_tiles(uint64_t* __restrict__ tile_sum, uint64_t n_tiles)
{
    __shared__ uint64_t s_w[16];
    __shared__ uint64_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (uint64_t b = 0; b < n_tiles; b += 1024) {
        const uint64_t i = b + threadIdx.x;
        const uint64_t x = i < n_tiles ? tile_sum[i] : 0;
        uint64_t v = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint64_t y = __shfl_up(v, o); if ((threadIdx.x & 63u) >= uint32_t(o)) v += y; }
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

__global__ __launch_bounds__(256) void scan_apply(const uint32_t* __restrict__ in, uint64_t n, const uint64_t* __restrict__ tile_sum,
                                                  uint64_t* out, uint64_t first, const uint64_t* first_dev)
{
    __shared__ uint64_t s[4];
    if (first_dev) first += *first_dev;            // (a table scanned slice by slice without the host in between: the slice before left its total there)
    const uint64_t base = uint64_t(blockIdx.x) * SCAN_TILE;
    uint32_t x[4];
    uint64_t v = 0;
    for (uint32_t k = 0; k < 4; ++k) { const uint64_t i = base + threadIdx.x * 4u + k; x[k] = i < n ? in[i] : 0u; v += x[k]; }
    uint64_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint64_t y = __shfl_up(incl, o); if ((threadIdx.x & 63u) >= uint32_t(o)) incl += y; }
    if ((threadIdx.x & 63u) == 63u) s[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint64_t before = first + tile_sum[blockIdx.x];
    for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += s[w];
    uint64_t run = before + incl - v;
    for (uint32_t k = 0; k < 4; ++k) { const uint64_t i = base + threadIdx.x * 4u + k; if (i < n) out[i] = run; run += x[k]; }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) out[n] = first + tile_sum[gridDim.x];      // the total, one past the end
}

// out[i] = first + in[0] + .. + in[i - 1] for i = 0 .. n (`first`: the running total of the slices before this one, when a table is
// scanned slice by slice -- v2p_batch_build_and_execute)
hipError_t launch_scan_u32_from(const uint32_t* in, uint64_t n, uint64_t* out, uint64_t* tile_scratch, uint64_t first, hipStream_t stream)
{
    const uint64_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (n == 0) return hipMemcpyAsync(out, &first, 8, hipMemcpyHostToDevice, stream);
    hipLaunchKernelGGL(scan_tile_sums, dim3(uint32_t(n_tiles)), dim3(256), 0, stream, in, n, tile_scratch);
    hipLaunchKernelGGL(scan_tiles, dim3(1), dim3(1024), 0, stream, tile_scratch, n_tiles);
    hipLaunchKernelGGL(scan_apply, dim3(uint32_t(n_tiles)), dim3(256), 0, stream, in, n, tile_scratch, out, first, static_cast<const uint64_t*>(nullptr));
    return hipGetLastError();
}
// ... continuing from the total the slice before left in out[0] (on the device: nobody reads it back); n >= 1
hipError_t launch_scan_u32_chained(const uint32_t* in, uint64_t n, uint64_t* out, uint64_t* tile_scratch, hipStream_t stream)
{
    const uint64_t n_tiles = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (n == 0) return hipSuccess;                                   // (out[0] already holds the total)
    hipLaunchKernelGGL(scan_tile_sums, dim3(uint32_t(n_tiles)), dim3(256), 0, stream, in, n, tile_scratch);
    hipLaunchKernelGGL(scan_tiles, dim3(1), dim3(1024), 0, stream, tile_scratch, n_tiles);
    hipLaunchKernelGGL(scan_apply, dim3(uint32_t(n_tiles)), dim3(256), 0, stream, in, n, tile_scratch, out, 0ull, static_cast<const uint64_t*>(out));
    return hipGetLastError();
}
hipError_t launch_scan_u32(const uint32_t* in, uint64_t n, uint64_t* out, uint64_t* tile_scratch, hipStream_t stream)
{
    if (n == 0) return hipMemsetAsync(out, 0, 8, stream);
    return launch_scan_u32_from(in, n, out, tile_scratch, 0, stream);
}

#ifdef V2P_BENCH_VARIANTS   // the GRID builders of rounds 2-3 (v2p_batch_build_on_device kernel 1 .. 5): libv2p_bench.so only -- no routing rule picks them
// ---- the walk over one transcript's tasks: EMIT = false counts descriptors, true writes them -------------------------------
__device__ __forceinline__ uint32_t pieces(uint64_t dst, uint64_t len, uint32_t W) { return uint32_t((dst + len - 1) / W - dst / W) + 1u; }

// A position in the result arena as (grid window, offset inside it).  64-bit division is ~100 instructions on this GPU and the
// walk needs window arithmetic for every piece: the transcript's base is divided once, everything after is 32-bit.
struct WinPos { uint64_t win; uint32_t off; };
__device__ __forceinline__ WinPos win_pos(const WinPos& base, uint64_t rel, uint32_t W)
{
    const uint64_t x = uint64_t(base.off) + rel;
    if (x <= 0xFFFFFFFFull) { const uint32_t x32 = uint32_t(x); return WinPos{base.win + x32 / W, x32 % W}; }
    return WinPos{base.win + x / W, uint32_t(x % W)};                              // (a transcript of more than 4 GiB of result)
}

// A lane's descriptors go to consecutive slots: two at a time as one 16-byte store (scattered stores cost the same whatever their
// width -- C5's emit pass wrote 147 M descriptors one by one in 2.9 ms of its 5.4).
struct PairSink {
    uint64_t hold = 0;
    uint32_t has = 0;
    __device__ __forceinline__ void store(uint64_t* desc, uint64_t k, uint64_t d)
    {
        typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
        uint64_t* p = desc + k;
        if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0u) { hold = d; has = 1u; return; }
        if (has) { u64x2 v; v[0] = hold; v[1] = d; *reinterpret_cast<u64x2*>(p - 1) = v; has = 0u; }
        else *p = d;
    }
    __device__ __forceinline__ void finish(uint64_t* desc, uint64_t k) { if (has) { desc[k - 1] = hold; has = 0u; } }   // k = one past the last slot
};

template <bool EMIT>
__device__ __forceinline__ uint32_t put(const BuildArgs& a, PairSink& sink, uint64_t& k, WinPos at, uint64_t len, unsigned space, uint64_t src)
{
    // one run of result bytes from (space, src) starting at `at`, cut at every multiple of the grid
    const uint32_t W = a.window;
    uint32_t n = 0;
    while (len) {
        const uint32_t room = W - at.off;
        const uint32_t piece = uint32_t(len < room ? len : room);
        if (a.split && at.off == 0u) {                                       // the window's spare slot (chunk_kernel may need it to split a descriptor)
            if (EMIT) { sink.store(a.desc, k, uint64_t(SPACE_FILL) << 62); a.chunk_first[at.win] = k; }
            ++k; ++n;
        }
        if (EMIT) {
            sink.store(a.desc, k, (src & SRC_MASK) | (uint64_t(piece & LEN_MASK) << 40) | (uint64_t(space) << 62));
            if (at.off == 0u && !a.split) a.chunk_first[at.win] = k;
        }
        ++k; ++n;
        if (space == SPACE_IMM) src = piece >= 8 ? 0 : src >> (8 * piece);
        else if (space != SPACE_FILL) src += piece;
        at.off += piece; len -= piece;
        if (at.off == W) { at.off = 0u; ++at.win; }
    }
    return n;
}

// sir_pack.hpp's ImageBuilder::stage() as a per-lane state machine: a reference copy, a 1-byte literal and a reference copy going on
// one residue later fuse into one descriptor when the three lie inside one grid window (long-run and dense routing).  Result
// positions are relative to the transcript's first result byte (`rel`).
struct Staged { uint32_t space; uint64_t src, len, rel; };

// MODE: 0 = per-block image (no fusion), 1 = long-run image (fused substitutions), 2 = dense image (and two in a row) -- template
// parameters, so that an image's walk carries only its own state machine
template <bool EMIT, int MODE>
struct Walker {
    const BuildArgs& a;
    uint64_t k;          // next descriptor index (EMIT) / unused
    WinPos base;         // window position of the transcript's first result byte
    uint32_t cnt = 0;    // descriptors so far
    Staged s0{0, 0, 0, 0}, s1{0, 0, 0, 0};                                          // (named slots: an indexed array lives in scratch memory)
    uint32_t len2 = 0, byte2 = 0;                                                    // st_n >= 3: length of the copy after the first literal; st_n == 4: the second literal
                                                                                     // (staged tasks are contiguous in the result: a gap flushes them, so their positions follow from s0.rel)
    int st_n = 0;
    uint64_t run_src = 0;                                                            // st_n >= 3: where the fused run of s0..s2 starts
    PairSink sink;
    __device__ Walker(const BuildArgs& a_, uint64_t k_, uint64_t base_) : a(a_), k(k_), base{base_ / a_.window, uint32_t(base_ % a_.window)} {}
    __device__ __forceinline__ void out(uint32_t space, uint64_t src, uint64_t len, uint64_t rel) { if (len) cnt += put<EMIT>(a, sink, k, win_pos(base, rel, a.window), len, space, src); }
    __device__ __forceinline__ void flush()
    {
        const int n = st_n;
        st_n = 0;
        if (MODE == 2 && n >= 3) {                                               // a complete substitution that waited for a second one
            fused(run_src, uint32_t(s0.len), uint32_t(s1.src), len2, s0.rel);
            if (n == 4) out(SPACE_IMM, byte2, 1, s0.rel + s0.len + 1u + len2);
            return;
        }
        if (n >= 1) out(s0.space, s0.src, s0.len, s0.rel);
        if (n == 2) out(s1.space, s1.src, s1.len, s1.rel);
    }
    __device__ __forceinline__ void fused(uint64_t src, uint32_t len1, uint32_t byte, uint32_t len2, uint64_t rel)
    {
        const uint64_t total = uint64_t(len1) + 1u + len2;
        const WinPos at = win_pos(base, rel, a.window);
        if (uint64_t(at.off) + total <= a.window) {                              // the three inside one window
            if (a.split && at.off == 0u) {
                if (EMIT) { sink.store(a.desc, k, uint64_t(SPACE_FILL) << 62); a.chunk_first[at.win] = k; }
                ++k; ++cnt;
            }
            if (EMIT) {
                sink.store(a.desc, k, SNV3_MARK | (uint64_t(byte & 0xFFu) << 53) | (uint64_t(len2 & 0xFFFu) << 41) | (uint64_t(len1 & 0xFFFu) << 29) | (src & SNV3_MAX_SRC));
                if (at.off == 0u && !a.split) a.chunk_first[at.win] = k;
            }
            ++k; ++cnt;
        } else {
            out(SPACE_PROTEOME, src, len1, rel);
            out(SPACE_IMM, byte, 1, rel + len1);
            out(SPACE_PROTEOME, src + len1 + 1, len2, rel + len1 + 1);
        }
    }
    // two substitutions in a row (sir_pack.hpp's emit_fused2): five tasks, one descriptor when they lie inside one window
    __device__ __forceinline__ void fused2(uint64_t src, uint32_t len1, uint32_t b1, uint32_t len2, uint32_t b2, uint32_t len3, uint64_t rel)
    {
        const uint64_t total = uint64_t(len1) + 1u + len2 + 1u + len3;
        const WinPos at = win_pos(base, rel, a.window);
        if (uint64_t(at.off) + total <= a.window) {
            if (EMIT) {
                sink.store(a.desc, k, SNV5_MARK | (uint64_t(b2 & 0xFFu) << 52) | (uint64_t(b1 & 0xFFu) << 44) | (uint64_t(len3 & 31u) << 39) | (uint64_t(len2 & 31u) << 34)
                                          | (uint64_t(len1 & 31u) << 29) | (src & SNV3_MAX_SRC));
                if (at.off == 0u) a.chunk_first[at.win] = k;
            }
            ++k; ++cnt;
        } else {
            fused(src, len1, b1, len2, rel);
            out(SPACE_IMM, b2, 1, rel + len1 + 1 + len2);
            out(SPACE_PROTEOME, src + len1 + 1 + len2 + 1, len3, rel + len1 + 1 + len2 + 1);
        }
    }
    __device__ __forceinline__ void stage(uint32_t space, uint64_t src, uint64_t len, uint64_t rel)
    {
        if (MODE == 0) { out(space, src, len, rel); return; }
        if (MODE == 2 && st_n == 4) {
            const uint64_t want = run_src + s0.len + 1 + len2 + 1;
            if (space == SPACE_PROTEOME && len <= SNV5_MAX_LEN && (len == 0 || src == want) && want + len <= SNV3_MAX_SRC) {
                st_n = 0;
                fused2(run_src, uint32_t(s0.len), uint32_t(s1.src), len2, byte2, uint32_t(len), s0.rel);
                return;
            }
            flush();
        }
        if (MODE == 2 && st_n == 3) {
            if (space == SPACE_IMM && len == 1) { byte2 = uint32_t(src); st_n = 4; return; }
            flush();
        }
        if (st_n == 2) {
            const bool fits = s0.len == 0 ? (len > 0 && src >= 1 && src - 1 + 1 + len <= SNV3_MAX_SRC) : (len == 0 || src == s0.src + s0.len + 1);
            if (space == SPACE_PROTEOME && len <= SNV3_MAX_LEN && fits) {
                const uint64_t run = s0.len == 0 ? src - 1 : s0.src;
                if (MODE == 2 && s0.len <= SNV5_MAX_LEN && len <= SNV5_MAX_LEN) { len2 = uint32_t(len); run_src = run; st_n = 3; return; }   // a dense image waits for a second one
                st_n = 0;
                fused(run, uint32_t(s0.len), uint32_t(s1.src), uint32_t(len), s0.rel);
                return;
            }
            flush();
        }
        if (st_n == 1) {
            if (space == SPACE_IMM && len == 1) { s1 = Staged{space, src, len, rel}; st_n = 2; return; }
            flush();
        }
        if (space == SPACE_PROTEOME && len <= SNV3_MAX_LEN && src + len + 1 + SNV3_MAX_LEN <= SNV3_MAX_SRC) { s0 = Staged{space, src, len, rel}; st_n = 1; return; }
        // a dense image: a lone literal may open a fused run (the copy before it is empty: flush() writes nothing for it)
        if ((MODE == 2 || (MODE == 1 && a.wave)) && space == SPACE_IMM && len == 1) { s0 = Staged{SPACE_PROTEOME, 0, 0, rel}; s1 = Staged{space, src, len, rel}; st_n = 2; return; }
        out(space, src, len, rel);
    }
};

// (Walker's methods are force-inlined: out of line, the Walker and the BuildArgs lived in scratch memory -- C5's build 8.6 -> 30 ms)
template <bool EMIT, uint32_t SLAB, int MODE>
__global__ __launch_bounds__(256) void walk_kernel(BuildArgs a)
{
    const uint64_t t = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= a.n_tx) return;
    const uint64_t base = a.tx_res_base[t], i0 = a.tx_task_begin[t], i1 = a.tx_task_begin[t + 1];
    const uint64_t alt0 = a.tx_alt_begin[t], n_alt = a.tx_alt_begin[t + 1] - alt0;
    const uint64_t poff = a.tx_proteome_off[t];
    const uint32_t ref_len = a.tx_ref_len[t], res_len = a.tx_res_len[t];
    Walker<EMIT, MODE> w(a, EMIT ? a.desc_base[t] : 0, base);
    // FASTA emit (personalized_genome.rs:90-113): the transcript's arena range is header, residues, line feed; result positions
    // of its tasks move behind the header
    const uint32_t hl = a.tx_header_len ? a.tx_header_len[t] : 0u;
    const uint64_t hsrc = hl ? a.proteome_len + a.tx_header_off[t] : 0ull;
    if (hl) w.out(SPACE_PROTEOME, hsrc, hl, 0);
    uint64_t cur = 0;
    bool ok = true;
    if (!EMIT && poff + ref_len > a.proteome_len) { breport(a.status, i0, STATUS_SRC_OOB); ok = false; }   // transcript outside the resident proteome
    // A lane reads its transcript's tasks SLAB at a time: sixteen tasks as four unaligned 16-byte loads per array, parked in the lane's own
    // column of an LDS tile and taken from there one by one.  Read one per step, a deep Task vector (C5: 129 tasks per transcript) had
    // every lane of a wave on its own cache line in four arrays at every step, lines that did not survive until the next step: the
    // task arrays came over the HBM interface some thirty times (38 ms for C5's build).  (The arrays sit back to back in one device
    // buffer with more behind them: reading a few entries past a transcript's last task is harmless.)
    // (SLAB = 4 for shallow vectors -- C2 has three tasks per transcript -- keeps eight workgroups per CU; 16 for deep ones)
    __shared__ uint32_t s_sp[SLAB][256], s_ln[SLAB][256], s_sr[SLAB][256];
    __shared__ uint8_t s_cd[SLAB][256];
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    struct __attribute__((packed, aligned(4))) Q4 { u32x4 v; };
    struct __attribute__((packed, aligned(1))) B16 { u32x4 v; };
    const uint32_t lane_col = threadIdx.x;
    for (uint64_t ib = i0; ib < i1 && ok; ib += SLAB) {
        const uint32_t m = uint32_t(i1 - ib < SLAB ? i1 - ib : SLAB);
        {
            const u32x4 cd = reinterpret_cast<const B16*>(a.code + ib)->v;
#pragma unroll
            for (uint32_t q = 0; q < SLAB / 4u; ++q) {
                if (4u * q >= m) break;
                const u32x4 vsp = reinterpret_cast<const Q4*>(a.start_pos + ib + 4u * q)->v;
                const u32x4 vln = reinterpret_cast<const Q4*>(a.length + ib + 4u * q)->v;
                const u32x4 vsr = reinterpret_cast<const Q4*>(a.start_pos_res + ib + 4u * q)->v;
#pragma unroll
                for (uint32_t e = 0; e < 4u; ++e) {
                    s_sp[4u * q + e][lane_col] = vsp[e]; s_ln[4u * q + e][lane_col] = vln[e]; s_sr[4u * q + e][lane_col] = vsr[e];
                    s_cd[4u * q + e][lane_col] = uint8_t(cd[q] >> (8u * e));
                }
            }
        }
        for (uint32_t j = 0; j < m && ok; ++j) {
        const uint64_t i = ib + j;
        const uint32_t code = s_cd[j][lane_col];
        const uint64_t sp = s_sp[j][lane_col], ln = s_ln[j][lane_col], sr = s_sr[j][lane_col];
        if (!EMIT) {
            // haplotype_instruction.rs:154 (stream code), task.rs:43/47 (slices), and the canonical order the image needs
            uint32_t why = 0;
            if (code > 1u) why = STATUS_BAD_CODE;
            else if (sr + ln > res_len) why = STATUS_RES_OOB;
            else if (sp + ln > (code == 0 ? uint64_t(ref_len) : n_alt)) why = STATUS_SRC_OOB;
            else if (sr < cur) why = STATUS_NOT_CONTIGUOUS;                  // result ranges overlap or go backwards
            if (why) { breport(a.status, i, why); ok = false; break; }
        }
        if (sr > cur) { w.flush(); w.out(SPACE_FILL, 0, sr - cur, hl + cur); }                // cells no task covers keep '.'
        if (code == 0) w.stage(SPACE_PROTEOME, poff + sp, ln, hl + sr);
        else if (ln >= 1 && ln <= IMM_MAX_BYTES) {                           // short alt payloads travel inside their descriptor
            // (one unaligned 8-byte load -- the payload arena has 32 readable bytes behind it -- instead of up to five dependent byte loads)
            struct __attribute__((packed, aligned(1))) U64 { uint64_t v; };
            const uint64_t lit = reinterpret_cast<const U64*>(a.alt + alt0 + sp)->v & (~0ull >> (64u - 8u * uint32_t(ln)));
            w.stage(SPACE_IMM, lit, ln, hl + sr);
        } else w.stage(SPACE_PAYLOAD, alt0 + sp, ln, hl + sr);
        cur = sr + ln;
        }
    }
    w.flush();
    if (ok && cur < res_len) w.out(SPACE_FILL, 0, res_len - cur, hl + cur);
    if (ok && hl) w.out(SPACE_PROTEOME, hsrc + hl - 1u, 1, uint64_t(hl) + res_len);              // the record's line feed = the header's own
    if (EMIT) w.sink.finish(a.desc, w.k);
    if (!EMIT) a.tx_desc_count[t] = ok ? w.cnt : 0u;
}

template <bool EMIT>
static void launch_walk(const BuildArgs& a, bool deep, uint32_t tx_blocks, hipStream_t stream)
{
    const int mode = a.dense ? 2 : (a.long_run ? 1 : 0);
#define V2P_WALK(SL, MD) hipLaunchKernelGGL((walk_kernel<EMIT, SL, MD>), dim3(tx_blocks), dim3(256), 0, stream, a)
    if (deep) { if (mode == 2) V2P_WALK(16u, 2); else if (mode == 1) V2P_WALK(16u, 1); else V2P_WALK(16u, 0); }
    else      { if (mode == 2) V2P_WALK(4u, 2); else if (mode == 1) V2P_WALK(4u, 1); else V2P_WALK(4u, 0); }
#undef V2P_WALK
}

// ---- chunk table on the grid ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hap_begin_kernel(BuildArgs a, uint64_t out_bytes)
{
    const uint64_t k = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (k <= a.n_haps) a.hap_out_begin[k] = k < a.n_haps ? a.tx_res_base[a.hap_tx_begin[k]] : out_bytes;   // res_counter at the haplotype's first transcript
}


// ---- a wave window that may split once (BuildArgs::split) ------------------------------------------------------------------
// The window's slots are [spare][d0 .. d(m-1)].  m <= 64: one chunk.  65 .. 127: two chunks, cut on the 1 KiB row nearest the middle
// that leaves both with <= 64 descriptors; the descriptor lying across the cut becomes two (its first part moves, with everything
// before it, one slot down into the spare; its second part takes its old slot) -- each kind splits into valid descriptors:
// copies and fills at any byte, an immediate by shifting its literal, a fused substitution into a copy and a fused substitution
// (or the reverse) on either side of its literal.
__device__ __forceinline__ uint32_t wave_desc_len(uint64_t d)
{
    return (d & SNV3_MARK) == SNV3_MARK ? uint32_t((d >> 29) & 0xFFFu) + 1u + uint32_t((d >> 41) & 0xFFFu) : uint32_t(d >> 40) & LEN_MASK;
}
__device__ __forceinline__ void split_desc(uint64_t d, uint32_t x, uint64_t& lo, uint64_t& hi)      // 0 < x < length
{
    if ((d & SNV3_MARK) == SNV3_MARK) {
        const uint64_t src = d & SNV3_MAX_SRC;
        const uint32_t len1 = uint32_t((d >> 29) & 0xFFFu), len2 = uint32_t((d >> 41) & 0xFFFu);
        const uint64_t byte = (d >> 53) & 0xFFull;
        if (x <= len1) {                                             // inside (or right behind) the first copy
            lo = (src & SRC_MASK) | (uint64_t(x) << 40);
            hi = SNV3_MARK | (byte << 53) | (uint64_t(len2) << 41) | (uint64_t(len1 - x) << 29) | ((src + x) & SNV3_MAX_SRC);
        } else {                                                     // behind the literal
            const uint32_t k2 = x - len1 - 1u;
            lo = SNV3_MARK | (byte << 53) | (uint64_t(k2) << 41) | (uint64_t(len1) << 29) | src;
            hi = ((src + x) & SRC_MASK) | (uint64_t(len2 - k2) << 40);
        }
        return;
    }
    const uint64_t space = d >> 62, src = d & SRC_MASK;
    const uint32_t len = uint32_t(d >> 40) & LEN_MASK;
    if (space == SPACE_IMM) { lo = (src & ((1ull << (8u * x)) - 1ull)) | (uint64_t(x) << 40) | (space << 62); hi = (src >> (8u * x)) | (uint64_t(len - x) << 40) | (space << 62); }
    else if (space == SPACE_FILL) { lo = (uint64_t(x) << 40) | (space << 62); hi = (uint64_t(len - x) << 40) | (space << 62); }
    else { lo = src | (uint64_t(x) << 40) | (space << 62); hi = ((src + x) & SRC_MASK) | (uint64_t(len - x) << 40) | (space << 62); }
}
// proteome slice key of a chunk: the first reference read among its first six descriptors (order_chunks_for_xcds)
__device__ __forceinline__ uint64_t first_ref_key(const BuildArgs& a, uint64_t d, bool mine, uint32_t rel)
{
    const bool snv = (d & SNV3_MARK) == SNV3_MARK;
    const uint64_t src = snv ? (d & SNV3_MAX_SRC) : (d & SRC_MASK);
    const bool cand = mine && rel < 6u && (snv || (d >> 62) == SPACE_PROTEOME) && src < a.proteome_len && wave_desc_len(d) != 0u;
    const unsigned long long m = __ballot(cand);
    if (!m) return 0;
    const int first = __ffsll(static_cast<long long>(m)) - 1;
    return (uint64_t(uint32_t(__shfl(int(uint32_t(src >> 32)), first, 64))) << 32) | uint32_t(__shfl(int(uint32_t(src)), first, 64));
}
__device__ __forceinline__ void put_chunk(const BuildArgs& a, uint64_t slot, uint64_t tb, uint64_t dst, uint32_t n, uint64_t key)
{
    a.chunks_tmp[slot] = Chunk{tb, dst | (uint64_t(n) << 48) | CHUNK_WAVE};
    const uint64_t per = (a.proteome_len + 7) / 8;
    const uint64_t bk = per ? key / per : 0;
    a.bucket[slot] = uint8_t(bk < 8 ? bk : 7);
    a.sub[slot] = xcd_sub_window(key, a.bucket[slot], per);
}
__device__ __forceinline__ void chunk_split_window(const BuildArgs& a, uint64_t k, uint64_t n_windows, uint64_t tb, uint32_t n_slots, uint32_t lane)
{
    const uint32_t m = n_slots ? n_slots - 1u : 0u;                  // real descriptors (slot 0 is the spare)
    if (m > 2u * CHUNK_TASKS_WAVE - 1u) { if (lane == 0) { breport(a.status, tb, STATUS_TOO_MANY); a.chunks_tmp[k] = Chunk{tb, (k * a.window) | CHUNK_WAVE}; a.bucket[k] = 0; a.sub[k] = 0; } return; }   // (the sorts behind this kernel read every entry)
    volatile uint32_t* meta = a.meta;
    if (lane == 0 && !(meta[0] & 4u)) atomicOr(&a.meta[0], 4u);
    const uint64_t d0 = lane < m ? a.desc[tb + 1u + lane] : (uint64_t(SPACE_FILL) << 62);
    const uint64_t d1 = lane + 64u < m ? a.desc[tb + 65u + lane] : (uint64_t(SPACE_FILL) << 62);
    if (m <= CHUNK_TASKS_WAVE) {
        const uint64_t key = first_ref_key(a, d0, lane < m, lane);
        if (lane == 0) put_chunk(a, k, tb + 1u, k * a.window, m, key);
        return;
    }
    // positions inside the window: descriptor i covers [start_i, end_i)
    const uint32_t l0 = wave_desc_len(d0), l1 = wave_desc_len(d1);
    uint32_t e0 = l0, e1 = l1;
#pragma unroll
    for (uint32_t s = 1; s < 64u; s <<= 1) { const uint32_t y0 = __shfl_up(e0, s, 64), y1 = __shfl_up(e1, s, 64); if (lane >= s) { e0 += y0; e1 += y1; } }
    e1 += uint32_t(__shfl(int(e0), 63, 64));
    const uint32_t s0 = e0 - l0, s1 = e1 - l1;
    const bool v0 = lane < m, v1 = lane + 64u < m;
    // the cut: a 1 KiB row R, both sides <= 64 descriptors (the one lying across R counts on both), nearest the window's middle
    uint32_t best = 0, best_dist = 0xFFFFFFFFu;
    for (uint32_t R = 1024u; R < a.window; R += 1024u) {
        const uint32_t nA = uint32_t(__popcll(__ballot(v0 && s0 < R))) + uint32_t(__popcll(__ballot(v1 && s1 < R)));          // start before R
        const uint32_t done = uint32_t(__popcll(__ballot(v0 && e0 <= R))) + uint32_t(__popcll(__ballot(v1 && e1 <= R)));      // end at or before R
        const uint32_t nB = m - done;
        const uint32_t dist = R > a.window / 2u ? R - a.window / 2u : a.window / 2u - R;
        if (nA >= 1u && nB >= 1u && nA <= CHUNK_TASKS_WAVE && nB <= CHUNK_TASKS_WAVE && dist < best_dist) { best = R; best_dist = dist; }
    }
    if (best == 0u) { if (lane == 0) { breport(a.status, tb, STATUS_TOO_MANY); a.chunks_tmp[k] = Chunk{tb, (k * a.window) | CHUNK_WAVE}; a.bucket[k] = 0; a.sub[k] = 0; } return; }
    const uint32_t R = best;
    const uint32_t nA = uint32_t(__popcll(__ballot(v0 && s0 < R))) + uint32_t(__popcll(__ballot(v1 && s1 < R)));
    const bool x0 = v0 && s0 < R && e0 > R, x1 = v1 && s1 < R && e1 > R;         // the descriptor lying across R (at most one), index nA - 1
    const bool across = (__ballot(x0) | __ballot(x1)) != 0ull;
    uint64_t tbA, tbB;
    uint32_t cntB;
    if (across) {
        // [d0 .. d(j-1), lo(dj)] move one slot down (slots tb .. tb + j), hi(dj) takes dj's slot, the rest stays
        uint64_t lo0 = d0, hi0 = 0, lo1 = d1, hi1 = 0;
        if (x0) split_desc(d0, R - s0, lo0, hi0);
        if (x1) split_desc(d1, R - s1, lo1, hi1);
        if (v0 && s0 < R) a.desc[tb + lane] = lo0;
        if (v1 && s1 < R) a.desc[tb + 64u + lane] = lo1;
        if (x0) a.desc[tb + 1u + lane] = hi0;                        // (after the moves of the same lane; other lanes write other slots)
        if (x1) a.desc[tb + 65u + lane] = hi1;
        tbA = tb; tbB = tb + nA; cntB = m - nA + 1u;
    } else {
        tbA = tb + 1u; tbB = tb + 1u + nA; cntB = m - nA;
    }
    // keys: chunk A from its first descriptors (unchanged by the split unless the very first one lies across: then its first part,
    // same source), chunk B from the descriptors from index nA - (across ? 1 : 0) on
    const uint64_t keyA = first_ref_key(a, d0, v0, lane);
    const uint32_t jB = across ? nA - 1u : nA;                       // index of B's first descriptor (its source: dj's, advanced -- same slice but for a pathological few)
    const uint32_t r0 = lane >= jB ? lane - jB : 0xFFFFu, r1 = lane + 64u >= jB ? lane + 64u - jB : 0xFFFFu;
    const uint64_t kb0 = first_ref_key(a, d0, v0 && lane >= jB, r0), kb1 = first_ref_key(a, d1, v1 && lane + 64u >= jB, r1);
    const uint64_t keyB = (jB < 64u && kb0) ? kb0 : (kb1 ? kb1 : kb0);
    if (lane == 0) {
        put_chunk(a, k, tbA, k * a.window, nA, keyA);
        const uint32_t o = atomicAdd(&a.meta[4], 1u);
        put_chunk(a, n_windows + o, tbB, k * a.window + R, cntB, keyB);
    }
}

// one WAVE per window: its descriptors read 64 at a time (a lane per window walked them one by one: 0.62 ms for C2's 559 k windows)
__global__ __launch_bounds__(256) void chunk_kernel(BuildArgs a, uint64_t n_windows, uint64_t n_desc, uint64_t out_bytes)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t k = uint64_t(blockIdx.x) * 4u + (threadIdx.x >> 6);
    if (k >= n_windows) return;
    const uint64_t tb = a.chunk_first[k];
    const uint64_t tb_next = k + 1 < n_windows ? a.chunk_first[k + 1] : n_desc;
    const uint64_t n = tb_next - tb;
    if (a.split) { chunk_split_window(a, k, n_windows, tb, uint32_t(n < 4096u ? n : 4096u), lane); return; }
    if (n > CHUNK_TASKS_DEEP) { if (lane == 0) { breport(a.status, tb, STATUS_TOO_MANY); a.chunks_tmp[k] = Chunk{tb, k * a.window}; a.bucket[k] = 0; a.sub[k] = 0; } return; }   // too many descriptors in one window: pick a smaller grid
    // tasks of the window (a fused substitution is up to three) and the proteome slice of its first reference read
    // (order_chunks_for_xcds looks at the first six descriptors)
    uint32_t tasks = 0;
    uint64_t key = 0;
    for (uint32_t q0 = 0; q0 < n; q0 += 64u) {
        const uint32_t q = q0 + lane;
        const uint64_t d = q < n ? a.desc[tb + q] : 0ull;
        const bool snv3 = (d & SNV3_MARK) == SNV3_MARK, snv5 = (d >> 60) == 0xDull, snv = snv3 || snv5;
        if (q < n) tasks += snv3 ? (((d >> 29) & 0xFFFu) ? 1u : 0u) + 1u + (((d >> 41) & 0xFFFu) ? 1u : 0u)
                          : (snv5 ? (((d >> 29) & 31u) ? 1u : 0u) + (((d >> 34) & 31u) ? 1u : 0u) + (((d >> 39) & 31u) ? 1u : 0u) + 2u : 1u);
        if (q0 == 0u) {
            const uint64_t src = snv ? (d & SNV3_MAX_SRC) : (d & SRC_MASK);
            const bool cand = q < n && q < 6u && (snv || (d >> 62) == SPACE_PROTEOME) && src < a.proteome_len;
            const unsigned long long m = __ballot(cand);
            if (m) { const int first = __ffsll(static_cast<long long>(m)) - 1; key = (uint64_t(uint32_t(__shfl(int(uint32_t(src >> 32)), first, 64))) << 32) | uint32_t(__shfl(int(uint32_t(src)), first, 64)); }
        }
    }
#pragma unroll
    for (uint32_t dlt = 32u; dlt; dlt >>= 1) tasks += uint32_t(__shfl_xor(int(tasks), int(dlt), 64));
    if (lane != 0) return;
    if (a.wave ? n > CHUNK_TASKS_WAVE : (a.long_run && tasks > 2u * CHUNK_TASKS)) { breport(a.status, tb, STATUS_TOO_MANY); a.chunks_tmp[k] = Chunk{tb, k * a.window}; a.bucket[k] = 0; a.sub[k] = 0; return; }
    uint64_t flags = 0;
    if (a.wave) flags = CHUNK_WAVE;
    else if (a.long_run) flags = CHUNK_LONG | (tasks > CHUNK_TASKS ? CHUNK_LONG2 : 0ull);
    else if (a.dense) flags = CHUNK_DENSE;
    a.chunks_tmp[k] = Chunk{tb, (k * a.window) | (n << 48) | flags};
    const uint64_t per = (a.proteome_len + 7) / 8;
    const uint64_t bk = per ? key / per : 0;
    a.bucket[k] = uint8_t(bk < 8 ? bk : 7);
    a.sub[k] = xcd_sub_window(key, a.bucket[k], per);
    // what the launcher needs to know about the table (one lane per window gets here: the atomics are only issued while they would
    // still change something -- half a million same-address atomics took 6 ms)
    volatile uint32_t* meta = a.meta;
    if (a.wave) { if (!(meta[0] & 4u)) atomicOr(&a.meta[0], 4u); }
    else if (a.long_run) { if (!(meta[0] & 1u)) atomicOr(&a.meta[0], 1u); if (tasks > CHUNK_TASKS && !(meta[1] & 1u)) atomicOr(&a.meta[1], 1u); }
    else if (a.dense) { if (!(meta[0] & 2u)) atomicOr(&a.meta[0], 2u); }
    else { if (!(meta[2] & 1u)) atomicOr(&a.meta[2], 1u); if (meta[3] < uint32_t(n)) atomicMax(&a.meta[3], uint32_t(n)); }
}

// ---- inside a slice, window-major: stable counting sort by window (sir_pack.hpp: order_chunks_for_xcds) ----------------------
__global__ __launch_bounds__(256) void sub_hist_kernel(const uint8_t* __restrict__ sub, uint64_t n, uint32_t* __restrict__ hist, uint64_t n_blocks)
{
    static_assert(XCD_SUB == 256, "one counter per thread");
    __shared__ uint32_t s[XCD_SUB];
    s[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t k = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (k < n) atomicAdd(&s[sub[k]], 1u);
    __syncthreads();
    hist[uint64_t(threadIdx.x) * n_blocks + blockIdx.x] = s[threadIdx.x];        // window-major: one scan gives every (window, block) its start
}

__global__ __launch_bounds__(256) void sub_scatter_kernel(const Chunk* __restrict__ in, const uint8_t* __restrict__ bucket, const uint8_t* __restrict__ sub,
                                                          uint64_t n, const uint64_t* __restrict__ start, uint64_t n_blocks,
                                                          Chunk* __restrict__ out, uint8_t* __restrict__ out_bucket)
{
    __shared__ uint8_t s_sub[256];
    const uint64_t k = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    const bool live = k < n;
    const uint8_t mine = live ? sub[k] : 0;
    s_sub[threadIdx.x] = mine;
    __syncthreads();
    if (!live) return;
    uint32_t rank = 0;                                                           // chunks of the same window earlier in this block
    for (uint32_t t = 0; t < threadIdx.x; ++t) rank += s_sub[t] == mine ? 1u : 0u;
    const uint64_t pos = start[uint64_t(mine) * n_blocks + blockIdx.x] + rank;
    out[pos] = in[k];
    out_bucket[pos] = bucket[k];
}

// ---- XCD-aware order: entry 8j + x = j-th chunk of slice x (stable) ---------------------------------------------------------
__global__ __launch_bounds__(256) void xcd_hist_kernel(const uint8_t* __restrict__ bucket, uint64_t n, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s[8];
    if (threadIdx.x < 8) s[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t k = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (k < n) atomicAdd(&s[bucket[k]], 1u);
    __syncthreads();
    if (threadIdx.x < 8) hist[uint64_t(blockIdx.x) * 8u + threadIdx.x] = s[threadIdx.x];
}

// exclusive prefix of each slice's counts over the blocks, totals behind the last block: one wave per slice, 64 blocks per step
__global__ __launch_bounds__(512) void xcd_scan_kernel(uint32_t* __restrict__ hist, uint64_t n_blocks)
{
    const uint32_t x = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t run = 0;
    for (uint64_t b0 = 0; b0 < n_blocks; b0 += 64u) {
        const uint64_t b = b0 + lane;
        const uint32_t c = b < n_blocks ? hist[b * 8u + x] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t y = __shfl_up(incl, d, 64); if (lane >= d) incl += y; }
        if (b < n_blocks) hist[b * 8u + x] = run + incl - c;
        run += __shfl(incl, 63, 64);
    }
    if (lane == 0) hist[n_blocks * 8u + x] = run;
}

__global__ __launch_bounds__(256) void xcd_scatter_kernel(const Chunk* __restrict__ in, const uint8_t* __restrict__ bucket, uint64_t n,
                                                          const uint32_t* __restrict__ hist, uint64_t n_blocks, Chunk* __restrict__ out)
{
    __shared__ uint32_t s_wave[4][8];
    const uint64_t k = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const bool live = k < n;
    const uint32_t x = live ? bucket[k] : 8u;
    // rank inside the block among the chunks of the same slice: ballot per slice
    uint32_t in_wave = 0;
#pragma unroll
    for (uint32_t q = 0; q < 8; ++q) {
        const unsigned long long m = __ballot(x == q);
        if (x == q) in_wave = uint32_t(__popcll(m & ((1ull << lane) - 1ull)));
        if (lane == 0) s_wave[wid][q] = uint32_t(__popcll(m));
    }
    __syncthreads();
    if (!live) return;
    uint32_t r = hist[uint64_t(blockIdx.x) * 8u + x] + in_wave;
    for (uint32_t w = 0; w < wid; ++w) r += s_wave[w][x];
    // position of (rank r, slice x) in rank-major order: every slice contributes min(count, r) entries before rank r, and the
    // slices below x that still have a rank-r entry come first
    uint64_t pos = 0;
#pragma unroll
    for (uint32_t q = 0; q < 8; ++q) {
        const uint32_t c = hist[n_blocks * 8u + q];
        pos += c < r ? c : r;
        if (q < x && c > r) ++pos;
    }
    out[pos] = in[k];
}

#endif   // V2P_BENCH_VARIANTS: the grid builders and their one-block sorts

// ---- the same two sorts for ALL blocks of the table in one launch each (sir_pack.hpp: the order is applied inside blocks of the
// arena): grid = (thread blocks of the largest block, blocks); block y holds the entries [first(y), first(y + 1)) -- pure arithmetic,
// xcd_order_block_first -- and, since the blocks follow each other in the table, ONE scan over [block][window][thread block]
// counters gives every entry its final place in the window order, and the slice deal only adds the block's first entry.
__global__ __launch_bounds__(256) void sub_hist_seg_kernel(const uint8_t* __restrict__ sub, uint64_t n, uint32_t nb, uint32_t tbmax, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s[XCD_SUB];
    s[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t y = blockIdx.y, x = blockIdx.x;
    const uint64_t k0 = xcd_order_block_first(n, nb, y), k1 = xcd_order_block_first(n, nb, y + 1u);
    const uint64_t k = k0 + uint64_t(x) * 256u + threadIdx.x;
    if (k < k1) atomicAdd(&s[sub[k]], 1u);
    __syncthreads();
    hist[(uint64_t(y) * XCD_SUB + threadIdx.x) * tbmax + x] = s[threadIdx.x];
}

__global__ __launch_bounds__(256) void sub_scatter_seg_kernel(const Chunk* __restrict__ in, const uint8_t* __restrict__ bucket, const uint8_t* __restrict__ sub,
                                                              uint64_t n, uint32_t nb, uint32_t tbmax, const uint64_t* __restrict__ start,
                                                              Chunk* __restrict__ out, uint8_t* __restrict__ out_bucket)
{
    // (rank among the thread block's earlier entries of the same window -- stable: the lanes of a wave that hold the same window find
    // each other with eight ballots, a wave's count per window goes through LDS.  The loop over every earlier thread this replaces
    // was 0.105 ms of C3 whole's build.)
    __shared__ uint32_t s_cnt[4][XCD_SUB];
    const uint32_t y = blockIdx.y, x = blockIdx.x;
    const uint64_t k0 = xcd_order_block_first(n, nb, y), k1 = xcd_order_block_first(n, nb, y + 1u);
    const uint64_t k = k0 + uint64_t(x) * 256u + threadIdx.x;
    const bool live = k < k1;
    const uint32_t mine = live ? sub[k] : 0u;
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
#pragma unroll
    for (uint32_t q = 0; q < 4u; ++q) s_cnt[q][threadIdx.x] = 0u;
    __syncthreads();
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (uint32_t bit = 0; bit < 8u; ++bit) {
        const bool set = ((mine >> bit) & 1u) != 0u;
        const unsigned long long m = __ballot(set);
        peers &= set ? m : ~m;
    }
    const uint32_t in_wave = uint32_t(__popcll(peers & ((1ull << lane) - 1ull)));
    if (live && in_wave == 0u) s_cnt[wid][mine] = uint32_t(__popcll(peers));
    __syncthreads();
    if (!live) return;
    uint32_t rank = in_wave;
    for (uint32_t w = 0; w < wid; ++w) rank += s_cnt[w][mine];
    const uint64_t pos = start[(uint64_t(y) * XCD_SUB + mine) * tbmax + x] + rank;
    out[pos] = in[k];
    out_bucket[pos] = bucket[k];
}

__global__ __launch_bounds__(256) void xcd_hist_seg_kernel(const uint8_t* __restrict__ bucket, uint64_t n, uint32_t nb, uint32_t tbmax, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t s[8];
    if (threadIdx.x < 8) s[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t y = blockIdx.y, x = blockIdx.x;
    const uint64_t k0 = xcd_order_block_first(n, nb, y), k1 = xcd_order_block_first(n, nb, y + 1u);
    const uint64_t k = k0 + uint64_t(x) * 256u + threadIdx.x;
    if (k < k1) atomicAdd(&s[bucket[k]], 1u);
    __syncthreads();
    if (threadIdx.x < 8) hist[(uint64_t(y) * tbmax + x) * 8u + threadIdx.x] = s[threadIdx.x];
}

// per block: exclusive prefix of each slice's counts over the block's thread blocks, totals in tot[y * 8 + slice]
__global__ __launch_bounds__(512) void xcd_scan_seg_kernel(uint32_t* __restrict__ hist, uint32_t tbmax, uint32_t* __restrict__ tot)
{
    const uint32_t y = blockIdx.x, x = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t* h = hist + uint64_t(y) * tbmax * 8u;
    uint32_t run = 0;
    for (uint32_t b0 = 0; b0 < tbmax; b0 += 64u) {
        const uint32_t b = b0 + lane;
        const uint32_t c = b < tbmax ? h[b * 8u + x] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
        if (b < tbmax) h[b * 8u + x] = run + incl - c;
        run += __shfl(incl, 63, 64);
    }
    if (lane == 0) tot[y * 8u + x] = run;
}

__global__ __launch_bounds__(256) void xcd_scatter_seg_kernel(const Chunk* __restrict__ in, const uint8_t* __restrict__ bucket, uint64_t n, uint32_t nb, uint32_t tbmax,
                                                              const uint32_t* __restrict__ hist, const uint32_t* __restrict__ tot, Chunk* __restrict__ out)
{
    __shared__ uint32_t s_wave[4][8];
    const uint32_t y = blockIdx.y, xb = blockIdx.x;
    const uint64_t k0 = xcd_order_block_first(n, nb, y), k1 = xcd_order_block_first(n, nb, y + 1u);
    const uint64_t k = k0 + uint64_t(xb) * 256u + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
    const bool live = k < k1;
    const uint32_t x = live ? bucket[k] : 8u;
    uint32_t in_wave = 0;
#pragma unroll
    for (uint32_t q = 0; q < 8; ++q) {
        const unsigned long long m = __ballot(x == q);
        if (x == q) in_wave = uint32_t(__popcll(m & ((1ull << lane) - 1ull)));
        if (lane == 0) s_wave[wid][q] = uint32_t(__popcll(m));
    }
    __syncthreads();
    if (!live) return;
    uint32_t r = hist[(uint64_t(y) * tbmax + xb) * 8u + x] + in_wave;
    for (uint32_t w = 0; w < wid; ++w) r += s_wave[w][x];
    uint64_t pos = 0;
#pragma unroll
    for (uint32_t q = 0; q < 8; ++q) {
        const uint32_t c = tot[y * 8u + q];
        pos += c < r ? c : r;
        if (q < x && c > r) ++pos;
    }
    out[k0 + pos] = in[k];
}

// both sorts over the first n entries of the table, block by block (nb blocks): scratch as launch_sub_order / launch_xcd_order,
// sized for (n / 256 + nb) thread blocks; tot = 8 * nb u32
hipError_t launch_order_blocks(const Chunk* in, const uint8_t* bucket, const uint8_t* sub, uint64_t n, uint32_t nb, uint32_t* subhist, uint64_t* substart,
                               uint64_t* tiles, Chunk* by_window, uint8_t* bucket2, uint32_t* hist8, uint32_t* tot, Chunk* out, hipStream_t stream)
{
    if (n == 0 || nb == 0) return hipSuccess;
    uint64_t biggest = 0;
    for (uint32_t y = 0; y < nb; ++y) { const uint64_t e = xcd_order_block_first(n, nb, y + 1u) - xcd_order_block_first(n, nb, y); if (e > biggest) biggest = e; }
    const uint32_t tbmax = uint32_t((biggest + 255) / 256);
    const dim3 grid(tbmax, nb);
    hipLaunchKernelGGL(sub_hist_seg_kernel, grid, dim3(256), 0, stream, sub, n, nb, tbmax, subhist);
    hipError_t e = launch_scan_u32(subhist, uint64_t(nb) * XCD_SUB * tbmax, substart, tiles, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sub_scatter_seg_kernel, grid, dim3(256), 0, stream, in, bucket, sub, n, nb, tbmax, substart, by_window, bucket2);
    hipLaunchKernelGGL(xcd_hist_seg_kernel, grid, dim3(256), 0, stream, bucket2, n, nb, tbmax, hist8);
    hipLaunchKernelGGL(xcd_scan_seg_kernel, dim3(nb), dim3(512), 0, stream, hist8, tbmax, tot);
    hipLaunchKernelGGL(xcd_scatter_seg_kernel, grid, dim3(256), 0, stream, by_window, bucket2, n, nb, tbmax, hist8, tot, out);
    return hipGetLastError();
}
uint64_t order_blocks_thread_blocks(uint64_t n, uint32_t nb) { return (n + 255) / 256 + 2ull * nb + 2; }   // >= nb * tbmax for any equal-share split

#ifdef V2P_BENCH_VARIANTS
hipError_t launch_build(const BuildArgs& a, uint64_t n_windows, uint64_t n_desc, uint64_t out_bytes, int phase, hipStream_t stream)
{
    const uint32_t tx_blocks = uint32_t((a.n_tx + 255) / 256);
    if (phase == 0) {                       // count
        const bool deep = a.n_tx && a.n_tasks / a.n_tx > 6u;
        if (a.n_tx) launch_walk<false>(a, deep, tx_blocks, stream);
    } else if (phase == 1) {                // emit + chunk table
        const bool deep = a.n_tx && a.n_tasks / a.n_tx > 6u;
        if (a.n_tx) launch_walk<true>(a, deep, tx_blocks, stream);
        hipLaunchKernelGGL(hap_begin_kernel, dim3(uint32_t((a.n_haps + 1 + 255) / 256)), dim3(256), 0, stream, a, out_bytes);
        if (n_windows) hipLaunchKernelGGL(chunk_kernel, dim3(uint32_t((n_windows + 3) / 4)), dim3(256), 0, stream, a, n_windows, n_desc, out_bytes);
    }
    return hipGetLastError();
}

#endif

uint64_t scan_tiles_for(uint64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE + 1; }

#ifdef V2P_BENCH_VARIANTS

hipError_t launch_sub_order(const Chunk* in, const uint8_t* bucket, const uint8_t* sub, uint64_t n, uint32_t* hist, uint64_t* start, uint64_t* tiles,
                            Chunk* out, uint8_t* out_bucket, hipStream_t stream)
{
    const uint64_t n_blocks = (n + 255) / 256;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(sub_hist_kernel, dim3(uint32_t(n_blocks)), dim3(256), 0, stream, sub, n, hist, n_blocks);
    hipError_t e = launch_scan_u32(hist, uint64_t(XCD_SUB) * n_blocks, start, tiles, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sub_scatter_kernel, dim3(uint32_t(n_blocks)), dim3(256), 0, stream, in, bucket, sub, n, start, n_blocks, out, out_bucket);
    return hipGetLastError();
}

hipError_t launch_xcd_order(const Chunk* in, const uint8_t* bucket, uint64_t n, uint32_t* hist, Chunk* out, hipStream_t stream)
{
    const uint64_t n_blocks = (n + 255) / 256;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(xcd_hist_kernel, dim3(uint32_t(n_blocks)), dim3(256), 0, stream, bucket, n, hist);
    hipLaunchKernelGGL(xcd_scan_kernel, dim3(1), dim3(512), 0, stream, hist, n_blocks);
    hipLaunchKernelGGL(xcd_scatter_kernel, dim3(uint32_t(n_blocks)), dim3(256), 0, stream, in, bucket, n, hist, n_blocks, out);
    return hipGetLastError();
}

#endif   // V2P_BENCH_VARIANTS: launch_sub_order / launch_xcd_order (one block)

__global__ void code_object_loader_c() {}
hipError_t preload_build_kernels(hipStream_t stream)
{
    hipLaunchKernelGGL(code_object_loader_c, dim3(1), dim3(64), 0, stream);
    return hipGetLastError();
}

}  // namespace v2p
