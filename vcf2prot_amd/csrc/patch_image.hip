// patch_image.hip -- PATCH images (patch_image.h): the builder (ONE kernel, one workgroup per 8 KiB window of the result arena) and the
// executor (stitch_patch_kernel, one workgroup per window) for batches of DEEP Task vectors.  gfx950, wave64; no MFMA (byte / index work).
//
//   positions   res_counter of haplotype_instruction.rs:90,132 as a scan over the transcripts' arena lengths (build_kernels.hip: launch_scan_u32)
//   build       workgroup = window [c * 8 KiB, (c + 1) * 8 KiB) of the arena: the transcripts whose records overlap it (a table: the first transcript at or
//               behind every window's first byte), one WAVE per transcript, lane = Task (task.rs:2-9) in windows of 64 with two context lanes either side.
//               A lane classifies its Task from its own fields and its neighbours' (DPP): a one-residue alt Task between two reference
//               copies that go on one residue later is a PATCH; a reference copy that follows such a patch CONTINUES the segment of the copy
//               before it; every other Task starts a segment (reference / alt payload / literal); cells no Task covers are '.' segments
//               (haplotype_instruction.rs:78); FASTA headers and line feeds are segments reading the resident header table
//               (personalized_genome.rs:90-113).  A segment's end is the end of its chain of continuations: ballot masks, one
//               count-trailing-zeros, two ds_bpermute.  Everything is clipped to the window and appended to the window's own slots
//               (one LDS atomic per wave and window of Tasks).  update_task's and Task::execute's panics
//               (haplotype_instruction.rs:140-158, task.rs:43,47) are reported by Task index; a transcript is processed by every
//               window it overlaps (7 % of redundant reads for 800-residue transcripts).
//   execute     stitch_patch_kernel: segments -> LDS, a block map (which segment covers the first byte of each 16-byte block), one
//               byte-granular 16-byte gather per block into a 8 KiB LDS image, the segments' ragged first pieces OR-ed in, the
//               patches written as bytes, the image out as aligned non-temporal 16-byte stores.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "patch_image.h"
#include "build_kernels.h"
#include "stitch_device.hpp"

namespace v2p {
namespace {

__device__ __forceinline__ uint32_t prev_lane(uint32_t x, uint32_t first) { return uint32_t(__builtin_amdgcn_update_dpp(int(first), int(x), 0x138, 0xf, 0xf, false)); }   // lane i <- lane i - 1 (wave_shr:1)
__device__ __forceinline__ uint32_t next_lane(uint32_t x, uint32_t last) { return uint32_t(__builtin_amdgcn_update_dpp(int(last), int(x), 0x130, 0xf, 0xf, false)); }    // lane i <- lane i + 1 (wave_shl:1)
__device__ __forceinline__ uint32_t lanes_below(uint64_t m) { return __builtin_amdgcn_mbcnt_hi(uint32_t(m >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(m), 0u)); }

__global__ __launch_bounds__(256) void patch_arena_len_kernel(PatchBuildArgs a)
{
    const uint64_t t = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (t >= a.n_tx) return;
    const uint32_t hl = a.tx_header_len ? a.tx_header_len[t] : 0u;
    a.tx_arena_len[t] = a.tx_res_len[t] + (hl ? hl + 1u : 0u);          // (< 2^32: checked on the host for FASTA streams)
}

__global__ __launch_bounds__(256) void patch_hap_begin_kernel(PatchBuildArgs a)
{
    const uint64_t h = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (h > a.n_haps) return;
    a.hap_out_begin[h] = a.tx_res_base[h < a.n_haps ? a.hap_tx_begin[h] : a.n_tx];
}

// chunk_tx[c] = first transcript t in [0, n_tx] whose record begins at or behind the chunk's first byte (tx_res_base is non-decreasing):
// transcript t writes the entries of the chunks that begin in (base[t - 1], base[t]] -- every chunk gets exactly one writer.  (The first
// form of the builder had every WORKGROUP binary-search the 2 M offsets twice: 42 dependent loads, 4 ms of BASELINE config 5's 4.9.)
__global__ __launch_bounds__(256) void patch_chunk_tx_kernel(PatchBuildArgs a)
{
    const uint64_t t = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    if (t > a.n_tx) return;
    const uint64_t b1 = a.tx_res_base[t];
    // chunks c with c * G <= b1 and (t == 0 or c * G > base[t - 1])
    uint64_t c_lo = 0;
    if (t > 0) { const uint64_t b0 = a.tx_res_base[t - 1]; c_lo = b0 / PATCH_G + 1u; }
    uint64_t c_hi = b1 / PATCH_G;                                      // last chunk whose first byte is <= b1
    if (c_hi >= a.n_chunks) c_hi = a.n_chunks ? a.n_chunks - 1u : 0u;
    for (uint64_t c = c_lo; c <= c_hi && c < a.n_chunks; ++c) a.chunk_tx[c] = t;
}

template <bool FASTA>
__global__ __launch_bounds__(256) void patch_build_kernel(PatchBuildArgs a)
{
    __shared__ uint32_t s_nseg, s_npatch, s_over;
    __shared__ unsigned long long s_key;                  // the window's first reference read: min over its proteome segments of start << 34 | source
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t c = blockIdx.x;
    const uint64_t lo = c * PATCH_G, hi = lo + PATCH_G < a.out_bytes ? lo + PATCH_G : a.out_bytes;
    const bool last_chunk = c + 1u == a.n_chunks;
    if (tid == 0) { s_nseg = 0u; s_npatch = 0u; s_over = 0u; s_key = ~0ull; }
    __syncthreads();
    // transcripts of this window: the one lying across its first byte, then every one that begins inside it (an empty record belongs to
    // the window its offset falls in; those at the very end of the arena to the last window)
    uint64_t t_a = a.chunk_tx[c];                                        // (= first_at_or_after(tx_res_base, lo), from patch_chunk_tx_kernel)
    if (t_a > 0 && (t_a > a.n_tx || a.tx_res_base[t_a] > lo)) --t_a;
    const uint64_t t_b = last_chunk ? a.n_tx : a.chunk_tx[c + 1];          // (= first_at_or_after(tx_res_base, hi))
    uint64_t* const seg_out = a.seg + c * PATCH_SEG_CAP;
    uint32_t* const patch_out = a.patch + c * PATCH_PATCH_CAP;
    constexpr int NS = FASTA ? 5 : 3;                     // segments a lane may emit: [header] gap fill, its own, tail fill [line feed]

    // The window's transcripts go through an LDS table, 64 at a time (ONE round trip for all their rows), and every wave requests the
    // Tasks of its NEXT window of 64 -- of the same transcript or of its next one -- before it works on the current one: a wave's life
    // was a chain of dependent round trips in the first form of this kernel (50 us per workgroup, 4.8 ms for BASELINE config 5 whatever
    // its Task count).
    __shared__ uint64_t s_rb[64], s_poff[64], s_alt0[64], s_tb0[64], s_hsrc[64];
    __shared__ uint32_t s_n[64], s_reflen[64], s_reslen[64], s_nalt[64], s_hl[64];
    const uint64_t t_end = t_b < a.n_tx ? t_b : a.n_tx;
    auto uni32 = [](uint32_t x) { return uint32_t(__builtin_amdgcn_readfirstlane(int(x))); };
    auto uni64 = [&](uint64_t x) { return (uint64_t(uni32(uint32_t(x >> 32))) << 32) | uni32(uint32_t(x)); };
    for (uint64_t g0 = t_a; g0 < t_end; g0 += 64u) {
        const uint32_t ng = uint32_t(t_end - g0 < 64u ? t_end - g0 : 64u);
        __syncthreads();
        if (tid < ng) {
            const uint64_t t = g0 + tid;
            const uint32_t hl_ = FASTA ? a.tx_header_len[t] : 0u;
            const uint64_t tb0_ = a.tx_task_begin[t], tb1_ = a.tx_task_begin[t + 1], alt0_ = a.tx_alt_begin[t];
            s_rb[tid] = a.tx_res_base[t]; s_poff[tid] = a.tx_proteome_off[t]; s_alt0[tid] = alt0_; s_tb0[tid] = tb0_;
            s_hsrc[tid] = (FASTA && hl_) ? a.proteome_len + a.tx_header_off[t] : 0ull;
            s_n[tid] = uint32_t(tb1_ - tb0_ < 0xFFFFFFFFull ? tb1_ - tb0_ : 0xFFFFFFFFull);
            s_reflen[tid] = a.tx_ref_len[t]; s_reslen[tid] = a.tx_res_len[t]; s_nalt[tid] = uint32_t(a.tx_alt_begin[t + 1] - alt0_); s_hl[tid] = hl_;
        }
        __syncthreads();
        uint32_t r = wave, w = 0u;
        uint32_t code = 0, sp = 0, ln = 0, sr = 0;                       // the Tasks of the window about to be worked on
        auto fetch = [&](uint32_t rr, uint32_t ww, uint32_t& f_code, uint32_t& f_sp, uint32_t& f_ln, uint32_t& f_sr) {
            f_code = 0; f_sp = 0; f_ln = 0; f_sr = 0;
            if (rr < ng) {
                const uint32_t jj = ww + lane - 2u;
                if (jj < s_n[rr]) { const uint64_t i = s_tb0[rr] + jj; f_code = a.code[i]; f_sp = a.start_pos[i]; f_ln = a.length[i]; f_sr = a.start_pos_res[i]; }
            }
        };
        fetch(r, w, code, sp, ln, sr);
        uint32_t carry_e = 0u;                                          // end of the Task before the window's first lane (lane 0's predecessor)
        while (r < ng) {
            const uint32_t n = uni32(s_n[r]);
            // the window after this one: requested now
            uint32_t nr = r, nw = w + 60u;
            if (n == 0u || nw >= n) { nr = r + 4u; nw = 0u; }
            uint32_t x_code, x_sp, x_ln, x_sr;
            fetch(nr, nw, x_code, x_sp, x_ln, x_sr);
            const uint64_t rb = uni64(s_rb[r]), poff = uni64(s_poff[r]), alt0 = uni64(s_alt0[r]), tb0 = uni64(s_tb0[r]);
            const uint32_t hl = FASTA ? uni32(s_hl[r]) : 0u;
            const uint64_t hsrc = FASTA ? uni64(s_hsrc[r]) : 0ull;
            const uint64_t base = rb + hl;                                   // arena offset of the transcript's first result cell
            const uint32_t ref_len = uni32(s_reflen[r]), res_len = uni32(s_reslen[r]), n_alt = uni32(s_nalt[r]);
            const bool owned = (rb >= lo && rb < hi) || (last_chunk && rb >= hi);      // the window that reports this transcript's panics
            if (w == 0u) {
                carry_e = 0u;
                if (owned && lane == 0u && poff + ref_len > a.proteome_len) atomicMin(a.status, (unsigned long long)((tb0 << 8) | STATUS_SRC_OOB));
            }
            // lane l <-> Task j = w - 2 + l of the transcript: lanes 0, 1 are context (emitted by the window before), 62, 63 look-ahead
            // lane l <-> Task j = w - 2 + l of the transcript: lanes 0, 1 are context (emitted by the window before), 62, 63 look-ahead
            const uint32_t j = w + lane - 2u;
            const bool valid = j < n;                                    // (j wraps below 0 for the context lanes of the first window)
            const bool own = valid && lane >= 2u && lane < 62u;
            // ---- update_task / Task::execute, as rows_parse_kernel checks them ----
            const bool res_oob = valid && (ln > res_len || sr > res_len - ln);
            const uint32_t e = valid && !res_oob ? sr + ln : 0u;
            const uint32_t pe_raw = prev_lane(e, carry_e);
            const uint32_t pe = j == 0u ? 0u : pe_raw;                    // end of the Task before (0 for a transcript's first)
            const uint32_t bound = code == 1u ? n_alt : ref_len;
            const bool src_oob = ln > bound || sp > bound - ln;
            const bool bad = valid && (code > 1u || res_oob || src_oob || (j != 0u && sr < pe));
            if (__ballot(bad && own) != 0ull) {
                if (bad && own && owned) {
                    const uint32_t why = code > 1u ? STATUS_BAD_CODE : (res_oob ? STATUS_RES_OOB : (src_oob ? STATUS_SRC_OOB : STATUS_NOT_CONTIGUOUS));
                    atomicMin(a.status, (unsigned long long)(((tb0 + j) << 8) | why));
                }
            }
            const bool good = valid && !bad;
            const bool isRef = good && code == 0u, isAlt = good && code == 1u;
            // ---- classification (lane = Task, neighbours by DPP) ----
            // An alt Task of ONE residue followed directly by a non-empty reference copy (result contiguous, the copy not at the reference's
            // very first residue) is a PATCH, and that copy's run begins one cell EARLY, under the patch, with the reference byte before
            // its own first one: for a missense that is the substituted residue itself and -- when the copy before the patch ends exactly
            // there -- the run simply CONTINUES that copy's segment; for a deletion's anchor it is a new segment that starts one cell early.
            const bool refc = isRef && ln >= 1u;
            const uint32_t n_refc = next_lane(refc ? 1u : 0u, 0u), n_sp = next_lane(sp, 0u), n_sr = next_lane(sr, 0u);
            const bool A = isAlt && ln == 1u && lane <= 62u && n_refc != 0u && n_sr == sr + 1u && n_sp >= 1u;
            const uint64_t mA = __ballot(A);
            const uint32_t srcend = sp + ln;                             // (reference copies: the residue behind the copy)
            const uint32_t p2_refc = prev_lane(prev_lane(refc ? 1u : 0u, 0u), 0u), p2_e = prev_lane(prev_lane(e, 0u), 0u), p2_srcend = prev_lane(prev_lane(srcend, 0u), 0u);
            const bool absorbs = refc && lane >= 1u && ((mA >> (lane - 1u)) & 1ull);      // the Task before is a patch this copy lies under
            const bool K = absorbs && lane >= 2u && p2_refc != 0u && p2_e + 1u == sr && p2_srcend + 1u == sp;    // ... and the copy two back goes straight on: one segment
            const uint64_t mK = __ballot(K), mX = ((mK >> 1) & mA) | mK;  // X: the patches and copies INSIDE a chain (behind its head)
            const bool X = (mX >> lane) & 1ull;
            const bool isP = (mA >> lane) & 1ull;
            const bool H = refc && !K;                                    // a copy that starts a segment
            const bool starter = own && (H || (lane == 2u && X));
            // the chain a starter heads: lanes l + 1 .. m while they are X, at most to lane 61
            uint32_t m = lane;
            {
                const uint64_t run = lane < 63u ? mX >> (lane + 1u) : 0ull;
                const uint32_t k = uint32_t(__builtin_ctzll(~run));
                m = lane + k < 61u ? lane + k : (lane > 61u ? lane : 61u);
                if (m < lane) m = lane;
            }
            const uint32_t m_e = uint32_t(__shfl(int(e), int(m))), m_sr = uint32_t(__shfl(int(sr), int(m)));
            const bool m_isP = (mA >> m) & 1ull;
            // ---- what the lane emits: up to NS segments (arena coordinates) and one patch ----
            uint64_t ss[NS], se[NS], sx[NS];
            unsigned sk[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) { ss[i] = 0; se[i] = 0; sx[i] = 0; sk[i] = SPACE_FILL; }
            constexpr int IG = FASTA ? 1 : 0, IM = IG + 1, IT = IG + 2;
            if (own) {
                if (FASTA && hl) {
                    if (j == 0u) { ss[0] = rb; se[0] = rb + hl; sx[0] = hsrc; sk[0] = SPACE_PROTEOME; }
                    if (j + 1u == n) { ss[NS - 1] = base + res_len; se[NS - 1] = base + res_len + 1u; sx[NS - 1] = hsrc + hl - 1u; sk[NS - 1] = SPACE_PROTEOME; }
                }
                if (good && sr > pe) { ss[IG] = base + pe; se[IG] = base + sr; }                       // cells no Task covers: '.'
                if (j + 1u == n && good && res_len > e) { ss[IT] = base + e; se[IT] = base + res_len; }
                if (starter) {
                    // where the run begins: a head that lies under the patch before it one cell early -- but not lane 2 of a later window when
                    // that patch's cell was covered by the window before (it was, exactly when lane 2 is INSIDE a chain); a patch heading
                    // a forced run (lane 2 inside a chain) begins at its own cell with the byte before its copy's first one
                    const bool early = H && absorbs;
                    const uint32_t s_res = early ? sr - 1u : sr;
                    const uint64_t s_src = isP ? poff + n_sp - 1u : (early ? poff + sp - 1u : poff + sp);
                    ss[IM] = base + s_res; se[IM] = base + (m_isP ? m_sr + 1u : m_e);
                    sx[IM] = s_src; sk[IM] = SPACE_PROTEOME;
                } else if (isAlt && !isP && ln >= 1u) {
                    ss[IM] = base + sr; se[IM] = base + e;
                    if (ln <= PATCH_IMM_MAX) {
                        struct __attribute__((packed, aligned(1))) U32 { uint32_t v; };
                        const uint32_t lit = reinterpret_cast<const U32*>(a.alt + alt0 + sp)->v;
                        sx[IM] = ln == 4u ? lit : lit & ((1u << (8u * ln)) - 1u); sk[IM] = SPACE_IMM;
                    } else { sx[IM] = alt0 + sp; sk[IM] = SPACE_PAYLOAD; }
                }
            }
            const bool P = isP;
            // clip to the window; count
            uint32_t cnt = 0;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const uint64_t s0 = ss[i] > lo ? ss[i] : lo, e0 = se[i] < hi ? se[i] : hi;
                if (se[i] > ss[i] && e0 > s0) {
                    const uint64_t cut = s0 - ss[i];
                    if (sk[i] == SPACE_IMM) sx[i] >>= 8u * cut; else if (sk[i] != SPACE_FILL) sx[i] += cut;
                    ss[i] = s0; se[i] = e0; ++cnt;
                } else { ss[i] = 0; se[i] = 0; }
            }
            const bool patch_here = own && P && base + sr >= lo && base + sr < hi;
            uint32_t pbyte = 0;
            if (patch_here) pbyte = a.alt[alt0 + sp];
            // slots: one LDS atomic per wave for the segments, one for the patches
            const uint32_t incl = wave_incl_scan(cnt), total = uint32_t(__builtin_amdgcn_readlane(int(incl), 63));
            if (total) {
                uint32_t b0 = 0;
                if (lane == 0u) b0 = atomicAdd(&s_nseg, total);
                b0 = uint32_t(__builtin_amdgcn_readfirstlane(int(b0)));
                uint32_t k = b0 + incl - cnt;
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    if (se[i] > ss[i]) {
                        const uint32_t start = uint32_t(ss[i] - lo), len = uint32_t(se[i] - ss[i]);
                        if (k < PATCH_SEG_CAP && sx[i] <= PATCH_SRC_MAX) seg_out[k] = patch_seg(sx[i], start, len, sk[i]);
                        else s_over = 1u;
                        if (sk[i] == SPACE_PROTEOME && sx[i] < a.proteome_len) atomicMin(&s_key, (unsigned long long)((uint64_t(start) << 34) | sx[i]));
                        ++k;
                    }
                }
            }
            const uint64_t pm = __ballot(patch_here);
            if (pm) {
                uint32_t b0 = 0;
                if (lane == 0u) b0 = atomicAdd(&s_npatch, uint32_t(__popcll(pm)));
                b0 = uint32_t(__builtin_amdgcn_readfirstlane(int(b0)));
                const uint32_t k = b0 + lanes_below(pm);
                if (patch_here) { if (k < PATCH_PATCH_CAP) patch_out[k] = patch_word(uint32_t(base + sr - lo), pbyte); else s_over = 1u; }
            }
            carry_e = uint32_t(__builtin_amdgcn_readlane(int(e), 59));   // (the next window's lane 0 is Task w + 58; its predecessor w + 57 sits in this window's lane 59)
            if (n == 0u && lane == 0u) {
                // a transcript without Tasks: its cells are '.', its record still has its header and line feed (transcript_instructions.rs:338-343)
                uint64_t ss[3] = {rb, base, base + res_len}, se[3] = {rb + hl, base + res_len, base + res_len + (hl ? 1u : 0u)};
                uint64_t sx[3] = {hsrc, 0, hsrc + hl - 1u};
                const unsigned sk[3] = {SPACE_PROTEOME, SPACE_FILL, SPACE_PROTEOME};
                for (int i = 0; i < 3; ++i) {
                    if (!FASTA && i != 1) continue;
                    const uint64_t s0 = ss[i] > lo ? ss[i] : lo, e0 = se[i] < hi ? se[i] : hi;
                    if (se[i] > ss[i] && e0 > s0) {
                        if (sk[i] != SPACE_FILL) sx[i] += s0 - ss[i];
                        const uint32_t k = atomicAdd(&s_nseg, 1u);
                        if (k < PATCH_SEG_CAP && sx[i] <= PATCH_SRC_MAX) seg_out[k] = patch_seg(sx[i], uint32_t(s0 - lo), uint32_t(e0 - s0), sk[i]);
                        else s_over = 1u;
                    }
                }
            }

            code = x_code; sp = x_sp; ln = x_ln; sr = x_sr; r = nr; w = nw;
        }
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t ns = s_nseg, np = s_npatch;
        const bool over = s_over != 0u || ns > PATCH_SEG_CAP || np > PATCH_PATCH_CAP;
        if (over) atomicMin(a.status, (unsigned long long)((c << 8) | STATUS_PATCH_DECLINED));
        a.chunks[c] = Chunk{(c * PATCH_SEG_CAP) | (uint64_t(over ? 0u : np) << TB_IDX_BITS), lo | (uint64_t(over ? 0u : ns) << 48) | CHUNK_PATCH};
        const uint64_t key = s_key == ~0ull ? 0ull : (s_key & PATCH_SRC_MAX);
        const uint64_t per = (a.proteome_len + 7) / 8;
        const uint64_t bk = per ? key / per : 0;
        const uint8_t bucket = uint8_t(bk < 8 ? bk : 7);
        a.bucket[c] = bucket;
        a.sub[c] = xcd_sub_window(key, bucket, per);
    }
}

// segments and patches of the whole image: summed over the chunk records, one atomic pair per 256 chunks (one pair per CHUNK from the
// build kernel itself -- 390 000 atomics on two addresses -- was 4 of that kernel's 4.8 ms)
__global__ __launch_bounds__(256) void patch_totals_kernel(PatchBuildArgs a)
{
    __shared__ unsigned long long s_t[2];
    if (threadIdx.x < 2u) s_t[threadIdx.x] = 0ull;
    __syncthreads();
    const uint64_t c = uint64_t(blockIdx.x) * 256u + threadIdx.x;
    uint64_t ns = 0, np = 0;
    if (c < a.n_chunks) { ns = (a.chunks[c].dst_n >> 48) & CHUNK_N_MASK; np = patch_chunk_patches(a.chunks[c].task_begin); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { ns += __shfl_xor(ns, o); np += __shfl_xor(np, o); }
    if ((threadIdx.x & 63u) == 0u) { atomicAdd(&s_t[0], (unsigned long long)ns); atomicAdd(&s_t[1], (unsigned long long)np); }
    __syncthreads();
    if (threadIdx.x == 0) { atomicAdd(reinterpret_cast<unsigned long long*>(a.totals), s_t[0]); atomicAdd(reinterpret_cast<unsigned long long*>(a.totals) + 1, s_t[1]); }
}

// ---- the executor ----------------------------------------------------------------------------------------------------------
// up to n (1..16) bytes of x, first byte lowest, OR-ed into the zeroed image at byte offset o (stitch_dense_kernel's put)
__device__ __forceinline__ void patch_put(uint32_t* img, const u32x4* lowmask, uint32_t o, u32x4 x, uint32_t n)
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

// (Round 5 also tried RESIDENT workgroups -- eight per CU, each walking chunks a grid apart with the next chunk's segment and patch words and
// the record after that requested a chunk ahead, so that a chunk's life held one memory round trip instead of three: 1.64 ms against
// 0.98 for BASELINE config 5.  gfx950 counts loads and stores in one in-order counter: a resident workgroup's gathers queue behind its
// own previous chunk's stores, as round 3 found for the stitch kernels.  One workgroup per chunk it is.)
// Everything a workgroup reads from memory is requested as early as its address is known: the chunk's segment words and patch words at
// once (registers), the segments' ragged first pieces as soon as the words are decoded -- they fly under the block map's construction --
// and the blocks' gathers right behind the map.  (The first form of this kernel went phase by phase and waited 81 % of its wave cycles:
// 1.12 ms for BASELINE config 5 at 20 000 haplotypes with half the vector instructions of stitch_dense_kernel's 0.85 ms.)
template <bool NT>
__global__ __launch_bounds__(256) void stitch_patch_kernel(PatchExecArgs a)
{
    constexpr uint32_t SPL = PATCH_SEG_CAP / 256u, PPL = PATCH_PATCH_CAP / 256u, BPL = PATCH_G / 16u / 256u;     // segments, patches, blocks per lane
    __shared__ __attribute__((aligned(16))) uint32_t s_img[PATCH_G / 4u + 8u];
    __shared__ uint64_t s_seg[PATCH_SEG_CAP];
    __shared__ uint16_t s_map[PATCH_G / 16u];             // segment covering the first byte of each 16-byte block
    __shared__ u32x4 s_low[17];                           // s_low[j]: the low j bytes of a block
    __shared__ uint32_t s_bad;
    const uint32_t tid = threadIdx.x;
    const uint32_t c = blockIdx.x;
    if (c >= a.n_chunks) return;
    const uint64_t tbw = a.chunks[c].task_begin, dn = a.chunks[c].dst_n;
    if ((dn & CHUNK_PATCH) != CHUNK_PATCH) { if (tid == 0) report(a.status, c, STATUS_RES_OOB); return; }     // (not a patch image's chunk: refused, not guessed)
    const uint64_t dst = dn & DST_MASK, seg0 = tbw & TB_IDX_MASK;
    const uint32_t n_seg = uint32_t(dn >> 48) & CHUNK_N_MASK, n_patch = patch_chunk_patches(tbw);
    if (dst % PATCH_G != 0u || dst >= a.out_len || n_seg > PATCH_SEG_CAP || n_patch > PATCH_PATCH_CAP || seg0 != (dst / PATCH_G) * PATCH_SEG_CAP) {
        if (tid == 0) report(a.status, c, STATUS_RES_OOB);
        return;
    }
    const uint32_t span = uint32_t(a.out_len - dst < PATCH_G ? a.out_len - dst : PATCH_G);
    const uint32_t* const patches = a.patch + (dst / PATCH_G) * PATCH_PATCH_CAP;
    // ---- requests: segment words, patch words ----
    uint64_t w[SPL];
    uint32_t pw[PPL];
#pragma unroll
    for (uint32_t k = 0; k < SPL; ++k) { const uint32_t s = tid + 256u * k; w[k] = s < n_seg ? a.seg[seg0 + s] : 0ull; }
#pragma unroll
    for (uint32_t k = 0; k < PPL; ++k) { const uint32_t q = tid + 256u * k; pw[k] = q < n_patch ? patches[q] : 0xFFFFu; }    // (position 0x3FFF: outside every chunk)
    if (tid < 17u) {
        u32x4 m;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) m[k] = tid >= 4u * k + 4u ? 0xFFFFFFFFu : (tid <= 4u * k ? 0u : (1u << (8u * (tid - 4u * k))) - 1u);
        s_low[tid] = m;
    }
    if (tid == 0) s_bad = 0u;
#pragma unroll
    for (uint32_t k = 0; k < BPL; ++k) s_map[tid + 256u * k] = 0xFFFFu;
    __syncthreads();
    auto first16 = [&](uint64_t ww, uint32_t off) -> u32x4 {             // 16 bytes of a segment's stream from its byte `off` on
        const unsigned space = patch_seg_space(ww);
        const uint64_t src = patch_seg_src(ww);
        if (space == SPACE_IMM) return u32x4{uint32_t(src) >> (8u * (off & 3u)), 0u, 0u, 0u};
        if (space == SPACE_FILL) return u32x4{0x2E2E2E2Eu, 0x2E2E2E2Eu, 0x2E2E2E2Eu, 0x2E2E2E2Eu};
        return gather16(reinterpret_cast<uint64_t>(space == SPACE_PROTEOME ? a.src0 : a.src1) + src + off);
    };
    // ---- lane = segment: checks, the ragged first piece's gather (in flight from here on), the word to LDS, the block map ----
    u32x4 g[SPL];
#pragma unroll
    for (uint32_t k = 0; k < SPL; ++k) {
        const uint32_t s = tid + 256u * k;
        g[k] = u32x4{0u, 0u, 0u, 0u};
        if (s < n_seg) {
            const uint32_t start = patch_seg_start(w[k]), len = patch_seg_len(w[k]);
            const unsigned space = patch_seg_space(w[k]);
            const uint64_t src = patch_seg_src(w[k]);
            bool ok = len != 0u && start + len <= span;                  // never write out of bounds
            if (space == SPACE_PROTEOME) ok = ok && src + len <= a.src0_len;  // never read out of bounds (task.rs would panic)
            else if (space == SPACE_PAYLOAD) ok = ok && src + len <= a.src1_len;
            else if (space == SPACE_IMM) ok = ok && len <= PATCH_IMM_MAX;
            if (!ok) { s_bad = 1u; report(a.status, seg0 + s, space == SPACE_IMM || len == 0u || start + len > span ? STATUS_RES_OOB : STATUS_SRC_OOB); w[k] = patch_seg(0, start, 0, SPACE_FILL); }
            else {
                if ((start & 15u) != 0u) g[k] = first16(w[k], 0u);
                for (uint32_t b = (start + 15u) >> 4; (b << 4) < start + len; ++b) s_map[b] = uint16_t(s);
            }
            s_seg[s] = w[k];
        }
    }
    __syncthreads();
    if (s_bad) return;                                                   // reported; the chunk is not executed
    // ---- lane = block: the covering segment's stream, cut where the segment ends (what follows comes from the segments that start inside the block) ----
    const uint32_t nblk = (span + 15u) >> 4;
    u32x4 v[BPL];
    uint32_t keep[BPL];
#pragma unroll
    for (uint32_t k = 0; k < BPL; ++k) {
        const uint32_t b = tid + 256u * k;
        v[k] = u32x4{0u, 0u, 0u, 0u}; keep[k] = 0u;
        if (b < nblk) {
            const uint32_t s = s_map[b];
            if (s != 0xFFFFu) {
                const uint64_t ww = s_seg[s];
                const uint32_t start = patch_seg_start(ww), end = start + patch_seg_len(ww), p = b << 4;
                v[k] = first16(ww, p - start);
                keep[k] = end - p < 16u ? end - p : 16u;
            }
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < BPL; ++k) {
        const uint32_t b = tid + 256u * k;
        if (b < nblk) {
            const u32x4 m = s_low[keep[k]];
            u32x4 x = v[k];
            x[0] &= m[0]; x[1] &= m[1]; x[2] &= m[2]; x[3] &= m[3];
            *reinterpret_cast<u32x4*>(&s_img[b * 4u]) = x;
        }
    }
    __syncthreads();
    // ---- lane = segment: a segment that starts inside a block ORs its first piece (up to the block's end) into the image ----
#pragma unroll
    for (uint32_t k = 0; k < SPL; ++k) {
        const uint32_t s = tid + 256u * k;
        if (s < n_seg) {
            const uint32_t start = patch_seg_start(w[k]), len = patch_seg_len(w[k]), q = start & 15u;
            if (q != 0u && len != 0u) patch_put(s_img, s_low, start, g[k], len < 16u - q ? len : 16u - q);
        }
    }
    __syncthreads();
    // ---- the substituted residues ----
#pragma unroll
    for (uint32_t k = 0; k < PPL; ++k) {
        const uint32_t pos = pw[k] & 0x3FFFu;
        if (pos < span) reinterpret_cast<uint8_t*>(s_img)[pos] = uint8_t(pw[k] >> 16);
    }
    __syncthreads();
    // ---- the image leaves as aligned 16-byte stores ----
    uint8_t* const out0 = a.out + dst;
#pragma unroll
    for (uint32_t k = 0; k < BPL; ++k) {
        const uint32_t b = tid + 256u * k;
        if (b >= nblk) continue;
        const u32x4 x = *reinterpret_cast<const u32x4*>(&s_img[b * 4u]);
        const uint32_t p = b << 4;
        if (p + 16u <= span) {
            if (NT) __builtin_nontemporal_store(x, reinterpret_cast<u32x4*>(out0 + p));
            else *reinterpret_cast<u32x4*>(out0 + p) = x;
        } else for (uint32_t q = 0; p + q < span; ++q) out0[p + q] = uint8_t(x[q >> 2] >> (8u * (q & 3u)));     // (the arena's last, ragged block)
    }
}

__global__ void code_object_loader_e() {}

}  // namespace

hipError_t launch_patch_positions(const PatchBuildArgs& a, uint64_t* scan_scratch, hipStream_t stream)
{
    if (a.n_tx) hipLaunchKernelGGL(patch_arena_len_kernel, dim3(uint32_t((a.n_tx + 255) / 256)), dim3(256), 0, stream, a);
    return launch_scan_u32(a.tx_arena_len, a.n_tx, a.tx_res_base, scan_scratch, stream);
}

hipError_t launch_patch_hap_begin(const PatchBuildArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(patch_hap_begin_kernel, dim3(uint32_t((a.n_haps + 1 + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_patch_build(const PatchBuildArgs& a, hipStream_t stream)
{
    if (a.n_chunks == 0) return hipSuccess;
    if (a.n_chunks > 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(patch_chunk_tx_kernel, dim3(uint32_t((a.n_tx + 1 + 255) / 256)), dim3(256), 0, stream, a);
    if (a.tx_header_len) hipLaunchKernelGGL(patch_build_kernel<true>, dim3(uint32_t(a.n_chunks)), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(patch_build_kernel<false>, dim3(uint32_t(a.n_chunks)), dim3(256), 0, stream, a);
    hipLaunchKernelGGL(patch_totals_kernel, dim3(uint32_t((a.n_chunks + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_stitch_patch(const PatchExecArgs& a, hipStream_t stream, bool nontemporal)
{
    if (a.n_chunks == 0) return hipSuccess;
    if (nontemporal) hipLaunchKernelGGL(stitch_patch_kernel<true>, dim3(a.n_chunks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(stitch_patch_kernel<false>, dim3(a.n_chunks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t preload_patch_image(hipStream_t stream)
{
    hipLaunchKernelGGL(code_object_loader_e, dim3(1), dim3(64), 0, stream);
    return hipGetLastError();
}

}  // namespace v2p
