/* v2p_bench.h -- entry points of libv2p_bench.so: micro-benchmarks of the stitch kernels' data movement (tools/copy_bench.py,
 * copy_mix.py, hbm_ceiling.py, gather_ceiling.py, wave_copy_bench.py).  Development tools; not part of the engine's C ABI
 * (include/vcf2prot_hip.h).  libv2p_bench.so also carries a V2P_BENCH_VARIANTS build of the engine itself, whose
 * v2p_stitch_launch() accepts the timing-only ablation bits (results are wrong) that libvcf2prot_hip.so refuses. */
#ifndef V2P_BENCH_H
#define V2P_BENCH_H
#include <stdint.h>
#include "../../../include/vcf2prot_hip.h"
#ifdef __cplusplus
namespace v2p {
hipError_t launch_gather_bench(const uint8_t* src, uint64_t window, uint32_t misalign, uint32_t iters, uint32_t blocks,
                               uint32_t* sink, hipStream_t stream);
hipError_t launch_copy_bench(const uint8_t* src, uint64_t window, uint32_t shift, uint8_t* out, uint64_t bytes, int mode,
                             uint32_t* sink, hipStream_t stream);
hipError_t launch_copy_prefetch(const uint8_t* src, uint64_t window, uint32_t shift, uint8_t* out, uint64_t bytes, int depth, int desc_bytes,
                                uint32_t grid, hipStream_t stream);
hipError_t launch_copy_mix(const uint8_t* src, uint64_t window, uint32_t shift, uint8_t* out, uint64_t bytes, const uint8_t* dsc,
                           uint32_t bytes_per_lane, uint32_t every, uint32_t stride, uint32_t flags, hipStream_t stream);
hipError_t launch_fill(uint8_t* out, uint64_t bytes, uint32_t word, int nontemporal, hipStream_t stream);
}
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif
/* the engine's launcher with the packed flag word (V2P_BENCH_VARIANTS build only): bit 0 non-temporal stores | v2p_stitch_launch_bits() |
 * kernel variant << 12 (1 / 2 per-block gathers, 4..6 stitch4 rows per round, 7 / 11 LDS-staged reference, 9 dword-aligned dense gathers) |
 * timing-only ablation << 16 (results are wrong) | KiB of idle LDS << 24 | waves per workgroup selector << 28; V2P_PHASE_BYTES,
 * V2P_WAVE_SC1, V2P_PHASE_MIN_CHUNKS, V2P_PHASE_NO_TOUCH / _OWN_TOUCH / _ONE_LAUNCH are read from the environment at every call */
int v2p_stitch_launch(void* hip_stream, const uint64_t* d_desc, uint64_t n_desc, const v2p_chunk* d_chunks, uint32_t n_chunks,
                      const uint8_t* d_src0, uint64_t src0_len, const uint8_t* d_src1, uint64_t src1_len,
                      uint8_t* d_out, uint64_t out_len, uint64_t* d_status, int nontemporal, uint32_t max_blocks);
int v2p_fill_launch(void* hip_stream, uint8_t* d_out, uint64_t bytes, uint32_t word, int nontemporal);

/* ---- what no routing rule of the product picks (moved out of include/vcf2prot_hip.h in round 6; V2P_BENCH_VARIANTS build only) ----------------
 * The engine's own entry points keep their signatures; in libv2p_bench.so they ALSO accept:
 *
 * v2p_batch_build_on_device(b, s, window_bytes, kernel, build_ms), the GRID builders of rounds 2-3: chunks cut on a fixed grid of `window_bytes`
 * of result (a multiple of 4096 -- wave images: of 1024 --, <= 65536 - 4096; a window holding more descriptors than its kernel takes is
 * V2P_ERR_UNSUPPORTED: pick a smaller one):
 * kernel 1 .. 5 -- the grid builders of rounds 2 and 3, kept for the kernels only they feed:
 * 1 = route every chunk to the long-run kernel (<= 512 descriptors per window), 2 = per-block kernel, 3 = dense image
 * (stitch_dense_kernel: short tasks, fused substitutions, <= 1024 descriptors per window: windows of 4, 8 or 12 KiB, larger ones are
 * V2P_ERR_INVALID_ARG), 4 = wave image (stitchw_kernel, one wave per window: windows of 1 .. 10 KiB in steps of 1 KiB with <= 64
 * descriptors each, fused substitutions; the choice for long reference runs), 5 = wave image whose windows may SPLIT ONCE (2 .. 10 KiB:
 * a window of 65 .. 127 descriptors becomes two chunks, cut on the 1 KiB row nearest its middle that leaves both with <= 64, the
 * descriptor under the cut split in two -- every window carries one spare descriptor slot for that; the grid can then be as coarse
 * as the AVERAGE window allows: C3 at 8 KiB executes within 2 % of the host packer's greedy cuts, at the 4 KiB kernel 4 needs 30 % slower).
 *
 * kernel 8 of v2p_batch_build_on_device / _build_from_stream / _build_and_execute -- PATCH images:
 * The image of a batch of DEEP Task vectors (a few result bytes per Task: 64 alterations in an 800-residue transcript) whose commonest
 * Task triple -- reference copy, ONE substituted residue, the reference going on one residue later (a missense,
 * transcript_instructions.rs:654-663) -- does not end a copy: SEGMENTS (8 bytes: source:34 | start inside the chunk:14 | length:14 |
 * space:2 -- a run of one source, under any number of substituted residues) and PATCHES (4 bytes: position inside the chunk:14 | byte
 * << 16) on a fixed grid of 8 KiB chunks of the arena; chunk k's segments sit in slots [1024 k, 1024 (k + 1)) of the segment array, its
 * patches in slots [1024 k, ..) of the patch array, in no particular order; the chunk record holds first segment slot | patches << 42
 * and arena offset | segments << 48 | bits 60 and 61.  Built by ONE kernel (one workgroup per chunk; no count pass, no scan of
 * descriptor counts, no compaction, no cutter), executed by stitch_patch_kernel (vcf2prot_amd/csrc/patch_image.hip).  Semantics and
 * panics are the reference's (task.rs:38-50, haplotype_instruction.rs:78,140-158).  V2P_ERR_UNSUPPORTED: a window of the result holds more
 * segments or patches than its slots (or the sources exceed 16 GB) -- the batch is left empty: build a dense rows image (kernel 7).
 * v2p_batch_counts reports the segments as descriptors.
 *
 * v2p_batch_build_and_execute(.., n_slices > 1): the image built slice by slice on a second HIP stream while the slice before it is stitched
 * (<= 32 slices; measured slower on every cohort: profiles/r05_oneshot_slices.json).
 *
 * v2p_launch_opts.reserved (v2p_stitch_launch_opts): 3 = per-block kernel also where the dense one would be picked, 8 = the dense kernel for
 * every per-block chunk (routing-only A/B switches). */
/* For checkers of PATCH images: the raw arrays (1024 / 1024 slots per chunk, the chunk table in launch order; any pointer may be NULL) and the totals. */
int v2p_batch_download_patch_image(v2p_batch* b, uint64_t* seg, uint32_t* patch, v2p_chunk* chunks, uint64_t* n_segments, uint64_t* n_patches);
/* The A/B switches of the builders and launchers of a context (0: the product's rules):
 * 16 = ONE launch for all phases of a wave image (read-ahead workgroups of phase g + 1 in the grid before the stitch workgroups of phase g),
 * 17 = the read-ahead as kernels of its own, 18 = no read-ahead, 19 = the phases in halves on two launch streams;
 * 20 / 21 = the one call's builder: tiles dealt to the XCDs by workgroup index / tile tables made inside the call;
 * 22 / 24 = a wave image is compacted / stays padded whatever the rule says; 23 / 25 = no staging of a padded image's descriptors / dense rows
 * images staged as well; 23 / 26 = a padded image stays padded when it is executed again (read in place / staged);
 * 27 = a padded image is built in three slices whose parses are launched ahead of the cutters (measured slower);
 * 28 = deep Task vectors stay on the dense rows image and stitch_dense_kernel (never a tile image, never re-written as pieces);
 * 29 = v2p_batch_build_and_execute behaves as if the device had no room for its one-pass scratch (tests of the fallback) */
int v2p_bench_set_variant(v2p_ctx* ctx, uint32_t variant);
/* microbenchmark: `blocks` workgroups x 4 waves each issue `iters` 16-byte-per-lane gathers (1 KiB per wave
 * instruction) from a window of `window` bytes at byte misalignment `misalign` (0 = aligned); d_sink: one u32 per wave */
/* microbenchmark: the stitch kernel's data movement without its bookkeeping -- a cache-resident window of `window` bytes
 * (>= 1 MiB + 64 KiB, 64 readable bytes of slack around it) read at byte misalignment `shift` and streamed into d_out with
 * non-temporal stores; mode 0 byte-granular gathers, 1 aligned loads + lane exchange, 2 two aligned loads, 3 dword-aligned
 * loads, 4 stores only, 5 loads only; d_sink: one u32 per 32 KiB of d_out */
int v2p_copy_bench_launch(void* hip_stream, const uint8_t* d_src, uint64_t window, uint32_t shift, uint8_t* d_out, uint64_t bytes,
                          int mode, uint32_t* d_sink);
/* microbenchmark: the same copy with persistent workgroups (`grid` of them) whose per-span "descriptor" (desc_bytes = 8, 4 or 0
 * bytes per lane, streamed from HBM behind d_out) is requested `depth` spans before it is used */
int v2p_copy_prefetch_launch(void* hip_stream, const uint8_t* d_src, uint64_t window, uint32_t shift, uint8_t* d_out, uint64_t bytes,
                             int depth, int desc_bytes, uint32_t grid);
/* microbenchmark: the copy of v2p_copy_bench_launch (mode 0) plus a streamed read of `bytes_per_lane` (4, 8, 16) bytes per lane by
 * every `every`-th workgroup from d_desc (pieces `stride` bytes apart); flags bit 0 non-temporal loads, bit 1 one wave only,
 * bit 2 read issued after the first store */
int v2p_copy_mix_launch(void* hip_stream, const uint8_t* d_src, uint64_t window, uint32_t shift, uint8_t* d_out, uint64_t bytes,
                        const uint8_t* d_desc, uint32_t bytes_per_lane, uint32_t every, uint32_t stride, uint32_t flags);
int v2p_gather_bench_launch(void* hip_stream, const uint8_t* d_src, uint64_t window, uint32_t misalign, uint32_t iters,
                            uint32_t blocks, uint32_t* d_sink);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* V2P_BENCH_H */
