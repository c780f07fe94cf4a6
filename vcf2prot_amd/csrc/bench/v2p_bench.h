/* v2p_bench.h -- entry points of libv2p_bench.so: micro-benchmarks of the stitch kernels' data movement (tools/copy_bench.py,
 * copy_mix.py, hbm_ceiling.py, gather_ceiling.py, wave_copy_bench.py).  Development tools; not part of the engine's C ABI
 * (include/vcf2prot_hip.h).  libv2p_bench.so also carries a V2P_BENCH_VARIANTS build of the engine itself, whose
 * v2p_stitch_launch() accepts the timing-only ablation bits (results are wrong) that libvcf2prot_hip.so refuses. */
#ifndef V2P_BENCH_H
#define V2P_BENCH_H
#include <stdint.h>
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
/* the engine's launcher with the packed flag word (V2P_BENCH_VARIANTS build only): bit 0 non-temporal stores | v2p_stitch_launch_bits() |
 * kernel variant << 12 (1 / 2 per-block gathers, 4..6 stitch4 rows per round, 7 / 11 LDS-staged reference, 9 dword-aligned dense gathers) |
 * timing-only ablation << 16 (results are wrong) | KiB of idle LDS << 24 | waves per workgroup selector << 28; V2P_PHASE_BYTES,
 * V2P_WAVE_SC1, V2P_PHASE_MIN_CHUNKS, V2P_PHASE_NO_TOUCH / _OWN_TOUCH / _ONE_LAUNCH are read from the environment at every call */
int v2p_stitch_launch(void* hip_stream, const uint64_t* d_desc, uint64_t n_desc, const void* d_chunks, uint32_t n_chunks,
                      const uint8_t* d_src0, uint64_t src0_len, const uint8_t* d_src1, uint64_t src1_len,
                      uint8_t* d_out, uint64_t out_len, uint64_t* d_status, int nontemporal, uint32_t max_blocks);
int v2p_fill_launch(void* hip_stream, uint8_t* d_out, uint64_t bytes, uint32_t word, int nontemporal);
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

#ifdef __cplusplus
}
#endif
#endif /* V2P_BENCH_H */
