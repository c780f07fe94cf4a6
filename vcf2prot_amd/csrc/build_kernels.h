// build_kernels.h -- argument block and host launchers of the device-side image builder (build_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sir_pack.hpp"
#include "stitch_kernels.h"

namespace v2p {

constexpr uint32_t STATUS_TOO_MANY = 5;     // a grid window holds more descriptors than a chunk may

struct BuildArgs {
    // the transcript stream (v2p_txstream), on the device
    uint64_t n_haps, n_tx, n_tasks;
    const uint64_t* hap_tx_begin;
    const uint64_t* tx_proteome_off;
    const uint32_t* tx_ref_len;
    const uint32_t* tx_res_len;
    const uint64_t* tx_task_begin;
    const uint64_t* tx_alt_begin;
    const uint8_t*  code;
    const uint32_t* start_pos;
    const uint32_t* length;
    const uint32_t* start_pos_res;
    const uint8_t*  alt;
    const uint64_t* tx_header_off;   // FASTA emit (nullptr: plain tapes): record header of transcript t in the resident header table ...
    const uint32_t* tx_header_len;   // ... and its length (0: no record text for this transcript)
    uint64_t proteome_len;
    uint32_t window;              // result bytes per chunk (grid)
    int      long_run;            // route every chunk to stitch4_kernel
    int      dense;               // ... to stitch_dense_kernel (fusion on as for long_run)
    int      wave;                // with long_run: ... to stitchw_kernel instead (windows of <= 10 KiB and <= 64 descriptors)
    int      split;               // with wave: a window of 65 .. 127 descriptors becomes TWO chunks, cut on a 1 KiB row (the descriptor under the
                                  // cut is split in two; every window carries a spare descriptor slot for that).  chunks_tmp / bucket / sub then hold
                                  // 2 * n_windows entries: window k's (first) chunk at k, second chunks from n_windows on (meta[4] counts them)
    // scans and outputs
    const uint64_t* tx_res_base;  // [n_tx + 1] exclusive prefix of the transcripts' arena lengths (tx_res_len, + header + line feed with FASTA emit)
    uint32_t* tx_desc_count;      // [n_tx]
    const uint64_t* desc_base;    // [n_tx + 1] exclusive prefix of tx_desc_count
    uint64_t* desc;
    uint64_t* chunk_first;        // [n_windows] index of the first descriptor of each chunk
    Chunk*    chunks_tmp;         // [n_windows] in result order
    uint8_t*  bucket;             // [n_windows] proteome slice of each chunk
    uint8_t*  sub;                // [n_windows] window of that slice (xcd_sub_window)
    uint64_t* hap_out_begin;      // [n_haps + 1]
    uint32_t* meta;               // [8]: any long-run chunk, any with > 256 descriptors, any per-block chunk, most descriptors of a per-block chunk, second chunks of split windows
    unsigned long long* status;
};

hipError_t launch_scan_u32(const uint32_t* in, uint64_t n, uint64_t* out, uint64_t* tile_scratch, hipStream_t stream);
hipError_t launch_scan_u32_from(const uint32_t* in, uint64_t n, uint64_t* out, uint64_t* tile_scratch, uint64_t first, hipStream_t stream);
hipError_t launch_scan_u32_chained(const uint32_t* in, uint64_t n, uint64_t* out, uint64_t* tile_scratch, hipStream_t stream);   // first = out[0] as the slice before left it   // ... + first
// phase 0: count descriptors per transcript (and validate); phase 1: emit descriptors + chunk table + hap_out_begin
hipError_t launch_build(const BuildArgs& a, uint64_t n_windows, uint64_t n_desc, uint64_t out_bytes, int phase, hipStream_t stream);
hipError_t launch_xcd_order(const Chunk* in, const uint8_t* bucket, uint64_t n, uint32_t* hist, Chunk* out, hipStream_t stream);
// stable counting sort of the table by window (the pass before launch_xcd_order): hist = XCD_SUB * n_blocks u32, start = that many
// + 1 u64, tiles = scratch of launch_scan_u32 for that many entries; out / out_bucket receive the permuted table and slices
hipError_t launch_sub_order(const Chunk* in, const uint8_t* bucket, const uint8_t* sub, uint64_t n, uint32_t* hist, uint64_t* start, uint64_t* tiles,
                            Chunk* out, uint8_t* out_bucket, hipStream_t stream);
uint64_t scan_tiles_for(uint64_t n);
// both sorts over the first n entries of a table in arena order, inside nb blocks of it (sir_pack.hpp: xcd_order_blocks /
// xcd_order_block_first), one launch each for all blocks; subhist / substart / tiles sized for XCD_SUB * order_blocks_thread_blocks(n, nb)
// counters, hist8 for 8 * that many, tot for 8 * nb
hipError_t launch_order_blocks(const Chunk* in, const uint8_t* bucket, const uint8_t* sub, uint64_t n, uint32_t nb, uint32_t* subhist, uint64_t* substart,
                               uint64_t* tiles, Chunk* by_window, uint8_t* bucket2, uint32_t* hist8, uint32_t* tot, Chunk* out, hipStream_t stream);
uint64_t order_blocks_thread_blocks(uint64_t n, uint32_t nb);

}  // namespace v2p
