// build_rows.h -- argument block and host launchers of the ONE-PASS device image builder (build_rows.hip; image format: rows_image.hpp).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sir_pack.hpp"
#include "rows_image.hpp"
#include "stitch_kernels.h"

namespace v2p {

constexpr uint32_t STATUS_ROWS_TOO_MANY = 5;       // a 1 KiB row holds more descriptors than a chunk may (= build_kernels.h: STATUS_TOO_MANY)
constexpr uint32_t STATUS_ROWS_STAGE = 6;          // a tile's descriptors / rows do not fit the wave's LDS stage: rebuild with the two-phase kernel
constexpr uint32_t STATUS_ROWS_CAP = 7;            // the descriptor array is too small (tasks behind gaps): rebuild with the full bound
constexpr uint32_t STATUS_ROWS_SPAN = 8;           // a tile of transcripts spans more than 2 GiB of result

struct RowsArgs {
    // the transcript stream (v2p_txstream), on the device
    uint64_t n_tx, n_tasks, n_alt;
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
    const uint64_t* tx_header_off;      // FASTA emit (nullptr: plain tapes)
    const uint32_t* tx_header_len;
    const uint64_t* hap_tx_begin;
    uint64_t n_haps;
    uint64_t proteome_len;
    uint64_t headers_len = 0;           // the FASTA header table behind the proteome (sources of header / line-feed runs: proteome_len + offset)
    // tiles: K consecutive transcripts each, 1 <= K <= 64
    uint32_t K;
    uint64_t n_tiles;
    // the range of tiles / cutter segments one launch works on (v2p_batch_build_and_execute builds an image slice by slice while the
    // slice before it executes; the one-call builder: everything).  Tile and segment numbers stay global; desc_pad holds the RANGE's tiles.
    uint64_t tile0 = 0, tile1 = 0;      // parse, compact: tiles [tile0, tile1)  (0, 0: all)
    uint64_t seg0 = 0, seg1 = 0;        // cut, chunk compact: segments [seg0, seg1)  (0, 0: all)
    uint64_t* tile_bytes;               // [n_tiles] arena bytes of each tile -> (scan) tile_res_base
    uint64_t* tile_res_base;            // [n_tiles + 1]
    uint32_t* tile_count;               // [n_tiles] descriptors of each tile -> (scan) tile_desc_base
    uint64_t* tile_desc_base;           // [n_tiles + 1]
    uint64_t* desc_pad;                 // [n_tiles * ROWS_PAD] the tiles' descriptors before compaction
    uint32_t  pad_chunks = 0;           // the cutter's records (and the keys) address the PADDED array (sir_pack.hpp: ROWS_TILE_SLOTS slots per tile, GLOBAL tile
                                        // numbers: desc = the whole array, desc_pad = its first slot of tile0); totals[3] bit 1: a chunk did not fit the form
    uint32_t  xcd_tiles = 1;            // the parse deals contiguous eighths of its tiles to the XCDs (0: tile = workgroup index; A/B switch)
    uint32_t  tile_slots = 0;           // TILE images (ROWS_TILES; dense_pieces.h): piece slots per tile in desc_pad (<= 2048) ...
    uint32_t  tile_span_max = 0;        // ... and the most result bytes a tile may hold (the executor's LDS image; <= 16368)
    uint64_t* totals;                   // [4]: -, -, result offset of the last chunk, -
    uint64_t* desc;
    uint64_t  desc_cap;
    uint64_t* cover;                    // [n_rows] per 1 KiB row of the arena, the descriptor covering its first byte: tile : 26 | index inside the tile : 16 |
                                        // offset inside the descriptor : 22 -- or, from the two-pass form, 1 << 63 | index << 22 | offset
    uint64_t  n_rows, out_bytes;
    // the cutter
    uint32_t* seg_count;                // [n_segs] chunks of each segment of ROWS_SEG rows
    const uint64_t* seg_base;           // [n_segs + 1] exclusive prefix
    uint64_t  n_segs;
    Chunk*    chunks_tmp;               // arena order
    uint64_t  chunk_cap = ~0ull;        // records chunks_tmp (and bucket / sub) have room for: rows_chunk_compact_kernel never writes past it
    Chunk*    chunks_pad;               // [n_segs * chunk_pad] the cutter's single pass
    uint32_t  chunk_pad = 96;           // chunk slots per segment in chunks_pad (rows_chunk_pad_for)
    uint8_t*  bucket;
    uint8_t*  sub;
    uint32_t  hap_major = 0;            // the chunks' window keys are all 0: haplotype-major inside an XCD's slice (sir_pack.hpp: xcd_order_window_major)
    uint64_t* hap_out_begin;
    unsigned long long* status;
};

inline void rows_ranges(const RowsArgs& a, uint64_t& t0, uint64_t& t1, uint64_t& s0, uint64_t& s1)
{
    t0 = a.tile0; t1 = a.tile1 ? a.tile1 : a.n_tiles;
    s0 = a.seg0; s1 = a.seg1 ? a.seg1 : a.n_segs;
}
// arena bytes per tile (+ u64 exclusive scan into tile_res_base, total behind the last tile)
hipError_t launch_rows_tile_bytes(const RowsArgs& a, uint64_t* scan_scratch, hipStream_t stream);
uint64_t rows_scan_scratch_entries(uint64_t n);
constexpr uint32_t ROWS_CHUNK_PAD = 96;            // chunk slots per segment of 640 rows in the cutter's padded table, at least (64 ten-row chunks tile a segment)
// ... and for a stream of these sizes: a chunk takes rows while its descriptors fit its kernel (64 / 1024), so a stream of b result bytes
// per Task fills a chunk after about max_desc * b / 1024 rows -- 1.5 x the segment's chunks at that rate, between 96 and 640 (one row per
// chunk: the most a segment can hold).  Until round 5 the table had 96 slots whatever the stream: every wave image between 24 and ~100 bytes
// per Task overflowed it, and the one call then built a second time in one piece.  16 bytes per slot, touched only where chunks are written.
inline uint32_t rows_chunk_pad_for(uint64_t out_bytes, uint64_t n_items, int mode)
{
    const uint64_t max_rows = mode == ROWS_DENSE ? ROWS_MAX_DENSE : ROWS_MAX_WAVE, max_desc = mode == ROWS_DENSE ? CHUNK_TASKS_DEEP : CHUNK_TASKS_WAVE;
    uint64_t rows = n_items ? max_desc * (out_bytes / n_items) / ROW_BYTES : max_rows;
    if (rows < 1) rows = 1;
    if (rows > max_rows) rows = max_rows;
    uint64_t pad = (3u * ((ROWS_SEG + rows - 1) / rows) / 2u + 31u) & ~uint64_t(31);
    if (pad < ROWS_CHUNK_PAD) pad = ROWS_CHUNK_PAD;
    if (pad > ROWS_SEG) pad = ROWS_SEG;
    return uint32_t(pad);
}
constexpr uint32_t ROWS_PAD_SLOTS = ROWS_TILE_SLOTS;           // descriptor slots per tile in the padded array (= build_rows.hip: ROWS_PAD)
// the parse: mode ROWS_WAVE / ROWS_DENSE; phase 0: descriptors into the padded array + tile_count (a tile that does not fit its slots
// is reported: STATUS_ROWS_STAGE), 1: tile_count only, 2: descriptors straight to desc + tile_desc_base[tile] (the two-pass form: any tile)
hipError_t launch_rows_parse(const RowsArgs& a, int mode, bool fasta, int phase, hipStream_t stream);
// padded -> dense (tile_desc_base = the scan of tile_count)
hipError_t launch_rows_compact(const RowsArgs& a, hipStream_t stream);
hipError_t launch_rows_hap_begin(const RowsArgs& a, hipStream_t stream);
// the cutter, pass 0: count (seg_count, totals[2] = last chunk's result offset); 1: emit at seg_base (the scan of seg_count) into
// chunks_tmp; 2: count AND emit into chunks_pad (chunk_pad slots per segment; totals[3] != 0: a segment did not fit, run pass 1)
hipError_t launch_rows_cut(const RowsArgs& a, int mode, int pass, hipStream_t stream);
// pass 2's padded table -> chunks_tmp in arena order
hipError_t launch_rows_chunk_compact(const RowsArgs& a, hipStream_t stream);
// a padded image's chunk records -> the dense image's (v2p_batch_download_image)
hipError_t launch_rows_chunks_dense(const Chunk* in, uint64_t n, const uint64_t* tile_desc_base, Chunk* out, hipStream_t stream);
hipError_t launch_rows_keys(const RowsArgs& a, uint64_t n_chunks, uint64_t n_desc, hipStream_t stream);
// totals[4..6] = *desc_end (null: 0), *chunk_end (null: 0), *status: what the host reads once per slice, in one block
hipError_t launch_rows_summary(const uint64_t* desc_end, const uint64_t* chunk_end, const unsigned long long* status, uint64_t* totals, hipStream_t stream);

}  // namespace v2p
