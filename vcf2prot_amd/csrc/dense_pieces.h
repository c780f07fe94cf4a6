// dense_pieces.h -- PIECE images: a dense rows image (sir_pack.hpp: chunks of up to 1 024 descriptors and twelve 1 KiB rows, C5's deep Task
// vectors) re-written for RE-execution as a list of PIECES -- at most sixteen result bytes from one source each, position inside the
// chunk and at most one substituted residue included -- so that the executor (stitch_pieces_kernel) has nothing to decode, nothing to
// scan and no list of continuation pieces to build: lane = piece, one record load, one gather, one put into the chunk's LDS image.
// Built on the device from the image a batch already holds, the first time the batch is executed AGAIN (v2p_api.hip: to_pieces);
// a cohort that is executed once never pays for it.  Semantics: task.rs:38-50 for every descriptor of the image, unchanged.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sir_pack.hpp"

namespace v2p {

// a piece (8 bytes): src:31 | space:2 | offset inside the chunk's result:14 | bytes - 1:4 | position of the substituted residue:4 | has one:1 | its byte:8
// -- up to 16 result bytes of one source (one gather, one put), at most one substituted residue.  (Pieces of up to 32 bytes -- a second
// gather and put for the lanes that have them -- were measured: 0.94 ms against 0.73 for C5, the wave runs both puts.)
// (space 3, an immediate: bits 0..30 and 51..63 ARE the bytes -- up to five, first byte lowest)
constexpr uint32_t PIECE_SRC_BITS = 31;
constexpr uint32_t PIECE_BYTES = 16;
constexpr uint64_t PIECE_SRC_MAX = (1ull << PIECE_SRC_BITS) - 1;   // sources beyond 2 GiB: the image is not converted (the dense kernel keeps executing it)
constexpr uint32_t PIECE_CHUNK_MAX = 2047;                       // pieces per chunk (count field of the chunk record: 11 bits)
// chunk record: task_begin = first piece : 40 | bytes of the chunk : 14 << 40;  dst_n = result offset : 48 | pieces : 11 << 48 | CHUNK_DENSE | CHUNK_CLIP

struct PieceBuildArgs {
    const uint64_t* desc;            // the dense rows image
    uint64_t        n_desc;
    const Chunk*    chunks;          // in launch order (kept)
    uint32_t        n_chunks;
    uint64_t        src0_len, src1_len, out_len;
    uint32_t*       count;           // [n_chunks] pieces of every chunk (pass 0) -> scan -> base
    const uint64_t* base;            // [n_chunks + 1]
    uint64_t*       pieces;          // pass 1
    Chunk*          chunks2;         // pass 1
    unsigned long long* status;      // ~0: clean; else index << 8 | reason (a descriptor the dense kernel would refuse, a chunk too large)
};
// pass 0: count (and validate); pass 1: write pieces and chunk records
hipError_t launch_pieces_build(const PieceBuildArgs& a, int pass, hipStream_t stream);

struct PieceExecArgs {
    const uint64_t* pieces;
    const Chunk*    chunks2;
    uint32_t        n_chunks;
    const uint8_t*  src0;            // 32 readable bytes before and after (as for every stitch kernel)
    const uint8_t*  src1;
    uint8_t*        out;
    uint64_t        out_len;
};
hipError_t launch_stitch_pieces(const PieceExecArgs& a, hipStream_t stream, bool nontemporal);
hipError_t preload_dense_pieces(hipStream_t stream);

// ---- TILE images (round 6): pieces STRAIGHT FROM THE PARSE ----------------------------------------------------------------------------
// The one-pass parse of build_rows.hip (ROWS_TILES) writes pieces instead of descriptors: tile t -- K consecutive transcripts, one wave of
// the parse -- owns slots [tile_slots t, tile_slots (t + 1)) of the piece array, filled from the first, tile_count[t] of them; a piece's
// offset is relative to the tile's first result byte, tile_res_base[t] (res_counter of haplotype_instruction.rs:90,132 per tile).  The tile
// IS the executor's work item: no dense image first, no compaction, no row map, no cutter, no chunk table, no XCD sort, and nothing to
// re-write when the image is executed again -- what a deep cohort's one call runs is the parse and this kernel.  A tile's result range
// starts anywhere (not on a 1 KiB row): its LDS image is shifted by the range's offset inside its first 16-byte block, whole blocks leave
// as aligned stores, the ragged first and last block byte by byte (two neighbouring tiles share that block, each writes its own bytes).
constexpr uint32_t TILE_SPAN_MAX = 16368;          // result bytes of a tile (the 14-bit offset field, minus the shift)
constexpr uint32_t TILE_SLOTS_MAX = 2048;          // pieces of a tile (eight rounds of the executor's 256 lanes)
struct TileExecArgs {
    const uint64_t* pieces;          // [n_tiles * tile_slots]
    uint32_t        tile_slots;
    const uint32_t* tile_count;      // [n_tiles]
    const uint64_t* tile_res_base;   // [n_tiles + 1]
    uint64_t        n_tiles;         // tiles of this launch: [tile0, tile0 + n_tiles)
    const uint8_t*  src0;            // 32 readable bytes before and after (as for every stitch kernel)
    const uint8_t*  src1;
    uint8_t*        out;
    uint64_t        out_len;
    const unsigned long long* status;    // the build's status word: a build that reported anything is not executed
    uint64_t        tile0 = 0;       // first tile of the launch (the whole image: 0)
};
hipError_t launch_stitch_tiles(const TileExecArgs& a, hipStream_t stream, bool nontemporal);

}  // namespace v2p
