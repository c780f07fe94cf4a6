// patch_image.h -- PATCH images (round 5): the device image of a batch of DEEP Task vectors (a few result bytes per Task: BASELINE config 5,
// 64 alterations in an 800-residue transcript) and the kernels that build and execute it (patch_image.hip).
//
// What limits stitch_dense_kernel on such a batch is its vector instructions per DESCRIPTOR (decode, a first-piece put, a piece list):
// three rounds of tuning left it at 0.35 of the memory's rate.  Most Tasks of such a vector are one shape, though: a reference copy, ONE
// substituted residue, the reference going on one residue later (a missense, transcript_instructions.rs:654-663).  A patch image keeps
// that shape out of the copy path altogether:
//   SEGMENTS  a run of result bytes that is ONE contiguous run of a source -- the reference under any number of substituted residues, an
//             alt payload, a literal of up to four bytes, '.' fill (haplotype_instruction.rs:78), FASTA record text.  A missense does not
//             end a segment: the reference run simply goes on beneath it.  8 bytes: src:34 | start:14 | len:14 | space:2, start relative to
//             the segment's chunk.
//   PATCHES   the substituted residues: position inside the chunk and the byte, 4 bytes each, applied after the copy.
//   CHUNKS    a fixed grid of 8 KiB windows of the result arena (PATCH_G), one workgroup each, segments and patches clipped to it.  A
//             chunk's segments sit in its own PATCH_SEG_CAP slots of the segment array, its patches in its PATCH_PATCH_CAP slots of the
//             patch array (in no particular order: every segment says where it starts), its counts in the chunk record -- so the builder
//             needs no count pass, no scan of descriptor counts, no compaction and no cutter: ONE kernel, one workgroup per chunk.
// The reference's semantics are unchanged -- task.rs:38-50 for every Task, '.' where no Task writes; update_task's and Task::execute's
// panics (haplotype_instruction.rs:140-158, task.rs:43,47) are reported by row as the rows builder reports them.  A stream the format
// does not take (more segments or patches in a window than its slots, sources beyond 16 GB) is declined, not mangled: the caller
// builds a dense rows image instead.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sir_pack.hpp"
#include "patch_format.hpp"
#include "stitch_kernels.h"

namespace v2p {

struct PatchBuildArgs {
    // the transcript stream on the device (v2p_txstream)
    uint64_t n_tx, n_tasks, n_haps;
    const uint64_t* tx_proteome_off; const uint32_t* tx_ref_len; const uint32_t* tx_res_len;
    const uint64_t* tx_task_begin; const uint64_t* tx_alt_begin;
    const uint8_t* code; const uint32_t* start_pos; const uint32_t* length; const uint32_t* start_pos_res; const uint8_t* alt;
    const uint64_t* tx_header_off; const uint32_t* tx_header_len;      // FASTA emit (nullptr: plain tapes)
    const uint64_t* hap_tx_begin;
    uint64_t proteome_len, alt_len;
    // res_counter of haplotype_instruction.rs:90,132 as a scan: arena offset of every transcript's record ([n_tx + 1])
    uint32_t* tx_arena_len;             // [n_tx] scratch: result length (+ header + line feed)
    uint64_t* tx_res_base;
    uint64_t out_bytes, n_chunks;
    uint64_t* chunk_tx;                 // [n_chunks + 1] first transcript whose record begins at or behind each chunk's first byte
    // the image
    uint64_t* seg;                      // [n_chunks * PATCH_SEG_CAP]
    uint32_t* patch;                    // [n_chunks * PATCH_PATCH_CAP]
    Chunk*    chunks;                   // [n_chunks] arena order
    uint8_t*  bucket; uint8_t* sub;     // XCD slice / window of every chunk (order_chunks_for_xcds' keys)
    uint64_t* hap_out_begin;
    uint64_t* totals;                   // [4]: segments, patches of the whole image, -, -
    unsigned long long* status;
};

hipError_t launch_patch_positions(const PatchBuildArgs& a, uint64_t* scan_scratch, hipStream_t stream);   // tx_arena_len, tx_res_base (scan), hap_out_begin needs out_bytes: see launch_patch_hap_begin
hipError_t launch_patch_hap_begin(const PatchBuildArgs& a, hipStream_t stream);
hipError_t launch_patch_build(const PatchBuildArgs& a, hipStream_t stream);

struct PatchExecArgs {
    const uint64_t* seg; const uint32_t* patch; const Chunk* chunks; uint32_t n_chunks;
    const uint8_t* src0; uint64_t src0_len; const uint8_t* src1; uint64_t src1_len;
    uint8_t* out; uint64_t out_len;
    unsigned long long* status;
};
hipError_t launch_stitch_patch(const PatchExecArgs& a, hipStream_t stream, bool nontemporal);
hipError_t preload_patch_image(hipStream_t stream);

}  // namespace v2p
