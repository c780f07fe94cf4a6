// decode_kernels.h -- argument block and host launcher of the BCSQ bitmask decode (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace v2p {

// reasons in the low byte of the decode status word
enum : uint32_t {
    DEC_MASK_NEGATIVE = 1,     // text_parser.rs:210,244
    DEC_MASK_PARSE = 2,        // MaskDecoder.rs:41,47
    DEC_MASK_INDEX = 3,        // vcf_ds.rs:321,324
    DEC_COLUMNS = 4,           // vcf_ds.rs:148
    DEC_FIELD_TOO_LONG = 5,
    DEC_CAPACITY = 6
};

constexpr uint32_t DEC_ROWBLOCK = 64;       // records per row block (count / emit granularity)
constexpr uint32_t DEC_RANGE_HAPS = 6144;   // haplotypes one count / emit workgroup keeps in LDS (48 KiB of 64-bit cursors)
constexpr uint32_t DEC_SCAN_GROUPS = 64;    // the prefix down the row blocks runs in this many independent groups
constexpr uint32_t DEC_TILE = 4096;         // text bytes per parse step of a 256-thread workgroup (16 per thread)
constexpr uint32_t DEC_EMIT_GROUP = 8;      // records whose first 128 carriers the emit kernel fetches ahead together
constexpr uint32_t DEC_STAGE_IDS = 12288;  // ids of one (record block, haplotype range) that the staged emit kernel keeps in LDS (48 KiB)
constexpr uint32_t DEC_MULTI = 0x80000000u; // carrier entry: bit 31 set -> low 31 bits index the multi-word list

// One carrier: a (record, sample) whose filtered mask is not empty.  x = sample, y = the first word filtered to the
// supported consequences, or DEC_MULTI | offset into the multi-word list.
struct DecCarrier { uint32_t sample, entry; };

struct DecodeArgs {
    const uint8_t*  text;          // 16 readable bytes either side
    uint64_t        n_text;
    const uint64_t* row_begin;     // [n_rows]
    const uint64_t* row_end;       // [n_rows]
    uint32_t        n_rows;
    uint32_t        n_samples;
    uint32_t        parse_threads;  // 64 / 128 / 256 threads per record in the parse kernel, 0 = pick from n_samples
    const uint32_t* csq_begin;     // [n_rows + 1]
    const uint32_t* sup_pairs;     // [n_rows]
    const uint32_t* sup_bits;      // bitset over consequence ids
    // workspace
    DecCarrier*     carriers;      // [n_rows][n_samples] slots; record r uses the first row_nnz[r] of its row, in no particular order
    uint32_t*       row_nnz;       // [n_rows]
    uint32_t*       cnt;           // [n_rowblocks][2*n_samples] per-block counts, then exclusive prefix down the blocks
    uint32_t*       group_tot;     // [DEC_SCAN_GROUPS][2*n_samples] scratch of the scan
    uint32_t*       blk_total;     // [n_rowblocks * n_ranges] ids of each (record block, haplotype range): which emit kernel takes it
    uint32_t*       ovf;           // multi-word list: {n_words, words...} records
    uint64_t        ovf_capacity;  // in u32 words
    unsigned long long* ovf_used;  // device counter
    // outputs
    uint64_t*       hap_begin;     // [2*n_samples + 1]
    uint32_t*       ids;
    uint64_t        ids_capacity;
    unsigned long long* status;    // [0] min((row*n_samples + sample) << 8 | reason), [1] multi-word words needed
};

struct DecodeLayout {
    uint64_t carriers_off, nnz_off, cnt_off, group_off, blk_off, ovf_off, ovf_used_off, total;
    uint32_t n_rowblocks;
};
DecodeLayout decode_layout(uint64_t n_rows, uint64_t n_samples, uint64_t ovf_words);

// phases: bit 0 parse, bit 1 count, bit 2 scan, bit 3 emit
hipError_t launch_decode(const DecodeArgs& a, hipStream_t stream, unsigned phases);

}  // namespace v2p
