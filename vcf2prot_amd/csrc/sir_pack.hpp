// sir_pack.hpp -- device image of a batch of haplotype Task vectors (host side).
//
// The reference hands the executor, per haplotype, a Vec<Task> plus three char
// tapes (GIR, gir.rs:15-23; SoA form gir.rs:283-299).  The MI355X engine
// executes many haplotypes per launch from ONE concatenated image:
//
//   desc[N]      8-byte packed task descriptors, in result order
//                  bits  0..39  source offset (bytes) inside its source space
//                  bits 40..61  length (bytes), < 4 Mi  (longer tasks are split)
//                  bits 62..63  source space: 0 = resident proteome (ref tape),
//                               1 = batch payload arena (alt bytes, private ref tapes),
//                               2 = fill with '.' (cells no task covers keep the
//                                   '.' of haplotype_instruction.rs:78)
//                               3 = immediate: the low 40 bits ARE the bytes (1..5 of them,
//                                   first byte lowest) -- alt payloads of missense /
//                                   deletion / short insertion tasks travel inside their
//                                   descriptor and need no gather at all
//                  space 3 with bit 61 set = a fused single-residue substitution ("SNV3"), the commonest
//                  Task triple of all (transcript_instructions.rs:654-663 between two reference copies):
//                    copy len1 residues of the proteome from src, write one literal byte, copy len2
//                    residues from src + len1 + 1
//                  bits 0..28 src, 29..40 len1, 41..52 len2, 53..60 the byte.  One descriptor instead of
//                  three: the descriptor stream of an SNV-only cohort drops from 6 % to 2 % of the result
//                  bytes (a streamed HBM read above 1/32 of a saturated write stream costs that stream 40 %
//                  on MI355X, profiles/r02_copy_mix_*.json).  The kernel expands it back into three tasks.
//                  space 3 with bits 61..60 = 01 (an immediate's length field can never reach bit 60) = TWO
//                  substitutions in a row ("SNV5", dense images only: deep Task vectors are copy / literal / copy /
//                  literal / copy with a handful of residues between the literals): bits 0..28 src, 29..33 len1,
//                  34..38 len2, 39..43 len3, 44..51 the first byte, 52..59 the second -- five tasks in one descriptor.
//   chunks[C]    work items of <= 256 (deep Task vectors: 1024) consecutive descriptors and < 64 KiB of
//                result: {first descriptor, result offset:48 | descriptor count:11 | 0:4 | long-run flag:1}.
//                The flag routes the chunk: set = stitch4_kernel (<= 512 tasks, may hold fused descriptors),
//                clear = the per-block stitch_kernel (<= 1024 descriptors, one task each).
//                Inside a chunk result offsets are the exclusive prefix sum of the
//                lengths (computed on the device by a wave64 scan), so the 8-byte
//                start_pos_res of every Task never crosses PCIe or HBM.
//   payload[]    alt tapes (1 byte per residue) of all haplotypes, back to back
//   hap_out_begin[H+1]  result range of each haplotype inside the arena
//
// A Task vector is *canonical* when its result ranges are ascending and
// non-overlapping; gaps are legal (transcript_instructions.rs 'P' instruction
// leaves the last cell '.', golden case test_correct_translation_20) and become
// fill descriptors.  Overlapping or descending vectors keep the reference's
// "later task wins" semantics only on the ordered path (v2p_execute_gir).
#pragma once
#include <cstdint>
#if defined(__HIPCC__)
#define V2P_HOST_DEVICE __host__ __device__     // (this header is also compiled by g++ for the host-only library)
#else
#define V2P_HOST_DEVICE
#endif
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>

namespace v2p {

constexpr unsigned SPACE_PROTEOME = 0;
constexpr unsigned SPACE_PAYLOAD  = 1;
constexpr unsigned SPACE_FILL     = 2;
constexpr unsigned SPACE_IMM      = 3;
constexpr uint32_t IMM_MAX_BYTES  = 5;         // literal bytes that fit the 40-bit source field
constexpr uint64_t SNV3_MARK      = (3ull << 62) | (1ull << 61);
constexpr uint32_t SNV3_MAX_LEN   = 4095;      // len1, len2 (12 bits each)
constexpr uint64_t SNV3_MAX_SRC   = (1ull << 29) - 1;
constexpr uint64_t SNV5_MARK      = (3ull << 62) | (1ull << 60);     // bits 63..60 = 1101
constexpr uint32_t SNV5_MAX_LEN   = 31;        // len1, len2, len3 (5 bits each)

constexpr uint64_t SRC_MASK   = (1ull << 40) - 1;
constexpr uint32_t LEN_BITS   = 22;
constexpr uint32_t LEN_MASK   = (1u << LEN_BITS) - 1;
constexpr uint32_t CHUNK_TASKS = 256;          // tasks per work item for long-run images (one per lane)
constexpr uint32_t CHUNK_TASKS_MID = 512;      // ... for images with 40..110 result bytes per task: 2 per lane, per-block kernel
constexpr uint32_t CHUNK_TASKS_DEEP = 1024;    // ... for dense images (a few result bytes per task): 4 per lane
constexpr uint32_t CHUNK_TASKS_WAVE = 64;      // DESCRIPTORS per work item of a wave image (stitchw_kernel: one wave per chunk, one descriptor per lane)
constexpr uint32_t CHUNK_BYTES = 64u * 1024u - 16u;  // most result bytes a work item may hold (<= 4096 16-byte blocks incl. a ragged head)
constexpr uint32_t CHUNK_BYTES_LONG = 32u * 1024u;   // ... of a long-run work item: 2048 blocks = eight 1 KiB rows per wave, all gathered before the first store
constexpr uint32_t CHUNK_BYTES_WAVE = 10240;   // ... of a wave image, ragged head included: 640 blocks = ten 1 KiB rows of ONE wave (62 VGPRs; twelve rows do not fit 64: hipcc spills the gathers' destination registers)
constexpr uint32_t CUT_ALIGN_WAVE = 1024;      // preferred cut of a wave image: whole 1 KiB rows
constexpr uint32_t CHUNK_BYTES_DENSE = 12272;  // ... of a dense image: the 12 KiB LDS image of stitch_dense_kernel takes the chunk in one window
constexpr uint32_t DENSE_BELOW = 40;           // a chunk with fewer result bytes per task than this switches the builder to dense chunks
constexpr uint32_t CUT_ALIGN  = 4096;          // preferred chunk cut: 4 KiB multiples = full 256-lane passes of 16-byte blocks
constexpr uint64_t DST_MASK   = (1ull << 48) - 1;
constexpr uint64_t CHUNK_LONG = 1ull << 63;    // chunk header flag: long-run chunk
constexpr uint64_t CHUNK_LONG2 = 1ull << 62;   // ... holding 257..512 tasks (two task records per lane)
constexpr uint64_t CHUNK_DENSE = 1ull << 61;   // chunk header flag: chunk of a dense image (short tasks, fused descriptors allowed): stitch_dense_kernel
constexpr uint64_t CHUNK_WAVE = 1ull << 60;    // chunk header flag: chunk of a wave image (<= 64 descriptors, <= 10 KiB, fused substitutions allowed): stitchw_kernel
constexpr uint32_t CHUNK_N_MASK = 0x7FF;       // descriptor count: bits 48..58 of dst_n
inline uint32_t chunk_n(uint64_t dst_n) { return uint32_t(dst_n >> 48) & CHUNK_N_MASK; }
constexpr uint32_t WAVE_BYTES_PER_TASK = 24;    // an image whose first chunk has at least this many result bytes per task is a wave image (stitchw_kernel); below: dense
constexpr uint32_t LONG_RUN_BYTES_PER_TASK = 120;   // an image whose first chunk has at least this many result bytes per task goes to stitch4_kernel
constexpr uint32_t PAD_BYTES  = 32;            // readable slack before AND after a source arena: the kernel loads whole 16-byte aligned blocks
                                               // around a task's bytes (up to 30 bytes before its first byte in a chunk's ragged head block, 31 after its last)

struct Chunk {
    uint64_t task_begin;   // index of the first descriptor (low 42 bits) | rows images: head skip (11 bits) | tail clip (11 bits)
    uint64_t dst_n;        // result offset (48 bits) | descriptor count (16 bits)
};
static_assert(sizeof(Chunk) == 16, "Chunk is 16 bytes");

// ---- ROWS images (round 4; rows_image.hpp, build_rows.hip): the descriptors are written FIRST, whole -- nothing is cut at a chunk
// boundary -- and the chunks are cut afterwards on 1 KiB rows of the arena.  A descriptor that lies across a cut belongs to both
// chunks: the later one skips the bytes of its first descriptor that lie before the cut (HEAD SKIP), the earlier one drops the
// bytes of its last descriptor that lie behind it (TAIL CLIP): 11 bits each in the top of task_begin, so a descriptor of a rows
// image produces at most 2047 bytes (longer runs are several descriptors; a fused substitution takes copies of <= 1023 residues).
// CHUNK_CLIP marks the chunks of a rows image (they start on a multiple of 1024).  Images of the host packer never set any of it.
constexpr uint32_t TB_IDX_BITS = 42;
constexpr uint64_t TB_IDX_MASK = (1ull << TB_IDX_BITS) - 1;
constexpr uint32_t TB_SKIP_BITS = 11;
constexpr uint64_t CHUNK_CLIP  = 1ull << 59;
constexpr uint32_t ROW_BYTES   = 1024;
constexpr uint32_t PIECE_MAX   = (1u << TB_SKIP_BITS) - 1u;   // longest descriptor of a rows image: 2047 bytes
constexpr uint32_t ROWS_FUSE_LEN = 1023;                      // longest copy inside a fused substitution of a rows image (1023 + 1 + 1023 = 2047)
V2P_HOST_DEVICE inline uint64_t chunk_first(uint64_t task_begin) { return task_begin & TB_IDX_MASK; }
V2P_HOST_DEVICE inline uint32_t chunk_head_skip(uint64_t task_begin) { return uint32_t(task_begin >> TB_IDX_BITS) & PIECE_MAX; }
V2P_HOST_DEVICE inline uint32_t chunk_tail_clip(uint64_t task_begin) { return uint32_t(task_begin >> (TB_IDX_BITS + TB_SKIP_BITS)) & PIECE_MAX; }
V2P_HOST_DEVICE inline uint64_t chunk_dst(uint64_t dst_n) { return dst_n & ((1ull << 48) - 1); }
// PADDED rows images (round 5: what v2p_batch_build_and_execute leaves behind for a wave image; never seen by a host -- a download
// hands out the dense form).  The descriptors stay where the parse wrote them: tile t (K consecutive transcripts) owns slots
// [ROWS_TILE_SLOTS t, ROWS_TILE_SLOTS (t + 1)) of the array, filled from its first slot -- no compaction pass.  task_begin's index is
// a slot of that array, and a chunk whose descriptors go on into the NEXT tile says how many lie in its first one in the low ten bits
// of dst_n (a rows chunk starts on a multiple of 1024, so they are free): 0 = all of them (every dense-addressed chunk), n1 = the
// first n1 at task_begin, the others from the next tile's first slot.
constexpr uint32_t ROWS_TILE_SLOTS = 256;
constexpr uint32_t PAD_BYTES_PER_TASK_MAX = 125;   // a stream with at most this many result bytes per Task builds a padded wave image (C3: 84, C4: 117; C2: 134 keeps the compaction)
constexpr uint64_t CHUNK_N1_MASK = 1023;
V2P_HOST_DEVICE inline uint64_t chunk_next_tile(uint64_t first) { return (first / ROWS_TILE_SLOTS + 1u) * ROWS_TILE_SLOTS; }
// slot of the chunk's k-th descriptor (n1 = dst_n & CHUNK_N1_MASK)
V2P_HOST_DEVICE inline uint64_t chunk_desc_slot(uint64_t first, uint32_t n1, uint32_t k) { return (n1 == 0u || k < n1) ? first + k : chunk_next_tile(first) + (k - n1); }

inline uint64_t pack_desc(uint64_t src, uint32_t len, unsigned space) {
    return (src & SRC_MASK) | (uint64_t(len & LEN_MASK) << 40) | (uint64_t(space & 3u) << 62);
}
inline bool     desc_is_snv3(uint64_t d) { return (d & SNV3_MARK) == SNV3_MARK; }
inline uint64_t pack_snv3(uint64_t src, uint32_t len1, uint32_t len2, uint8_t byte) {
    return SNV3_MARK | (uint64_t(byte) << 53) | (uint64_t(len2 & 0xFFFu) << 41) | (uint64_t(len1 & 0xFFFu) << 29) | (src & SNV3_MAX_SRC);
}
inline bool     desc_is_snv5(uint64_t d) { return (d >> 60) == 0xDull; }
inline uint64_t pack_snv5(uint64_t src, uint32_t len1, uint8_t b1, uint32_t len2, uint8_t b2, uint32_t len3) {
    return SNV5_MARK | (uint64_t(b2) << 52) | (uint64_t(b1) << 44) | (uint64_t(len3 & 31u) << 39) | (uint64_t(len2 & 31u) << 34) | (uint64_t(len1 & 31u) << 29) | (src & SNV3_MAX_SRC);
}
inline uint64_t desc_src(uint64_t d)   { return desc_is_snv3(d) || desc_is_snv5(d) ? (d & SNV3_MAX_SRC) : (d & SRC_MASK); }
// result bytes a descriptor produces
inline uint32_t desc_len(uint64_t d)   {
    if (desc_is_snv5(d)) return uint32_t((d >> 29) & 31u) + 1u + uint32_t((d >> 34) & 31u) + 1u + uint32_t((d >> 39) & 31u);
    return desc_is_snv3(d) ? uint32_t((d >> 29) & 0xFFFu) + 1u + uint32_t((d >> 41) & 0xFFFu) : uint32_t(d >> 40) & LEN_MASK;
}
// source space; a fused substitution reads the proteome
inline unsigned desc_space(uint64_t d) { return desc_is_snv3(d) || desc_is_snv5(d) ? SPACE_PROTEOME : unsigned(d >> 62); }

enum PackStatus : int {
    PACK_OK = 0,
    PACK_BAD_CODE = 1,        // exe_code not in {0,1}: haplotype_instruction.rs:154 panics
    PACK_RES_OOB = 2,         // start_pos_res + length > result length: task.rs:43/47 panics
    PACK_SRC_OOB = 3,         // start_pos + length > tape length: task.rs:43/47 panics
    PACK_NOT_CANONICAL = 4,   // result ranges overlap or go backwards (ordered path required)
    PACK_TOO_LARGE = 5        // offset does not fit the descriptor
};

// Appends haplotypes to the concatenated image and cuts it into chunks.
class ImageBuilder {
public:
    std::vector<uint64_t> desc;
    std::vector<Chunk>    chunks;
    std::vector<uint8_t>  payload;
    std::vector<uint64_t> hap_out_begin{0};
    uint64_t n_copy_bytes = 0;     // A: residues written by copy tasks (Sum task.length)
    uint64_t n_ref_tasks = 0;      // N: Task descriptors consumed, zero-length ones included
    uint32_t chunk_tasks = CHUNK_TASKS;   // tasks per chunk (a fused substitution counts as the tasks it expands to)
    bool adaptive_tasks = true;       // switch between long-run and dense chunks by the last chunk's bytes per task
    uint32_t chunk_bytes = CHUNK_BYTES_LONG;
    bool adaptive_bytes = true;       // chunk_bytes follows the mode (CHUNK_BYTES_LONG / CHUNK_BYTES); false once the caller sets it
    uint32_t cut_align = CUT_ALIGN;   // power of two >= 16
    uint32_t max_chunk_tasks = 0;     // most tasks of any chunk routed to the per-block kernel (selects its tasks per lane)
    uint32_t max_long_tasks = 0;      // most tasks of any long-run chunk (<= 256: one task record per lane in stitch4_kernel, else two)
    uint64_t n_long_chunks = 0;       // chunks routed to stitch4_kernel
    int kernel_choice = 0;            // 0: undecided (adaptive images decide at their first chunk); 1: every chunk of <= 512 tasks long-run; 2: per-block kernel only, no fusion;
                                      // 3: dense image (stitch_dense_kernel, fusion on); 4: wave image (stitchw_kernel, fusion on) -- set with set_kernel()
    uint64_t n_wave_chunks = 0;       // chunks flagged for stitchw_kernel
    uint64_t n_dense_chunks = 0;      // chunks flagged for stitch_dense_kernel
    uint32_t soft_window = 8;         // tasks before the hard limit at which a chunk starts looking for its cut
    bool inline_payload = true;       // payload tasks of <= IMM_MAX_BYTES bytes become immediate descriptors
    uint32_t grid_bytes = 0;          // != 0: GRID cutting -- chunk k holds exactly the result bytes [k*grid, (k+1)*grid) (a multiple of 4 KiB; tasks
                                      // that straddle a grid line are split, zero-length tasks and fusion are dropped).  Chunk membership is then a pure
                                      // function of result offsets, which is what the device-side image builder (build_kernels.hip) computes in parallel
    bool fuse_snv = true;             // reference copy + 1-byte literal + reference copy going on one residue later -> one descriptor
    bool fuse_double = true;          // dense images: two substitutions in a row -> one descriptor
    uint64_t n_fused = 0;             // fused substitutions in the image
    bool grid_overflow = false;       // grid cutting: some window holds more descriptors than its kernel takes (PACK_TOO_LARGE)
    bool line_cut = true;             // a wave chunk that cannot reach a row boundary any more still ends on a 128-byte line (false: A/B runs)

    // Empty the image for another build; the vectors keep their capacity, the settings return to their defaults.
    void reset() {
        desc.clear(); chunks.clear(); payload.clear(); hap_out_begin.assign(1, 0);
        n_copy_bytes = n_ref_tasks = n_fused = n_long_chunks = n_dense_chunks = n_wave_chunks = 0;
        chunk_tasks = CHUNK_TASKS; adaptive_tasks = true; chunk_bytes = CHUNK_BYTES_LONG; adaptive_bytes = true; cut_align = CUT_ALIGN;
        grid_overflow = false; max_chunk_tasks = max_long_tasks = 0; soft_window = 8; inline_payload = true; fuse_snv = true; fuse_double = true; kernel_choice = 0; grid_bytes = 0; line_cut = true;
        cursor_ = extra_ = arena_cursor_ = open_begin_ = open_dst_ = 0;
        open_n_ = open_bytes_ = open_desc_ = 0; open_fused_ = false; st_n_ = 0; st0_virtual_ = false;
    }
    // Fix the kernel of the whole image up front (before the first task): chunk limits follow it.
    void set_kernel(int k) {
        kernel_choice = k;
        if (k == 4 && !grid_bytes) { chunk_tasks = CHUNK_TASKS_WAVE; chunk_bytes = CHUNK_BYTES_WAVE; adaptive_tasks = false; adaptive_bytes = false; }
        // (a dense image's FIRST chunk too: the limits used to follow the kernel only from the second chunk on, and 256 fused descriptors of
        // 50+ bytes overran the kernel's 12 KiB window -- found by tools/routing_sweep.py)
        if (k == 3 && !grid_bytes) { if (adaptive_tasks) chunk_tasks = CHUNK_TASKS_DEEP; if (adaptive_bytes) chunk_bytes = CHUNK_BYTES_DENSE; }
    }
    // Build a PART of a larger arena (threads packing haplotype ranges side by side): the part's first result byte sits at absolute
    // arena offset `o`, so that chunk cuts are aligned in the arena the kernels write, not in the part.  Call before the first task;
    // chunk offsets and hap_out_begin are then absolute.
    void set_origin(uint64_t o) { arena_cursor_ = o; hap_out_begin.assign(1, o); }
    uint64_t out_size() const { return hap_out_begin.back() - hap_out_begin.front(); }
    uint64_t n_haplotypes() const { return hap_out_begin.size() - 1; }

    // Reserve `n` payload bytes for the haplotype being added; returns their arena offset.
    uint64_t payload_alloc(uint64_t n) { uint64_t o = payload.size(); payload.resize(o + n); return o; }

    // One task of the current haplotype, already translated to a source space/offset.
    // `dst` is relative to the haplotype's result tape; tasks must arrive in canonical order.
    int add_task(unsigned space, uint64_t src, uint64_t len, uint64_t dst, uint64_t n_res) {
        if (dst < cursor_) return PACK_NOT_CANONICAL;
        if (dst + len > n_res || dst + len < dst) return PACK_RES_OOB;
        if (dst > cursor_) { flush(); emit(SPACE_FILL, 0, dst - cursor_); }
        if (src + len > SRC_MASK) return PACK_TOO_LARGE;
        ++n_ref_tasks;
        n_copy_bytes += len;
        if (inline_payload && space == SPACE_PAYLOAD && len >= 1 && len <= IMM_MAX_BYTES && src + len <= payload.size()) {
            uint64_t lit = 0;
            for (uint64_t k = 0; k < len; ++k) lit |= uint64_t(payload[src + k]) << (8 * k);
            stage(SPACE_IMM, lit, len);
        } else {
            stage(space, src, len);
        }
        cursor_ = dst + len;
        return PACK_OK;
    }
    void end_haplotype(uint64_t n_res) {
        flush();
        if (cursor_ < n_res) emit(SPACE_FILL, 0, n_res - cursor_);
        hap_out_begin.push_back(hap_out_begin.back() + n_res + extra_);
        cursor_ = 0;
        extra_ = 0;
    }
    // FASTA emit (personalized_genome.rs:90-113 fused into the scatter): bytes that are not
    // part of the result tape -- record headers and line feeds -- are ordinary descriptors
    // placed between the tasks; they advance the arena but not the tape cursor.
    void add_literal(unsigned space, uint64_t src, uint32_t len) { flush(); emit(space, src, len); extra_ += len; }
    // '.'-fill the result tape up to `dst` (cells no task covers before a record ends)
    // (also the end of a transcript: tasks held back for fusion never pair up with the next transcript's)
    void fill_to(uint64_t dst) { flush(); if (dst > cursor_) { emit(SPACE_FILL, 0, dst - cursor_); cursor_ = dst; } }
    // Close the open chunk; call once after the last haplotype.
    void finish() { flush(); close_chunk(); }

    // Raw append of one canonical descriptor (tasks staged for fusion must have been flushed).
    void emit(unsigned space, uint64_t src, uint64_t len) {
        if (grid_bytes) { emit_grid(space, src, len); return; }
        if (len == 0) { push(space, src, 0); return; }
        const uint32_t most = kernel_choice == 4 ? chunk_bytes - 16u : chunk_bytes;      // (a piece must fit an empty chunk whatever its start)
        while (len) {
            uint32_t piece = uint32_t(len < most ? len : most);
            push(space, src, piece);
            src = advance(space, src, piece);
            len -= piece;
        }
    }

    // source field of the remainder of a task after its first `n` bytes went into another descriptor
    static uint64_t advance(unsigned space, uint64_t src, uint32_t n) {
        if (space == SPACE_FILL) return src;
        if (space == SPACE_IMM) return n >= 8 ? 0 : src >> (8 * n);
        return src + n;
    }

private:
    uint64_t cursor_ = 0;            // next uncovered cell of the current haplotype
    uint64_t extra_ = 0;             // literal bytes (FASTA headers, line feeds) added to the current haplotype
    uint64_t arena_cursor_ = 0;      // result offset of the next descriptor
    uint64_t open_begin_ = 0, open_dst_ = 0;
    uint32_t open_n_ = 0, open_bytes_ = 0;   // tasks / result bytes of the open chunk
    uint32_t open_desc_ = 0;                 // descriptors of the open chunk (<= open_n_)
    // what chunk_tasks limits: tasks, or -- in a dense image, whose kernel has one lane slot per DESCRIPTOR -- descriptors
    uint32_t open_units() const { return kernel_choice == 3 || kernel_choice == 4 ? open_desc_ : open_n_; }
    // ... and a dense image cuts at any multiple of 16 (its chunks are a few KiB: a 4 KiB preference would cost a third of them)
    // a wave image at whole 1 KiB rows (its chunks are at most ten of them)
    uint32_t cut_pref() const { return kernel_choice == 3 && adaptive_bytes ? 16u : (kernel_choice == 4 && cut_align > CUT_ALIGN_WAVE ? CUT_ALIGN_WAVE : cut_align); }
    bool open_fused_ = false;                // it holds a fused descriptor
    // most result bytes the open (or next) chunk may hold: a wave chunk is ten rows INCLUDING the ragged head of an unaligned start
    uint32_t byte_limit() const { return kernel_choice == 4 ? chunk_bytes - uint32_t((open_n_ ? open_dst_ : arena_cursor_) & 15u) : chunk_bytes; }
    // wave image: the LAST preferred boundary (whole 1 KiB rows) the open chunk can reach -- where it is cut unless it runs out of
    // descriptor slots first
    uint64_t wave_target() const {
        const uint64_t base = open_n_ ? open_dst_ : arena_cursor_, end = base + byte_limit();
        uint64_t t = end & ~uint64_t(cut_pref() - 1u);
        if (t <= base) t = end & ~15ull;         // (explicit tiny limits, tests: no whole row fits)
        return t <= base ? end : t;
    }
    struct Staged { unsigned space; uint64_t src; uint64_t len; };
    Staged st_[4];                   // tasks held back because the next one may complete a fused substitution (dense images: a second one)
    int st_n_ = 0;
    uint64_t run_src_ = 0;           // st_n_ >= 3: where the fused run of st_[0..2] starts in the proteome
    bool st0_virtual_ = false;       // st_[0] is not a task: a literal opened the run (dense images), the copy before it is empty

    bool long_run_mode() const { return chunk_tasks <= CHUNK_TASKS; }
    // GRID cutting: pieces never cross a multiple of grid_bytes; a chunk closes exactly on the grid
    void emit_grid(unsigned space, uint64_t src, uint64_t len) {
        while (len) {
            const uint64_t room = grid_bytes - (arena_cursor_ % grid_bytes);
            const uint32_t piece = uint32_t(len < room ? len : room);
            append(space, src, piece);
            src = advance(space, src, piece);
            len -= piece;
            if (arena_cursor_ % grid_bytes == 0) close_chunk();
        }
    }
    void flush() {
        const int n = st_n_;
        st_n_ = 0;
        if (n >= 3) {                // a complete substitution that waited for a second one
            emit_fused(run_src_, uint32_t(st_[0].len), uint8_t(st_[1].src), uint32_t(st_[2].len));
            if (n == 4) emit(st_[3].space, st_[3].src, st_[3].len);
            st0_virtual_ = false;
            return;
        }
        for (int i = 0; i < n; ++i) if (!(i == 0 && st0_virtual_)) emit(st_[i].space, st_[i].src, st_[i].len);
        st0_virtual_ = false;
    }
    // One task in canonical order.  [reference copy] [1-byte literal] [reference copy one residue further on] becomes one
    // descriptor when it fits the open chunk whole; everything else goes in as it is.
    void stage(unsigned space, uint64_t src, uint64_t len) {
        // fusion needs a long-run chunk: decided images only (the first chunk of an adaptive image goes in unfused)
        // (on a grid, fusion is a local rule: long-run routing chosen by the caller, the triple inside one window)
        // (a dense image fuses as well: its kernel takes the triple as one reference run with one byte patched)
        const bool may_fuse = fuse_snv && (grid_bytes ? (kernel_choice == 1 || kernel_choice == 3 || kernel_choice == 4)
                                                      : (kernel_choice == 3 || kernel_choice == 4 || (long_run_mode() && kernel_choice != 2 && !(adaptive_tasks && kernel_choice == 0))));
        if (!may_fuse) { flush(); emit(space, src, len); return; }
        if (st_n_ == 4) {            // [copy][byte][copy][byte] + the copy going on one residue behind the third: two substitutions, one descriptor
            const uint64_t want = run_src_ + st_[0].len + 1 + st_[2].len + 1;
            if (space == SPACE_PROTEOME && len <= SNV5_MAX_LEN && (len == 0 || src == want) && want + len <= SNV3_MAX_SRC) {
                st_n_ = 0; st0_virtual_ = false;
                emit_fused2(run_src_, uint32_t(st_[0].len), uint8_t(st_[1].src), uint32_t(st_[2].len), uint8_t(st_[3].src), uint32_t(len));
                return;
            }
            flush();                 // the first substitution as it is, the literal on its own
        }
        if (st_n_ == 3) {
            if (space == SPACE_IMM && len == 1) { st_[3] = Staged{space, src, len}; st_n_ = 4; return; }
            flush();
        }
        if (st_n_ == 2) {
            // the copy after the literal goes on one residue behind the copy before it; an EMPTY copy (substitution at the first
            // or last residue: transcript_instructions.rs:734, :648-649) has no source to speak of and fits any neighbour
            const bool fits = st_[0].len == 0 ? (len > 0 && src >= 1 && src - 1 + 1 + len <= SNV3_MAX_SRC) : (len == 0 || src == st_[0].src + st_[0].len + 1);
            if (space == SPACE_PROTEOME && len <= SNV3_MAX_LEN && fits) {
                const Staged a = st_[0], b = st_[1];
                const uint64_t run = a.len == 0 ? src - 1 : a.src;
                if (kernel_choice == 3 && fuse_double && a.len <= SNV5_MAX_LEN && len <= SNV5_MAX_LEN) {      // a dense image waits for a second substitution
                    st_[2] = Staged{space, src, len}; run_src_ = run; st_n_ = 3;
                    return;
                }
                st_n_ = 0; st0_virtual_ = false;
                emit_fused(run, uint32_t(a.len), uint8_t(b.src), uint32_t(len));
                return;
            }
            flush();
        }
        if (st_n_ == 1) {
            if (space == SPACE_IMM && len == 1) { st_[1] = Staged{space, src, len}; st_n_ = 2; return; }
            flush();
        }
        if (space == SPACE_PROTEOME && len <= SNV3_MAX_LEN && src + len + 1 + SNV3_MAX_LEN <= SNV3_MAX_SRC) { st_[0] = Staged{space, src, len}; st_n_ = 1; st0_virtual_ = false; return; }
        // a dense image: a lone literal may open a fused run -- [byte][copy] and [byte][copy][byte][copy] are the fused forms with an empty
        // first copy (in a chain of substitutions every descriptor then carries two of them)
        // (a wave image as well: a second substitution in a transcript is then one descriptor, not two, and more of its chunks reach
        // their ten rows before their 64 descriptor slots run out)
        if (((kernel_choice == 3 && fuse_double) || kernel_choice == 4) && space == SPACE_IMM && len == 1) {
            st_[0] = Staged{SPACE_PROTEOME, 0, 0}; st_[1] = Staged{space, src, len}; st_n_ = 2; st0_virtual_ = true;
            return;
        }
        emit(space, src, len);
    }
    void emit_fused2(uint64_t src, uint32_t len1, uint8_t b1, uint32_t len2, uint8_t b2, uint32_t len3) {
        const uint32_t total = len1 + 1u + len2 + 1u + len3, cnt = (len1 ? 1u : 0u) + 1u + (len2 ? 1u : 0u) + 1u + (len3 ? 1u : 0u);
        bool whole;
        if (grid_bytes) whole = arena_cursor_ / grid_bytes == (arena_cursor_ + total - 1) / grid_bytes;
        else {
            const uint32_t soft_tasks = chunk_tasks > soft_window ? chunk_tasks - soft_window : chunk_tasks;
            const uint32_t soft_bytes = byte_limit() > cut_pref() ? byte_limit() - cut_pref() : byte_limit();
            whole = open_units() + 1u <= soft_tasks && open_bytes_ + total <= soft_bytes;
        }
        if (whole) {
            if (open_n_ == 0) { open_begin_ = desc.size(); open_dst_ = arena_cursor_; }
            desc.push_back(pack_snv5(src, len1, b1, len2, b2, len3));
            ++open_desc_; open_n_ += cnt; open_bytes_ += total; arena_cursor_ += total;
            n_fused += 2; open_fused_ = true;
            if (grid_bytes && arena_cursor_ % grid_bytes == 0) close_chunk();
            return;
        }
        // across a cut: the first substitution on its own (it may still fit), then the rest task by task
        emit_fused(src, len1, b1, len2);
        emit(SPACE_IMM, b2, 1);
        emit(SPACE_PROTEOME, src + len1 + 1 + len2 + 1, len3);
    }
    void emit_fused(uint64_t src, uint32_t len1, uint8_t byte, uint32_t len2) {
        const uint32_t total = len1 + 1u + len2, cnt = (len1 ? 1u : 0u) + 1u + (len2 ? 1u : 0u);
        if (grid_bytes) {
            if (arena_cursor_ / grid_bytes == (arena_cursor_ + total - 1) / grid_bytes) {
                if (open_n_ == 0) { open_begin_ = desc.size(); open_dst_ = arena_cursor_; }
                desc.push_back(pack_snv3(src, len1, len2, byte));
                ++open_desc_; open_n_ += cnt; open_bytes_ += total; arena_cursor_ += total;
                ++n_fused; open_fused_ = true;
                if (arena_cursor_ % grid_bytes == 0) close_chunk();
            } else {
                emit(SPACE_PROTEOME, src, len1);
                emit(SPACE_IMM, byte, 1);
                emit(SPACE_PROTEOME, src + len1 + 1, len2);
            }
            return;
        }
        const uint32_t soft_tasks = chunk_tasks > soft_window ? chunk_tasks - soft_window : chunk_tasks;
        if (kernel_choice == 4) {
            if (open_n_ && arena_cursor_ == wave_target()) close_chunk();      // (the run opens the next chunk fused)
            if (open_units() + 1u <= soft_tasks && arena_cursor_ + total <= wave_target()) {
                if (open_n_ == 0) { open_begin_ = desc.size(); open_dst_ = arena_cursor_; }
                desc.push_back(pack_snv3(src, len1, len2, byte));
                ++open_desc_; open_n_ += cnt; open_bytes_ += total; arena_cursor_ += total;
                ++n_fused; open_fused_ = true;
                return;
            }
            emit(SPACE_PROTEOME, src, len1);
            emit(SPACE_IMM, byte, 1);
            emit(SPACE_PROTEOME, src + len1 + 1, len2);
            return;
        }
        const uint32_t soft_bytes = byte_limit() > cut_pref() ? byte_limit() - cut_pref() : byte_limit();
        if (open_units() + (kernel_choice == 3 || kernel_choice == 4 ? 1u : cnt) <= soft_tasks && open_bytes_ + total <= soft_bytes) {
            if (open_n_ == 0) { open_begin_ = desc.size(); open_dst_ = arena_cursor_; }
            desc.push_back(pack_snv3(src, len1, len2, byte));
            ++open_desc_; open_n_ += cnt; open_bytes_ += total; arena_cursor_ += total;
            ++n_fused; open_fused_ = true;
            return;
        }
        // close to a cut: the three tasks go in one by one (the cut may split one of them)
        emit(SPACE_PROTEOME, src, len1);
        emit(SPACE_IMM, byte, 1);
        emit(SPACE_PROTEOME, src + len1 + 1, len2);
    }

    void close_chunk() {
        if (open_n_ == 0) return;
        // Which kernel takes the image is decided ONCE, when its first chunk closes (mixing costs: every workgroup of the other
        // kind still launches).  Measured on MI355X (tools/ab.py, profiles/r02_*): stitch4_kernel on chunks of <= 256 tasks / 32 KiB
        // wins for long runs that fuse (C2: -20 %); the per-block kernel wins on 256-task / 32 KiB chunks for C4 (112 bytes per
        // task, a proteome larger than L2), on 512-task / 32 KiB chunks for C3 (90), on 1024-task / 64 KiB chunks for dense images
        // (C5: 7).  An explicit chunk_tasks (adaptive_tasks off) keeps the choice per chunk: <= 256 tasks -> long-run.
        const uint32_t bpt = open_bytes_ / open_n_;
        if (grid_bytes) {                              // the caller chose the kernel (1: long-run, else per block); nothing adapts
            // a window with more descriptors (tasks) than its kernel takes: the image is refused (the device builder reports the same
            // window with STATUS_TOO_MANY) -- pick a smaller window
            if ((kernel_choice == 4 && open_desc_ > CHUNK_TASKS_WAVE) || open_desc_ > CHUNK_TASKS_DEEP || (kernel_choice == 1 && open_n_ > 2u * CHUNK_TASKS)) grid_overflow = true;
            const bool lg = kernel_choice == 1 && open_n_ <= 2u * CHUNK_TASKS, dn = kernel_choice == 3, wv = kernel_choice == 4;
            chunks.push_back(Chunk{open_begin_, (open_dst_ & DST_MASK) | (uint64_t(open_desc_) << 48) | (lg ? CHUNK_LONG : 0ull) | (lg && open_n_ > CHUNK_TASKS ? CHUNK_LONG2 : 0ull)
                                                | (dn ? CHUNK_DENSE : 0ull) | (wv ? CHUNK_WAVE : 0ull)});
            if (lg) { ++n_long_chunks; if (open_n_ > max_long_tasks) max_long_tasks = open_n_; } else if (dn) ++n_dense_chunks; else if (wv) ++n_wave_chunks; else if (open_n_ > max_chunk_tasks) max_chunk_tasks = open_n_;
            open_n_ = 0; open_bytes_ = 0; open_desc_ = 0; open_fused_ = false;
            return;
        }
        if (adaptive_tasks && kernel_choice == 0) {
            // stitchw_kernel, one wave per chunk, from 40 result bytes per task up: with its image read ahead of it phase by phase
            // (launch_stitch) it beats stitch4_kernel on long runs (C2: 3.7 -> 2.7 ms) and the per-block kernel on C3 (2.09 -> 1.76 ms),
            // C4 (1.13 -> 1.05) and 200-residue transcripts (2.19 -> 1.28); short tasks: the dense kernel (a tie at ~26 bytes per task)
            const int choice = bpt >= WAVE_BYTES_PER_TASK ? 4 : 3;
            if (choice == 4) {
                // the first chunk was filled under the undecided limits (<= 256 tasks, 32 KiB, nothing fused): its descriptors go
                // back and are cut again as wave chunks
                const std::vector<uint64_t> held(desc.begin() + int64_t(open_begin_), desc.end());
                desc.resize(open_begin_);
                arena_cursor_ = open_dst_;
                open_n_ = 0; open_bytes_ = 0; open_desc_ = 0; open_fused_ = false;
                set_kernel(4);
                for (uint64_t d : held) push(desc_space(d), desc_src(d), desc_len(d));
                close_chunk();
                return;
            }
            kernel_choice = choice;
        }
        const bool to_dense = kernel_choice == 3, to_wave = kernel_choice == 4;
        const bool to_long = !to_dense && !to_wave && (open_fused_ || (open_n_ <= 2u * CHUNK_TASKS && (kernel_choice == 1 || (kernel_choice == 0 && long_run_mode()))));
        chunks.push_back(Chunk{open_begin_, (open_dst_ & DST_MASK) | (uint64_t(open_desc_) << 48) | (to_long ? CHUNK_LONG : 0ull)
                                            | (to_long && open_n_ > CHUNK_TASKS ? CHUNK_LONG2 : 0ull) | (to_dense ? CHUNK_DENSE : 0ull) | (to_wave ? CHUNK_WAVE : 0ull)});
        if (to_long) { ++n_long_chunks; if (open_n_ > max_long_tasks) max_long_tasks = open_n_; }
        else if (to_dense) ++n_dense_chunks;
        else if (to_wave) ++n_wave_chunks;
        else if (open_n_ > max_chunk_tasks) max_chunk_tasks = open_n_;
        if (adaptive_tasks) {
            chunk_tasks = kernel_choice == 1 ? CHUNK_TASKS : (kernel_choice == 3 ? CHUNK_TASKS_DEEP
                          : (bpt >= LONG_RUN_BYTES_PER_TASK ? CHUNK_TASKS : (bpt >= DENSE_BELOW ? CHUNK_TASKS_MID : CHUNK_TASKS_DEEP)));
            // (512-task chunks of the per-block kernel: 64 KiB beats 32 KiB by 6 % on C3 and 10 % on C4 -- they close on the task limit at ~40 KiB)
            if (adaptive_bytes) chunk_bytes = kernel_choice == 3 ? CHUNK_BYTES_DENSE : (chunk_tasks <= CHUNK_TASKS ? CHUNK_BYTES_LONG : CHUNK_BYTES);
        }
        open_n_ = 0; open_bytes_ = 0; open_desc_ = 0; open_fused_ = false;
    }
    void append(unsigned space, uint64_t src, uint32_t len) {
        if (open_n_ == 0) { open_begin_ = desc.size(); open_dst_ = arena_cursor_; }
        desc.push_back(pack_desc(src, len, space));
        ++open_n_; ++open_desc_; open_bytes_ += len; arena_cursor_ += len;
    }
    // Chunks are cut at aligned result offsets whenever possible (the task that straddles the
    // cut is split into two descriptors): with a 16-byte multiple a workgroup's first and last
    // result blocks are whole (no byte-granular edge stores); with a 4 KiB multiple the chunk is
    // a whole number of 256-lane passes (measured: -5 % kernel time on C2).
    // A chunk enters "closing mode" 8 descriptors / 16 bytes before its hard limits; if the
    // cut is still unaligned at the hard limit it is made anyway (the kernel handles ragged edges).
    void push(unsigned space, uint64_t src, uint32_t len) {
        const uint32_t soft_tasks = chunk_tasks > soft_window ? chunk_tasks - soft_window : chunk_tasks;
        for (;;) {
            const uint32_t chunk_bytes = byte_limit();                         // (shadows the member: the limit of the chunk this piece goes to)
            if (kernel_choice == 4 && open_units() < soft_tasks) {
                // a wave chunk with descriptor slots to spare runs up to the last whole-row boundary it can reach; the piece that
                // straddles it is split there
                const uint64_t target = wave_target();
                if (open_n_ && arena_cursor_ == target) { close_chunk(); continue; }
                if (arena_cursor_ + len <= target) { append(space, src, len); return; }
                const uint32_t r = uint32_t(target - arena_cursor_);
                append(space, src, r);
                src = advance(space, src, r);
                len -= r;
                close_chunk();
                continue;
            }
            const uint32_t soft_bytes = chunk_bytes > cut_pref() ? chunk_bytes - cut_pref() : chunk_bytes;
            const bool closing = open_units() >= soft_tasks || open_bytes_ + len > soft_bytes;
            if (!closing) { append(space, src, len); return; }
            if (open_units() == chunk_tasks) { close_chunk(); continue; }     // ragged cut (many tiny tasks)
            // preferred cut: a multiple of cut_align (4 KiB = one full 256-lane pass of 16-byte blocks, so no
            // partially filled pass); when that cannot be reached any more, a multiple of 16
            uint32_t align = cut_pref();
            if (open_units() + 6 >= chunk_tasks || open_bytes_ + (align - uint32_t(arena_cursor_ & uint64_t(align - 1))) > chunk_bytes) {
                // a wave chunk that cannot reach a row boundary any more still ends on a 128-byte line while two slots are left: its rows
                // then cover whole lines, and no line of the arena is written in two halves by two waves
                const bool line = line_cut && kernel_choice == 4 && cut_pref() >= 128u && open_units() + 2 < chunk_tasks &&
                                  open_bytes_ + (128u - uint32_t(arena_cursor_ & 127ull)) <= chunk_bytes;
                align = line ? 128u : 16u;
            }
            const uint32_t misal = uint32_t(arena_cursor_ & uint64_t(align - 1));
            if (misal == 0 && open_n_ > 0) { close_chunk(); continue; }       // aligned cut
            const uint32_t r = align - misal;
            if (len <= r && open_bytes_ + len <= chunk_bytes) { append(space, src, len); return; }
            if (misal != 0 && open_bytes_ + r <= chunk_bytes) {               // split the straddling task at the boundary
                append(space, src, r);
                src = advance(space, src, r);
                len -= r;
                close_chunk();
                continue;
            }
            if (open_n_ == 0) { append(space, src, len); return; }            // a single piece always fits an empty chunk
            close_chunk();
        }
    }
};

// Launch bits of v2p_stitch_launch / launch_stitch for a chunk table: which kernels have work and how many tasks per lane they
// need (bit 1: chunks flagged for stitch_dense_kernel, bit 2: chunks flagged for stitchw_kernel, bit 3: a rows image (CHUNK_CLIP), bit 4: no long-run chunk, bit 5: no per-block chunk, bits 6..7: tasks per
// lane of stitch4_kernel, bits 8..11: of stitch_kernel).
inline int stitch_launch_bits(const Chunk* chunks, uint64_t n_chunks)
{
    uint32_t max_pb = 0;
    bool any_long = false, any_long2 = false, any_pb = false, any_dense = false, any_wave = false, any_clip = false;
    for (uint64_t i = 0; i < n_chunks; ++i) {
        const uint64_t dn = chunks[i].dst_n;
        any_clip = any_clip || (dn & CHUNK_CLIP) != 0;
        if (dn & CHUNK_LONG) { any_long = true; any_long2 = any_long2 || (dn & CHUNK_LONG2) != 0; }
        else if (dn & CHUNK_DENSE) any_dense = true;
        else if (dn & CHUNK_WAVE) any_wave = true;
        else { any_pb = true; const uint32_t n = chunk_n(dn); if (n > max_pb) max_pb = n; }
    }
    const int tpt = max_pb <= 256u ? 1 : (max_pb <= 512u ? 2 : 4);
    return (any_dense ? 2 : 0) | (any_wave ? 4 : 0) | (any_clip ? 8 : 0) | (any_long ? 0 : 16) | (any_pb ? 0 : 32) | ((any_long2 ? 2 : 1) << 6) | (tpt << 8);
}

constexpr uint32_t XCD_SUB = 256;                  // windows per proteome slice in the launch order
// window of reference position `key` inside slice `slice` (slices are `per` bytes): host and device builders must agree
V2P_HOST_DEVICE inline uint8_t xcd_sub_window(uint64_t key, uint32_t slice, uint64_t per)
{
    const uint64_t w = (per + XCD_SUB - 1) / XCD_SUB, in = key - uint64_t(slice) * per;
    const uint64_t q = w ? in / w : 0;
    return uint8_t(q < XCD_SUB ? q : XCD_SUB - 1);
}

// XCD-aware launch order.  Workgroups are dealt round-robin to the 8 XCDs (workgroup b runs on
// XCD b % 8, observed dispatch behaviour), each XCD has its own 4 MiB L2, and every haplotype
// re-reads the same proteome.  Reordering the chunk table so that entry 8*j + x is the j-th
// chunk whose reference reads fall into proteome slice x keeps 1/8 of the proteome hot in
// each L2 (measured on C2: HBM fetch 7.2 GB -> 0.6 GB per pass).  Placement only changes
// speed, never results: chunks are independent.
inline void order_chunks_for_xcds_range(Chunk* chunks, uint64_t n_chunks, const uint64_t* desc, uint64_t n_desc,
                                        uint64_t proteome_len, unsigned n_xcd = 8, bool window_major = true)
{
    if (n_chunks < 2 * n_xcd || proteome_len == 0 || n_desc == 0) return;
    std::vector<uint32_t> bucket(n_chunks);
    std::vector<uint8_t> sub(n_chunks);
    std::vector<uint64_t> count(n_xcd, 0);
    for (uint64_t c = 0; c < n_chunks; ++c) {
        const uint64_t tb = chunk_first(chunks[c].task_begin);
        const uint32_t n = chunk_n(chunks[c].dst_n);
        uint64_t key = 0;
        for (uint32_t k = 0; k < n && k < 6 && tb + k < n_desc; ++k)       // skip FASTA literals stored behind the proteome
            if (desc_space(desc[tb + k]) == SPACE_PROTEOME && desc_src(desc[tb + k]) < proteome_len) { key = desc_src(desc[tb + k]); break; }
        const uint64_t per = (proteome_len + n_xcd - 1) / n_xcd;
        uint64_t b = key / per;
        bucket[c] = uint32_t(b < n_xcd ? b : n_xcd - 1);
        sub[c] = window_major ? xcd_sub_window(key, bucket[c], per) : 0;
        ++count[bucket[c]];
    }
    // rank inside the bucket, then interleave: sort key = (rank, bucket)
    std::vector<uint64_t> start(n_xcd, 0), seen(n_xcd, 0);
    std::vector<Chunk> out(n_chunks);
    // position of (rank r, bucket x) = number of chunks with rank < r over all buckets + buckets < x holding rank r
    // computed by a counting pass over ranks: ranks are dense per bucket, so iterate rank-major
    // inside a slice: one window of the proteome after the other (XCD_SUB per slice), every haplotype's chunk of a window
    // together -- the workgroups an XCD runs at any time then read the SAME few tens of KiB of reference (C2: -5 % at 1 000
    // samples, -10 % at 250).  Stable counting sort by window, then the stable deal by slice below.
    std::vector<uint64_t> by_sub(n_chunks);
    {
        std::vector<uint64_t> start(XCD_SUB + 1, 0);
        for (uint64_t c = 0; c < n_chunks; ++c) ++start[sub[c] + 1u];
        for (uint32_t q = 0; q < XCD_SUB; ++q) start[q + 1] += start[q];
        for (uint64_t c = 0; c < n_chunks; ++c) by_sub[start[sub[c]]++] = c;
    }
    std::vector<std::vector<uint64_t>> idx(n_xcd);
    for (unsigned x = 0; x < n_xcd; ++x) idx[x].reserve(count[x]);
    for (uint64_t q = 0; q < n_chunks; ++q) idx[bucket[by_sub[q]]].push_back(by_sub[q]);
    uint64_t pos = 0, max_count = 0;
    for (unsigned x = 0; x < n_xcd; ++x) max_count = count[x] > max_count ? count[x] : max_count;
    for (uint64_t r = 0; r < max_count; ++r)
        for (unsigned x = 0; x < n_xcd; ++x)
            if (r < count[x]) out[pos++] = chunks[idx[x][r]];
    for (uint64_t c = 0; c < n_chunks; ++c) chunks[c] = out[c];
    (void)start; (void)seen;
}

// The order above, applied inside BLOCKS of the arena: block after block (haplotypes [h0, h1) whose results fill about
// `block_bytes`), and inside a block slice by slice, window by window.  One order over a whole cohort keeps the reference reads of
// the workgroups in flight together but scatters their STORES over the whole arena (C3 whole, 36 GB: every chunk of a phase in
// another 2 MB page); inside blocks of about eight times the proteome the reads still share their windows and the stores stay
// within a few hundred MB: C3 whole 8.86 -> 7.6-7.8 ms, a 2 000-sample slice 1.74 -> 1.56, C4 whole 7.0 -> 6.3, C2 unchanged
// (profiles/r03_ab_block_order.txt).  Blocks smaller than that lose the reuse of the reference (C4, 56 MB proteome: 8.5 ms with
// 40 MB blocks).  Blocks are equal shares of the table's ENTRIES, on multiples of 8 (entry 8j + x of a block = the j-th chunk of
// slice x, on XCD x).  max_blocks <= 1: one order for the whole table.
// An image is "rich" when its descriptors are more than 3 % of the result they describe (C2: 2.2 %, C4: 3.7 %, C3: 5 %): the launcher
// gives rich images smaller phases and plain stores, the chunk order gives thin ones of 2 GB and more ONE order for the whole table
// (profiles/r04_routing_sweep.json)
constexpr double IMAGE_RICH_SHARE = 0.03;
inline bool image_is_rich(uint64_t n_desc, uint64_t result_bytes) { return 8.0 * double(n_desc) > IMAGE_RICH_SHARE * double(result_bytes); }
constexpr uint64_t THIN_ONE_ORDER_FROM = 2ull << 30;
constexpr uint64_t PHASE_BYTES_DEFAULT = 64ull << 20;    // bytes of image (chunk records + descriptors) per launch phase
constexpr uint64_t PHASE_BYTES_RICH = 28ull << 20;       // ... of a rich image
constexpr uint64_t PHASE_BYTES_STAGED = 44ull << 20;  // ... of a rich image whose descriptors are STAGED (stitch_kernels.h) while the reference is at most
constexpr uint64_t PHASE_STAGED_SMALL_REF = 8ull << 20;   // this long (1 MB per XCD)
constexpr uint32_t PHASE_MIN_CHUNKS = 16384;            // below this a launch is one phase and is not preceded by a read-ahead
inline uint64_t xcd_order_block_bytes(uint64_t proteome_len) { const uint64_t b = 8u * proteome_len; return b < (32ull << 20) ? (32ull << 20) : b; }
constexpr uint32_t XCD_ORDER_MAX_BLOCKS = 4096;             // blocks of one table (both builders)
// number of blocks for a table whose results span `span_bytes`, and the table entries [first, last) of block k: equal shares of
// the ENTRIES (in arena order), on multiples of 8 -- the one rule both builders use
inline uint32_t xcd_order_blocks(uint64_t span_bytes, uint64_t proteome_len, uint64_t n_entries, uint32_t max_blocks, uint64_t n_desc)
{
    // (an image whose descriptors are a thin stream -- C2: 2.2 % of its result -- gains nothing from blocks and loses 5-7 % with
    // a few dozen of them: one order for the whole table, as in round 2.  Below 2 GB the sweep of round 4 says the opposite --
    // 1.5 GB images of 800- and 1600-residue transcripts: 0.23-0.24 ms in blocks against 0.26-0.29 -- so small images keep the blocks)
    if (!image_is_rich(n_desc, span_bytes) && span_bytes >= THIN_ONE_ORDER_FROM) return 1;
    const uint64_t bb = xcd_order_block_bytes(proteome_len);
    uint64_t nb = (span_bytes + bb - 1) / bb;
    if (nb > max_blocks) nb = max_blocks;
    if (nb * 64u > n_entries) nb = n_entries / 64u;          // (a block is at least 56 entries: 8 per XCD to deal)
    return uint32_t(nb < 1 ? 1 : nb);
}
// Inside an XCD's proteome slice: window by window (every haplotype's chunk of window 0, then of window 1 ...: the workgroups in flight
// share their reference window) -- except for a THIN image of 2 GB and more (C2: one order for the whole table), which goes HAPLOTYPE
// after haplotype (round 6).  Window-major, an XCD sweeps the whole arena once per window, every chunk of a sweep a haplotype (8 MB) away
// from the last; that runs at 2.45 ms on some arenas and 3.1 on others -- where the allocation landed, nothing else
// (profiles/r06_arena_placement.txt) -- while haplotype-major, whose consecutive chunks are neighbours in the arena, runs at 2.70 on every
// arena (slow ones were four of five in the run that compared them).  Rich images keep the windows (C3 whole 7.68 against 7.87 ms, C4 whole
// 6.30 against 7.91): they are dealt inside 64 MB blocks, where a sweep is short anyway and the reference window is what matters.
inline bool xcd_order_window_major(uint64_t span_bytes, uint64_t n_desc) { return image_is_rich(n_desc, span_bytes) || span_bytes < THIN_ONE_ORDER_FROM; }
V2P_HOST_DEVICE inline uint64_t xcd_order_block_first(uint64_t n_entries, uint32_t n_blocks, uint32_t k)
{
    return k >= n_blocks ? n_entries : ((n_entries * k / n_blocks) & ~uint64_t(7));
}
inline void order_chunks_for_xcds(Chunk* chunks, uint64_t n_chunks, const uint64_t* desc, uint64_t n_desc,
                                  uint64_t proteome_len, unsigned n_xcd = 8, bool window_major = true, uint32_t max_blocks = XCD_ORDER_MAX_BLOCKS)
{
    if (max_blocks <= 1 || n_chunks < 2 * n_xcd || proteome_len == 0) { order_chunks_for_xcds_range(chunks, n_chunks, desc, n_desc, proteome_len, n_xcd, window_major); return; }
    // the table in arena order first (the packers emit it so; a caller's table may not be)
    bool sorted = true;
    for (uint64_t c = 1; c < n_chunks && sorted; ++c) sorted = chunk_dst(chunks[c - 1].dst_n) <= chunk_dst(chunks[c].dst_n);
    if (!sorted) std::stable_sort(chunks, chunks + n_chunks, [](const Chunk& a, const Chunk& b) { return chunk_dst(a.dst_n) < chunk_dst(b.dst_n); });
    const uint32_t nb = xcd_order_blocks(chunk_dst(chunks[n_chunks - 1].dst_n), proteome_len, n_chunks, max_blocks, n_desc);
    window_major = window_major && xcd_order_window_major(chunk_dst(chunks[n_chunks - 1].dst_n), n_desc);
    for (uint32_t k = 0; k < nb; ++k) {
        const uint64_t c0 = xcd_order_block_first(n_chunks, nb, k), c1 = xcd_order_block_first(n_chunks, nb, k + 1);
        order_chunks_for_xcds_range(chunks + c0, c1 - c0, desc, n_desc, proteome_len, n_xcd, window_major);
    }
}

// Interleaves one haplotype's tasks with FASTA record literals.  Records tile the result tape
// in ascending order (annotation of haplotype_instruction.rs:120-125); record i covers
// [rec_res_end[i-1], rec_res_end[i]) and is written as header bytes, its tasks, '\n'.
// emit_task(i) must call ImageBuilder::add_task for task i.  Returns PACK_OK or PACK_RES_OOB
// when a task straddles a record boundary.
template <class EmitTask>
int interleave_fasta(ImageBuilder& img, const uint64_t* start_pos_res, const uint64_t* length, uint64_t n_tasks,
                     const uint64_t* rec_res_end, const uint64_t* rec_header_src, const uint32_t* rec_header_len,
                     uint64_t n_rec, unsigned header_space, bool lf_precedes_headers, EmitTask&& emit_task)
{
    // lf_precedes_headers: the byte in front of every header in its table is '\n', so the line feed
    // that closes record r-1 and the header of record r are ONE literal (one descriptor less per record)
    uint64_t i = 0;
    for (uint64_t r = 0; r < n_rec; ++r) {
        const uint64_t end = rec_res_end[r];
        if (r == 0 || !lf_precedes_headers) img.add_literal(header_space, rec_header_src[r], rec_header_len[r]);
        else img.add_literal(header_space, rec_header_src[r] - 1, rec_header_len[r] + 1);
        while (i < n_tasks && (start_pos_res[i] < end || (length[i] == 0 && start_pos_res[i] == end && r + 1 == n_rec))) {
            if (start_pos_res[i] + length[i] > end) return PACK_RES_OOB;
            const int rc = emit_task(i);
            if (rc != PACK_OK) return rc;
            ++i;
        }
        img.fill_to(end);
        if (r + 1 == n_rec || !lf_precedes_headers)
            img.add_literal(header_space, rec_header_src[r] + rec_header_len[r] - 1, 1);   // the header's own line feed
    }
    return i == n_tasks ? PACK_OK : PACK_RES_OOB;
}

// Maps an offset inside a haplotype's private ref_stream (the concatenation of its
// mutated transcripts' references, haplotype_instruction.rs:118,130) to the resident
// proteome.  seg_ref_begin is ascending with seg_ref_begin[0] == 0.
struct RefSegments {
    const uint64_t* seg_ref_begin;   // [n_seg + 1], last = ref_stream length
    const uint64_t* seg_proteome_off;// [n_seg]
    uint64_t n_seg;
    mutable uint64_t hint = 0;
    // returns false when [pos, pos+len) is not inside one segment
    bool map(uint64_t pos, uint64_t len, uint64_t* out) const {
        if (n_seg == 0) return false;
        uint64_t s = hint < n_seg ? hint : 0;
        if (!(seg_ref_begin[s] <= pos && pos < seg_ref_begin[s + 1])) {
            // tasks walk the tape forward, so try the next segment before searching
            if (s + 1 < n_seg && seg_ref_begin[s + 1] <= pos && pos < seg_ref_begin[s + 2]) ++s;
            else {
                uint64_t lo = 0, hi = n_seg;          // last segment with begin <= pos
                while (hi - lo > 1) { uint64_t mid = (lo + hi) / 2; if (seg_ref_begin[mid] <= pos) lo = mid; else hi = mid; }
                s = lo;
            }
        }
        // zero-length tasks may sit exactly on a segment end
        if (pos + len > seg_ref_begin[s + 1] || pos < seg_ref_begin[s]) {
            if (len == 0 && pos == seg_ref_begin[n_seg]) { *out = seg_proteome_off[n_seg - 1] + (pos - seg_ref_begin[n_seg - 1]); return true; }
            return false;
        }
        hint = s;
        *out = seg_proteome_off[s] + (pos - seg_ref_begin[s]);
        return true;
    }
};

}  // namespace v2p
