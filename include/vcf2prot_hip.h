/*
 * vcf2prot_hip.h -- C ABI of the MI355X (gfx950) backend engine for vcf2prot's
 * step 6, the SIR executor.  This is what the reference's `-g gpu` plugin arm
 * binds (Rust `extern "C"` block shown in INTEGRATION.md).
 *
 * Reference interfaces replaced (paths under /root/reference/src):
 *   data_structures/InternalRep/gir.rs:236-239   Engine::GPU arm of GIR::execute
 *       -> v2p_execute_gir(); argument shapes = the SoA marshaller the reference
 *          already carries for its device engine (gir.rs:283-299) with
 *          usize -> uint64_t and char -> uint32_t.
 *   data_structures/InternalRep/gir.rs:203-229   DEBUG_CPU_EXEC validation
 *   README.md:156-157 (DEBUG_GPU)                -> v2p_validate_gir()
 *   data_structures/InternalRep/engines.rs:17-29 Engine::from_str -> v2p_engine_from_str()
 *   data_structures/InternalRep/personalized_genome.rs:61-69 + parts/exec.rs:34-40
 *       (two GIRs per sample, many samples in flight) -> the v2p_batch_* calls,
 *       which execute many haplotypes per launch from one concatenated image.
 *
 * Conventions: plain C, no exceptions cross the boundary.  Every call returns an
 * int status: 0 = V2P_OK, negative = error; the message (and, for task errors,
 * the offending row) is retrievable with v2p_last_error()/v2p_last_error_index().
 * The reference's convention for every one of these errors is panic!().
 * The caller owns every host buffer for the duration of a call; the library
 * copies what it needs and retains nothing.  Device memory belongs to the ctx.
 * A ctx (and its batches) may be used from one thread at a time; create one ctx
 * per worker thread (Rayon worker) -- contexts are independent and each owns a
 * HIP stream.
 * Environment (debugging only, read once per process): V2P_DEBUG_POISON=1 fills every device buffer with 0xA5 whenever
 * a call sizes it -- reused allocations included -- so that no result can depend on what fresh or recycled device memory held;
 * V2P_DECODE_CURSOR64 (v2p_frontend.h) forces the decode's 64-bit cursor kernels; the coalescing queue of v2p_execute_gir_shared
 * takes V2P_COALESCE_MB / _US / _BATCHES / _PROFILE (below).  Launch options are set through v2p_set_launch_opts, not the environment.
 */
#ifndef VCF2PROT_HIP_H
#define VCF2PROT_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)      /* the libraries are built with -fvisibility=hidden: what these headers declare is what they export */
#endif

#define V2P_OK                   0
#define V2P_BUSY                 1   /* v2p_gir_submit: every batch of the queue is in flight -- collect a ticket, then submit again; v2p_pipeline_submit_stream: every slot is in use */
#define V2P_ERR_INVALID_ARG     -1
#define V2P_ERR_HIP             -2   /* HIP runtime error, no device, out of memory        */
#define V2P_ERR_BAD_CODE        -3   /* exe_code not in {0,1} (haplotype_instruction.rs:154) */
#define V2P_ERR_RES_OOB         -4   /* start_pos_res+length beyond the result tape (task.rs:43,47) */
#define V2P_ERR_SRC_OOB         -5   /* start_pos+length beyond the ref/alt tape (task.rs:43,47)    */
#define V2P_ERR_NOT_CONTIGUOUS  -6   /* gir.rs:208-226 predicate failed                   */
#define V2P_ERR_NOT_CANONICAL   -7   /* batch image needs ascending, non-overlapping result ranges */
#define V2P_ERR_NON_BYTE_CHAR   -8   /* a char > 0xFF cannot enter the 1-byte-per-residue batch image */
#define V2P_ERR_UNSUPPORTED     -9
#define V2P_ERR_STATE          -10   /* call sequence error (e.g. execute before finalize) */

/* engines.rs:15 */
#define V2P_ENGINE_ST  0
#define V2P_ENGINE_MT  1
#define V2P_ENGINE_GPU 2

/* v2p_init flags */
#define V2P_FLAG_DEBUG_GPU   1u   /* validate every GIR on the device before executing it (DEBUG_GPU) */
#define V2P_FLAG_TEMPORAL    2u   /* plain result stores instead of non-temporal ones                  */
#define V2P_FLAG_RESULT_ORDER 4u  /* launch chunks in result order instead of the XCD-aware order      */

typedef struct v2p_ctx v2p_ctx;
typedef struct v2p_batch v2p_batch;

/* 16-byte work item of the device image (see vcf2prot_amd/csrc/sir_pack.hpp): first descriptor; result offset (48 bits) |
 * descriptor count (11 bits) | kernel routing flags (bits 60..63, set by the image builder) */
typedef struct { uint64_t task_begin; uint64_t dst_n; } v2p_chunk;

/* ---- library / context ------------------------------------------------------- */
const char* v2p_version(void);
/* number of visible HIP devices, or a negative status */
int v2p_device_count(void);
/* engines.rs:17-29: "st"/"ST"/"mt"/"MT"/"gpu"/"GPU"; anything else -> V2P_ERR_INVALID_ARG */
int v2p_engine_from_str(const char* name, int* engine);

int  v2p_init(int device_ordinal, unsigned flags, v2p_ctx** out);
void v2p_destroy(v2p_ctx* ctx);
/* message of the last failing call on this ctx (ctx may be NULL: last failing v2p_init of this thread) */
const char* v2p_last_error(const v2p_ctx* ctx);
/* row (task index) the last task error refers to, -1 if none */
int64_t v2p_last_error_index(const v2p_ctx* ctx);
/* run the engine on a caller-provided hipStream_t instead of the ctx's own (NULL restores it) */
int v2p_set_stream(v2p_ctx* ctx, void* hip_stream);

/* ---- GIR-faithful mode: one haplotype per call ------------------------------- */
/* Engine::GPU arm of GIR::execute (gir.rs:197-241).  Executes
 *     res[start_pos_res[i] .. +length[i]] = (code[i]==0 ? ref : alt)[start_pos[i] .. +length[i]]
 * for i = 0..n_tasks in order (later tasks win where ranges overlap) on the GPU.
 * Tapes hold Rust chars (uint32_t); `res` comes in as the caller filled it
 * (haplotype_instruction.rs:78 fills '.') and cells no task covers are left as they are. */
int v2p_execute_gir(v2p_ctx* ctx,
                    const uint8_t* code, const uint64_t* start_pos, const uint64_t* length,
                    const uint64_t* start_pos_res, uint64_t n_tasks,
                    const uint32_t* ref, uint64_t n_ref,
                    const uint32_t* alt, uint64_t n_alt,
                    uint32_t* res, uint64_t n_res);

/* The same arm for MANY workers sharing ONE context (the reference enters GIR::execute from every Rayon worker: parts/exec.rs:36-39,
 * personalized_genome.rs:64-65).  Thread-safe: concurrent calls are gathered for V2P_COALESCE_US microseconds (default 100) into one
 * batch of at most V2P_COALESCE_MB megabytes of tapes (default 16) -- every caller narrows its own tapes and packs its own
 * descriptors into the batch's pinned staging on its own thread -- then ONE upload, ONE launch and ONE download serve them all, and
 * up to V2P_COALESCE_BATCHES batches (default 8) are alive on their own streams.  `code` holds the exec codes as the SoA marshaller has them
 * (gir.rs:283-299, usize -> uint64_t: no narrowing copy on the Rust side); any value other than 0 / 1 is V2P_ERR_BAD_CODE
 * (haplotype_instruction.rs:154).  Overlapping / descending Task vectors, tapes with a char above 0xFF and GIRs larger than a
 * batch take the one-haplotype path of v2p_execute_gir, serialised on the context.  *err_row (may be NULL): the task row a task
 * error refers to, -1 otherwise -- v2p_last_error() is shared by the callers of a context and only advisory here.
 * A lone caller pays the gathering window on every call: single-threaded hosts use v2p_execute_gir. */
int v2p_execute_gir_shared(v2p_ctx* ctx,
                           const uint64_t* code, const uint64_t* start_pos, const uint64_t* length,
                           const uint64_t* start_pos_res, uint64_t n_tasks,
                           const uint32_t* ref, uint64_t n_ref,
                           const uint32_t* alt, uint64_t n_alt,
                           uint32_t* res, uint64_t n_res, int64_t* err_row);
/* The same arm in two halves, for a worker that has something else to do while its batch is on the GPU -- packing the next haplotype's
 * GIR (personalized_genome.rs:64-65 calls GIR::execute for haplotype 1, then for haplotype 2): v2p_gir_submit checks the tasks,
 * narrows the tapes and stages this call's share of a batch on the calling thread, then returns; the batch's own runner thread
 * closes it when the gathering window ends, uploads, launches, downloads; v2p_gir_collect waits for that and widens the result into
 * `res`.  Every array given to submit (and `res`) must stay valid and untouched until collect, which must be called exactly once per
 * ticket -- also after an error, which it reports (a task the reference would panic on is reported by collect, not by submit).
 * A batch slot is recycled when all its members have collected, so submit never waits for one: when every batch is in flight it
 * returns V2P_BUSY (nothing staged, *ticket = NULL) and the caller collects its oldest ticket before submitting again -- a worker
 * with two haplotypes in flight cannot deadlock the queue.  v2p_execute_gir_shared is submit (waiting for a slot) + collect. */
typedef struct v2p_gir_ticket v2p_gir_ticket;
int v2p_gir_submit(v2p_ctx* ctx,
                   const uint64_t* code, const uint64_t* start_pos, const uint64_t* length,
                   const uint64_t* start_pos_res, uint64_t n_tasks,
                   const uint32_t* ref, uint64_t n_ref,
                   const uint32_t* alt, uint64_t n_alt,
                   uint32_t* res, uint64_t n_res, v2p_gir_ticket** ticket);
int v2p_gir_collect(v2p_ctx* ctx, v2p_gir_ticket* ticket, int64_t* err_row);
/* batches launched / calls served by v2p_execute_gir_shared on this context so far */
int v2p_coalesce_stats(v2p_ctx* ctx, uint64_t* n_batches, uint64_t* n_calls);

/* DEBUG_GPU: inspect the device input arrays for indexing errors.  *first_bad = first
 * offending row or -1; *reason = V2P_ERR_BAD_CODE / _RES_OOB / _SRC_OOB / _NOT_CONTIGUOUS or 0.
 * Returns V2P_OK when the inspection ran (whatever it found). */
int v2p_validate_gir(v2p_ctx* ctx,
                     const uint8_t* code, const uint64_t* start_pos, const uint64_t* length,
                     const uint64_t* start_pos_res, uint64_t n_tasks,
                     uint64_t n_ref, uint64_t n_alt, uint64_t n_res,
                     int64_t* first_bad, int* reason);

/* ---- resident reference ------------------------------------------------------ */
/* Upload the reference proteome once (1 byte per residue, transcripts back to back).
 * Batches built with v2p_batch_add_haplotype() read reference residues from it. */
int v2p_upload_proteome(v2p_ctx* ctx, const uint8_t* aa, uint64_t n);
/* Same, plus a resident table of FASTA record headers (">NAME_1\n" ... each ending in '\n') stored
 * behind the proteome; needed by v2p_batch_add_haplotype_fasta(). */
int v2p_upload_reference(v2p_ctx* ctx, const uint8_t* aa, uint64_t n, const uint8_t* record_headers, uint64_t n_headers);

/* ---- batched native mode: many haplotypes per launch -------------------------- */
int  v2p_batch_create(v2p_ctx* ctx, v2p_batch** out);
void v2p_batch_destroy(v2p_batch* b);

/* Append one haplotype given exactly as a GIR (private ref tape travels with the batch). */
int v2p_batch_add_gir(v2p_batch* b,
                      const uint8_t* code, const uint64_t* start_pos, const uint64_t* length,
                      const uint64_t* start_pos_res, uint64_t n_tasks,
                      const uint32_t* ref, uint64_t n_ref,
                      const uint32_t* alt, uint64_t n_alt,
                      uint64_t n_res);

/* Append one haplotype whose ref tape is the concatenation of n_seg proteome transcripts
 * (haplotype_instruction.rs:118,130): segment s covers ref-tape offsets
 * [seg_ref_begin[s], seg_ref_begin[s+1]) and starts at proteome offset seg_proteome_off[s].
 * Code-0 tasks are rebased onto the resident proteome; no reference bytes travel. */
int v2p_batch_add_haplotype(v2p_batch* b,
                            const uint8_t* code, const uint64_t* start_pos, const uint64_t* length,
                            const uint64_t* start_pos_res, uint64_t n_tasks,
                            const uint64_t* seg_ref_begin, const uint64_t* seg_proteome_off, uint64_t n_seg,
                            const uint8_t* alt, uint64_t n_alt,
                            uint64_t n_res);

/* FASTA emit fused into the scatter (replaces personalized_genome.rs:66-67,90-113 +
 * sequence_tape.rs:77-89 on the host): like v2p_batch_add_haplotype(), but the haplotype's arena
 * range holds file-ready bytes -- for every record i, in order, header bytes, the residues of
 * result range [rec_res_end[i-1], rec_res_end[i]), '\n'.  Records must tile the result tape;
 * rec_header_off/len address the resident header table of v2p_upload_reference(); every header
 * ends in '\n', and when the byte in front of each header is '\n' too (headers stored back to back
 * behind one leading line feed) the closing line feed and the next header are emitted as one descriptor. */
int v2p_batch_add_haplotype_fasta(v2p_batch* b,
                                  const uint8_t* code, const uint64_t* start_pos, const uint64_t* length,
                                  const uint64_t* start_pos_res, uint64_t n_tasks,
                                  const uint64_t* seg_ref_begin, const uint64_t* seg_proteome_off, uint64_t n_seg,
                                  const uint8_t* alt, uint64_t n_alt, uint64_t n_res,
                                  const uint64_t* rec_res_end, const uint64_t* rec_header_off, const uint32_t* rec_header_len,
                                  uint64_t n_rec);

/* Step 5 folded into the builder (replaces the host loop haplotype_instruction.rs:94-133): feed the
 * per-transcript GIRs exactly as TranscriptInstruction::get_g_rep returns them
 * (transcript_instructions.rs:335-427: offsets relative to the transcript, its own small alt tape),
 * in the order they should appear in the result.  Reference tasks read the resident proteome at
 * tx_proteome_off; no private ref tape is built, no counters are kept by the caller.  An empty GIR
 * (start-lost transcript, :338-343) is n_tasks = 0, res_len = 0.  header_len != 0 additionally emits
 * the FASTA record text (header from the resident table, residues, line feed). */
int v2p_batch_begin_haplotype(v2p_batch* b);
int v2p_batch_add_transcript(v2p_batch* b,
                             const uint8_t* code, const uint64_t* start_pos, const uint64_t* length,
                             const uint64_t* start_pos_res, uint64_t n_tasks,
                             uint64_t tx_proteome_off, uint64_t tx_ref_len,
                             const uint8_t* alt, uint64_t n_alt, uint64_t res_len,
                             uint64_t header_off, uint32_t header_len);
int v2p_batch_end_haplotype(v2p_batch* b);

/* ---- image build ON THE DEVICE (step 5 and the packer as kernels) ----------------------------
 * Input: the per-transcript GIRs exactly as TranscriptInstruction::get_g_rep returns them (transcript_instructions.rs:335-427:
 * task offsets relative to the transcript and to its own alt tape), concatenated over the transcripts of every haplotype of the
 * batch, in result order.  The device does the reference's step 5 (haplotype_instruction.rs:94-133: the three running sums become
 * prefix scans; reference tasks are rebased onto the resident proteome) and the image packing (result-order descriptors, '.'
 * fill for cells no task covers, immediate descriptors, chunk table, XCD-aware order, hap_out_begin). */
typedef struct {
    uint64_t n_haps, n_tx, n_tasks, n_alt;
    const uint64_t* hap_tx_begin;      /* [n_haps + 1] transcripts of each haplotype                                   */
    const uint64_t* tx_proteome_off;   /* [n_tx] where the transcript's reference sits in the resident proteome        */
    const uint32_t* tx_ref_len;        /* [n_tx] its length (bounds of code-0 tasks)                                    */
    const uint32_t* tx_res_len;        /* [n_tx] result length of the transcript GIR (cells no task covers keep '.')    */
    const uint64_t* tx_task_begin;     /* [n_tx + 1]                                                                    */
    const uint64_t* tx_alt_begin;      /* [n_tx + 1]                                                                    */
    const uint8_t*  code;              /* [n_tasks] task.rs:2-9, un-rebased                                             */
    const uint32_t* start_pos;         /* [n_tasks]                                                                     */
    const uint32_t* length;            /* [n_tasks]                                                                     */
    const uint32_t* start_pos_res;     /* [n_tasks]                                                                     */
    const uint8_t*  alt;               /* [n_alt] alt tapes of the transcripts back to back, 1 byte per residue         */
    /* FASTA emit fused into the build (personalized_genome.rs:90-113; NULL, NULL: plain result tapes).  Transcript t's arena range
     * then holds its record: the header text at [tx_header_off[t], +tx_header_len[t]) of the resident header table
     * (v2p_upload_reference; it ends in '\n'), its residues, '\n' -- a haplotype's arena range is file-ready.  tx_header_len[t] == 0
     * writes the residues of that transcript alone. */
    const uint64_t* tx_header_off;     /* [n_tx] */
    const uint32_t* tx_header_len;     /* [n_tx] */
} v2p_txstream;
/* kernel 6 / 7 -- ROWS images, the default since round 4 (vcf2prot_amd/csrc/rows_image.hpp, build_rows.hip): ONE pass over the stream,
 * lane = Task; descriptors are written whole (nothing is cut at a chunk boundary), chunks are cut afterwards on 1 KiB rows of the
 * arena, greedily, as many rows as the kernel takes while the descriptors fit -- 6: a wave image (stitchw_kernel: <= 10 rows and
 * <= 64 descriptors per chunk), 7: a dense image (stitch_dense_kernel: <= 12 rows, <= 1024 descriptors; two fused substitutions that
 * follow each other become one descriptor).  window_bytes is ignored.  A descriptor lying across a cut is shared by the two chunks:
 * the chunk record's task_begin holds first descriptor : 42 | bytes of it that belong to the chunk before : 11 | bytes of the chunk's
 * last descriptor that belong to the next : 11, and bit 59 of dst_n marks such chunks (they start on multiples of 1024).  A row with
 * more than 64 descriptors makes kernel 6 V2P_ERR_UNSUPPORTED (build a dense image), one with more than 1 024 makes kernel 7 V2P_ERR_UNSUPPORTED
 * (the host builder -- v2p_batch_add_transcript -- takes any stream).  The north star's cohort (10 000 samples, 9.6 GB of stream): 6.8 ms of kernels.
 * (The grid builders of rounds 2-3 -- kernel 1 .. 5, fixed windows of `window_bytes` -- and PATCH images -- kernel 8 -- were never picked by a
 * routing rule; since round 6 they live in the development library only: vcf2prot_amd/csrc/bench/v2p_bench.h.  Here: V2P_ERR_INVALID_ARG.)
 * The offset tables of the stream are checked on the host before anything is uploaded (ascending from 0, inside their arrays):
 * V2P_ERR_INVALID_ARG with the offending index.  V2P_ERR_UNSUPPORTED leaves the batch empty.
 * On success the batch is finalized (execute / sync / download / digests work as after v2p_batch_finalize).
 * *build_ms (optional): time of the build kernels alone (two HIP event brackets: counting passes, emitting passes; not the
 * allocation of the image in between), the stream already on the device. */
int v2p_batch_build_on_device(v2p_batch* b, const v2p_txstream* s, uint32_t window_bytes, int kernel, float* build_ms);
/* ---- Task vectors -> result bytes ONCE, in one call (round 5) ------------------------------------------------------------------
 * What a cohort pays exactly once is the reference's get_g_rep(..).execute(engine) (haplotype_instruction.rs:75-137 -> gir.rs:197-241,
 * personalized_genome.rs:64-65): build a haplotype's executable image, execute it.  v2p_batch_build_on_device + v2p_batch_execute do
 * that in two calls, the stream's H2D inside the first and the two strictly one after the other.  Here the stream is made RESIDENT
 * first (its tables are checked on the host as for v2p_batch_build_on_device, its arrays uploaded once, res_counter per tile of
 * transcripts and per haplotype -- haplotype_instruction.rs:90,132 as scans -- made on the device behind the upload), and ONE call
 * builds the rows image and executes it with no host round trip but one look at the counts in between.  A RICH stream (at most 125
 * result bytes per Task) builds a PADDED wave image: the descriptors stay in their tiles' slots (no compaction pass), the chunk records
 * address slots, and the launcher STAGES every phase's descriptors in launch order for the stitch kernel (vcf2prot_amd/csrc/
 * stitch_kernels.h).  The first v2p_batch_execute / v2p_batch_download_image behind the call makes such an image dense (the skipped
 * compaction, once): what a host ever sees, and what is executed again, is the image v2p_batch_build_from_stream builds.
 * (The image can also be built slice by slice while the slice before it is stitched -- n_slices -- which was measured slower.) */
typedef struct v2p_stream v2p_stream;
/* the stream's arrays to the device (the host copy may be freed on return); V2P_ERR_INVALID_ARG / _SRC_OOB with the offending index
 * as v2p_batch_build_on_device reports them.  The tables are checked on a thread beside the copy (C3 whole, 9.6 GB and 20 M transcripts:
 * 0.17 - 0.24 s, the link's rate) */
int  v2p_stream_upload(v2p_ctx* ctx, const v2p_txstream* s, v2p_stream** out);
/* Waits for the context's own streams (not for the device: other contexts keep running).  A batch built from the stream registers with
 * it, and its payload descriptors read the stream's alt bytes: destroying the stream ORPHANS those batches -- v2p_batch_execute on an
 * orphan is V2P_ERR_STATE, never a read of freed memory; its arena is the batch's own and stays readable (v2p_batch_download,
 * v2p_batch_digests, v2p_batch_sync); v2p_batch_reset makes the batch buildable again.  (The reference cannot dangle here: GIR::execute(self)
 * consumes its tapes, gir.rs:197,230-234.) */
void v2p_stream_destroy(v2p_stream* s);
int  v2p_stream_counts(const v2p_stream* s, uint64_t* n_haps, uint64_t* n_tx, uint64_t* n_tasks, uint64_t* out_bytes);
/* the one-piece builder on a resident stream, no H2D: kernel 6 / 7 (rows images), 9 (a TILE image, see v2p_batch_build_and_execute), 0: by the
 * routing rule -- a wave image from 24 result bytes per Task, below a tile image where the form takes the stream, else a dense rows image */
int  v2p_batch_build_from_stream(v2p_batch* b, const v2p_stream* s, int kernel, float* build_ms);
/* Build AND execute.  kernel: 0 (the routing rule), 6, 7, or 9 -- a TILE image (round 6; vcf2prot_amd/csrc/dense_pieces.h), what the rule picks for
 * DEEP Task vectors (fewer than 24 result bytes per Task: runs of substitutions, transcript_instructions.rs:508-629,654-663): the one-pass
 * parse writes PIECES -- at most 16 result bytes of one source, their offset inside the tile's result, at most one substituted residue --
 * straight into the slots of its tile of transcripts, and the tile is the executor's work item: no dense image first, no compaction, no row
 * map, no cutter, no chunk table, and nothing to re-write when the image is executed again.  A stream the form does not take (a tile of
 * transcripts with more than 16 368 result bytes or 2 048 pieces, sources beyond 2 GiB) builds a dense rows image under the rule and is
 * V2P_ERR_UNSUPPORTED by number.  n_slices: 0 or 1 (the image built slice by slice beside the stitch of the slice before was measured slower
 * on every cohort -- profiles/r05_oneshot_slices.json -- and lives in the development library).  Returns when the stitch kernels are enqueued on
 * the context's stream (like v2p_batch_execute: asynchronous); v2p_batch_sync collects the status.  The batch is finalized: execute /
 * digests / download work as after any build.  Streams the one-pass builder does not take (a tile of transcripts with more than 256
 * descriptors, a 1 KiB row with more descriptors than a chunk holds, more than 2^25 tiles) are built in its two-pass form and executed, in
 * the same call.  What the reference would panic on (update_task, Task::execute) is reported as by v2p_batch_build_on_device. */
int  v2p_batch_build_and_execute(v2p_batch* b, const v2p_stream* s, int kernel, uint32_t n_slices);
typedef struct {
    int32_t  kernel;             /* 6 / 7: the rows image that was built; 9: a tile image                                           */
    uint32_t n_slices;           /* 1; 0: the call fell back to the one-piece builder + one execute                                 */
    float    total_ms;           /* HIP events on the context's stream: before the first build kernel -> behind the last stitch kernel */
    float    build_ms;           /* sum of the slices' build kernels (they overlap the stitch of the slices before: not additive)   */
    double   call_wall_ms;       /* host wall-clock of the call (it returns when the last slice is enqueued)                        */
    float    slice_build_ms[32];
    float    tables_ms;          /* what runs in front of the first slice: the haplotype offsets' copy (the tables are the stream's) -- or, for
                                  * a stream whose tables do not fit the image kind asked for, arena bytes per tile, their scan, the offsets   */
} v2p_oneshot_info;
/* waits for the call's last kernel, then reports its times */
int  v2p_batch_oneshot_info(v2p_batch* b, v2p_oneshot_info* info);
/* a finalized batch back to empty, its device buffers kept: the next build recycles arena, descriptor array and scratch */
int  v2p_batch_reset(v2p_batch* b);
/* the image as it sits on the device (for checkers): sizes first (any pointer may be NULL), then the arrays.  (A padded image -- see
 * v2p_batch_build_and_execute -- is made dense first, for good.) */
int v2p_batch_download_image(v2p_batch* b, uint64_t* desc, v2p_chunk* chunks, uint64_t* hap_out_begin);

/* Adopt an already packed image (descriptors, chunks, payload, haplotype result ranges),
 * e.g. from the synthetic cohort generator (include/v2p_cohort.h).
 * Image format (vcf2prot_amd/csrc/sir_pack.hpp): 8-byte descriptors in result order -- src:40 | len:22 |
 * space:2 (0 resident proteome, 1 payload arena, 2 '.' fill, 3 immediate: the source field holds 1..5 literal bytes); space 3
 * with bit 61 = a fused substitution (src:29 | len1:12 | len2:12 | byte:8: copy, one literal byte, copy one residue on;
 * long-run and dense chunks only); space 3 with bits 61..60 = 01 = two substitutions in a row (src:29 | len1:5 | len2:5 | len3:5 |
 * byte1:8 | byte2:8; dense chunks only).  Chunks: {first descriptor, result offset:48 | descriptors:11 | flags: bit 63 long-run
 * (stitch4_kernel), bit 62 long-run with 257..512 tasks, bit 61 dense (stitch_dense_kernel), bit 60 wave (stitchw_kernel: at most 64
 * descriptors and 10240 result bytes incl. the 16-byte phase of its offset, fused substitutions allowed)}.  A descriptor a chunk's
 * kernel does not know is reported (source out of bounds) and the chunk is not executed. */
int v2p_batch_set_packed(v2p_batch* b,
                         const uint64_t* desc, uint64_t n_desc,
                         const v2p_chunk* chunks, uint64_t n_chunks,
                         const uint8_t* payload, uint64_t n_payload,
                         const uint64_t* hap_out_begin, uint64_t n_haps);

/* Cut the image into chunks (if built by add_*) and move it to the device. */
int v2p_batch_finalize(v2p_batch* b);
/* Enqueue one pass of the SIR executor over the whole batch on the ctx stream (asynchronous).
 * The first call that executes an image AGAIN may first bring it into its re-execution form, once: a padded wave image (what
 * v2p_batch_build_and_execute leaves behind for a rich stream) is made dense -- one copy kernel, enqueued, and an allocation of 8 bytes
 * per descriptor -- and a dense rows image (deep Task vectors) is re-written as pieces (vcf2prot_amd/csrc/dense_pieces.h): two kernels,
 * an allocation of about 11 bytes per descriptor and ONE wait for the stream in between (that call is not asynchronous: 1.5-2.5 ms for
 * 10^8 descriptors).  Results are the same bytes in every form. */
int v2p_batch_execute(v2p_batch* b);
/* Wait for the stream and collect the device status word; task errors surface here. */
int v2p_batch_sync(v2p_batch* b);

int v2p_batch_counts(const v2p_batch* b, uint64_t* n_haps, uint64_t* n_desc, uint64_t* n_chunks,
                     uint64_t* out_bytes, uint64_t* payload_bytes);
/* in what form the batch's image sits on the device right now (diagnostics, tests): bit 0 = padded wave image (v2p_batch_build_and_execute;
 * made dense by the next v2p_batch_execute / download), bit 1 = a piece image was built (a dense rows image that is executed again:
 * vcf2prot_amd/csrc/dense_pieces.h), bit 2 = staging buffers for the phases' descriptors exist, bit 3 = a TILE image (pieces in the tiles' slots:
 * v2p_batch_download_image has no descriptors or chunks to hand out, v2p_batch_counts reports pieces and tiles in their place) */
int v2p_batch_image_form(const v2p_batch* b);
/* result range of haplotype h inside the arena */
int v2p_batch_hap_range(const v2p_batch* b, uint64_t h, uint64_t* begin, uint64_t* len);
/* copy arena bytes [begin, begin+len) to the host (1 byte per residue) */
int v2p_batch_download(v2p_batch* b, uint64_t begin, uint64_t len, uint8_t* out);
/* per-haplotype digests computed on the device: sum_i (byte_i + 1) * 2^(8 * (i mod 8)) * splitmix64(i div 8)  (mod 2^64), i relative
 * to the haplotype's first byte -- word by word: sum_k splitmix64(k) * (little-endian word k + 0x01..01 over the bytes that exist).
 * (Round 5: one multiplier per 8 bytes; until round 4 it was one per byte and the kernel took 8 x an execute.) */
int v2p_batch_digests(v2p_batch* b, uint64_t* digests, uint64_t n_haps);
/* device pointer of the result arena (for callers that keep consuming on the GPU) */
void* v2p_batch_device_out(v2p_batch* b);
/* For checkers: overwrite the whole arena with `byte` (enqueued on the context's stream).  An image that is executed AGAIN writes the
 * bytes the first execute already left there, so a re-execution form that skipped chunks or wrote nothing would still pass a comparison:
 * every test, fuzzer and bench leg that verifies a re-execute scribbles first. */
int v2p_batch_scribble(v2p_batch* b, int byte);

/* ---- streamed pipeline: results that must return to the host ------------------------------ */
/* n_slots submissions are in flight at once: while slice k's results travel D2H, slice k+1 executes and slice k+2 travels H2D (pinned
 * staging on both sides).  Tickets are slot numbers and are reused round-robin; a slot must be released before it is submitted to again.
 *
 * v2p_pipeline_submit_stream (round 6) -- Task vectors in, host bytes out, nothing packed on the host: what the reference's driver does
 * per sample (parts/exec.rs:23-42: get_g_rep(..).execute(engine), personalized_genome.rs:61-69, then the bytes to the writer, :90-113)
 * for a SLICE of the cohort at a time.  `slice` is the transcript stream of a contiguous range of haplotypes, exactly as for
 * v2p_stream_upload (FASTA emit: tx_header_off / tx_header_len set -- the result is file-ready text).  The CALLING thread checks the
 * slice's tables (errors as v2p_stream_upload reports them) and copies its arrays into the slot's pinned staging with a small team of
 * copy threads (v2p_pipeline_reserve: how many), enqueues the H2D on the slot's stream and returns: the slice's arrays may be freed.
 * The pipeline's runner thread takes the slices in submission order through v2p_batch_build_and_execute's one call (kernel: 0, 6, 7 or 9)
 * on the context's streams and enqueues the arena's D2H into the slot's pinned result buffer on a stream of its own.  Submissions may
 * come from several threads (each stages its own slice; a submission takes the first free slot, and with every slot in use returns
 * V2P_BUSY -- nothing staged: wait for a ticket, release it, submit again).  v2p_pipeline_wait blocks until the result is in host memory -- without holding
 * the context: other threads submit meanwhile -- and reports what the reference would have panicked on; *result stays valid until
 * release.  V2P_SUBMIT_DIGESTS: the per-haplotype digests of v2p_batch_digests travel with the result (checkers).
 *
 * v2p_pipeline_submit (round 2) takes a host-PACKED device image (v2p_cohort_pack(), or a batch builder's output) against the
 * resident reference instead. */
typedef struct v2p_pipeline v2p_pipeline;
#define V2P_SUBMIT_DIGESTS 1u
typedef struct { double stage_ms; double runner_ms; } v2p_slice_times;   /* the submitter's check + staging copy; the runner's one call (enqueue + its looks at the counts) */
int  v2p_pipeline_create(v2p_ctx* ctx, uint32_t n_slots, v2p_pipeline** out);
void v2p_pipeline_destroy(v2p_pipeline* p);
/* optional: pin the slots' staging (stream_bytes: a slice's arrays, about 13 bytes per Task + 32 per transcript + its alt bytes) and result
 * buffers now rather than under the first submissions (pinning a GB takes a good fraction of a second); copy_threads (0: keep; default 8) */
int  v2p_pipeline_reserve(v2p_pipeline* p, uint64_t stream_bytes, uint64_t out_bytes, uint32_t copy_threads);
int  v2p_pipeline_submit_stream(v2p_pipeline* p, const v2p_txstream* slice, int kernel, unsigned flags, uint32_t* ticket);
int  v2p_pipeline_submit(v2p_pipeline* p,
                         const uint64_t* desc, uint64_t n_desc,
                         const v2p_chunk* chunks, uint64_t n_chunks,
                         const uint8_t* payload, uint64_t n_payload,
                         uint64_t out_bytes, uint32_t* ticket);
/* blocks until the submission's results are in host memory; *result stays valid until release */
int  v2p_pipeline_wait(v2p_pipeline* p, uint32_t ticket, const uint8_t** result, uint64_t* n);
/* a stream slice that has been waited for: where its haplotypes start inside *result ([n_haps + 1], res_counter of
 * haplotype_instruction.rs:90,132), its digests (NULL without V2P_SUBMIT_DIGESTS), its host-side times; any pointer may be NULL */
int  v2p_pipeline_result_info(v2p_pipeline* p, uint32_t ticket, const uint64_t** hap_out_begin, uint64_t* n_haps, const uint64_t** digests, v2p_slice_times* times);
int  v2p_pipeline_release(v2p_pipeline* p, uint32_t ticket);

/* ---- raw launchers on caller-owned device memory (torch tensors, other runtimes) ----- */
/* How a launch is done.  Zero-initialise, set what is needed: every 0 (store_sc1: -1) means "the library's choice". */
typedef struct {
    uint32_t nontemporal;        /* 1: non-temporal result stores (the default of every batch), 0: plain stores                       */
    uint32_t routing;            /* v2p_stitch_launch_bits() of the (host copy of the) chunk table: which kernels have work; a chunk
                                  * whose kernel is not named by it is not executed                                                  */
    uint64_t phase_bytes;        /* wave / long-run images are launched in phases of this many bytes of image (chunk records +
                                  * descriptors), each read ahead into the memory-side cache: 0 = 64 MB (28 MB where the image is more
                                  * than 3 % of its result), ~0 = one launch and no read-ahead                                      */
    uint32_t phase_min_chunks;   /* images with fewer chunks are launched at once (0 = 16384; tests lower it)                        */
    int32_t  store_sc1;          /* wave images: 1 / 0 force / forbid "sc1 nt" row stores, -1 = by the image's descriptor share      */
    uint32_t max_blocks;         /* != 0: cap the grid of the per-block kernel (persistent workgroups); refused for images with
                                  * long-run, dense or wave chunks                                                                  */
    uint32_t reserved;           /* 0 (the A/B switches of rounds 3-5 that lived here -- `variant` -- exist in the development library only:
                                  * vcf2prot_amd/csrc/bench/v2p_bench.h) */
} v2p_launch_opts;
/* src0/src1 must have 32 readable bytes before and after (the kernel loads whole aligned 16-byte
 * blocks around a task's bytes), and so must d_desc (16 before, 32 after: stitchw_kernel reads an immediate descriptor's literal
 * bytes as a stream out of the descriptor array itself); out must be 16-byte aligned; status is one device uint64 initialised
 * to ~0; chunks that point outside d_desc[0, n_desc) are reported in it, never followed. */
int v2p_stitch_launch_opts(void* hip_stream,
                           const uint64_t* d_desc, uint64_t n_desc, const v2p_chunk* d_chunks, uint32_t n_chunks,
                           const uint8_t* d_src0, uint64_t src0_len,
                           const uint8_t* d_src1, uint64_t src1_len,
                           uint8_t* d_out, uint64_t out_len,
                           uint64_t* d_status, const v2p_launch_opts* opts);
/* phase size, phase threshold and store policy of every batch this context executes from now on (A/B runs, tests); NULL: the defaults */
int v2p_set_launch_opts(v2p_ctx* ctx, const v2p_launch_opts* opts);
/* Host-side: which stitch kernels a chunk table needs and their tasks per lane (v2p_launch_opts.routing: bit 1 dense chunks,
 * bit 2 wave chunks, bit 3 a rows image, bit 4 no long-run chunk, bit 5 no per-block chunk, 6..7 / 8..11 tasks per lane): wave chunks go to
 * stitchw_kernel, long-run chunks to stitch4_kernel, the others to stitch_kernel (per block) or, when chunks hold more than 512
 * descriptors (short tasks), to stitch_dense_kernel.  A chunk flagged dense (bit 61 of dst_n) must hold at most 12288 bytes of result incl. its 16-byte phase --
 * the kernel's LDS image; the builders never make a larger one, and the kernel refuses one (status: result out of bounds) rather
 * than executing it. */
int v2p_stitch_launch_bits(const v2p_chunk* chunks, uint64_t n_chunks);
/* Host-side, no GPU work: the routing rules the library applies to an image of these sizes -- which kernel its builders pack it
 * for, how its chunk table is ordered, how it is launched -- so that hosts and tests can see them (measured:
 * profiles/r04_routing_sweep.json, tools/routing_sweep.py; pinned by tests/test_routing_rules.py). */
typedef struct v2p_routing {
    uint32_t wave_bytes_per_task;   /* an image with at least this many result bytes per Task is a wave image (stitchw_kernel: kernel 6 / 4), below a dense one (7 / 3) */
    uint32_t rich;                  /* 1: its descriptors are more than 3 % of its result */
    uint32_t phased;                /* 1: launched in phases with read-ahead (wave / long-run images of at least 16 384 chunks) */
    uint32_t store_sc1;             /* 1: "sc1 nt" row stores (thin images) */
    uint64_t phase_bytes;           /* bytes of image per phase: 28 MB rich, 64 MB thin */
    uint32_t order_blocks;          /* blocks of the arena inside which the chunk table is dealt to the XCDs (1: one order for the whole table) */
    uint32_t order_windows;         /* inside an XCD's proteome slice the table goes 1: window by window, 0: haplotype after haplotype (thin images of 2 GB and more; round 6) */
} v2p_routing;
int v2p_routing_rules(uint64_t n_desc, uint64_t n_chunks, uint64_t result_bytes, uint64_t proteome_len, int wave_image, v2p_routing* out);
/* Host-side: reorder a chunk table so that workgroup 8*j + x (XCD x) works on proteome slice x and, inside a slice, on one window
 * of it after the other -- for an image whose descriptors are more than 3 % of its result, block after block of the arena (equal
 * shares of the table's entries in result order, about eight times the proteome each): the reference reads of the workgroups in
 * flight share their windows AND their stores stay within one block (C3 whole: 9.2 -> 7.9 ms).  The table is brought into result
 * order first if it is not.  v2p_batch_finalize() does this itself (v2p_batch_build_on_device the same on the device); callers of v2p_stitch_launch_opts() may want it too.  Speed only: chunks are independent. */
int v2p_order_chunks_for_xcds(v2p_chunk* chunks, uint64_t n_chunks, const uint64_t* desc, uint64_t n_desc,
                              uint64_t proteome_len);
int v2p_digest_launch(void* hip_stream, const uint8_t* d_out, const uint64_t* d_hap_begin, uint64_t n_haps,
                      uint64_t out_bytes, uint64_t* d_digests);
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* VCF2PROT_HIP_H */
