/*
 * v2p_cohort.h -- synthetic cohorts at the Task boundary (plain C ABI, no HIP).
 *
 * Produces, per haplotype, exactly what the reference's steps 4-5 hand to the
 * step-6 executor: the rebased Vec<Task> in SoA form (gir.rs:283-299), the alt
 * tape, the layout of the private ref tape, and the result annotation
 * (haplotype_instruction.rs:75-137).  Task shapes follow
 * transcript_instructions.rs:335-780 for the alteration kinds the generator
 * draws (missense, inframe insertion/deletion, frameshift, stop gained,
 * stop lost, start lost); SURVEY.md Appendix A tabulates them.  The shapes are
 * pinned against the reference binary by tests/golden/c1_example.* (the same
 * cohort written as a VCF and run through bins/Linux/vcf2prot).
 *
 * Used by bench.py, the tests, and as the feeder of v2p_batch_set_packed().
 */
#ifndef V2P_COHORT_H
#define V2P_COHORT_H

#include <stdint.h>
#include "vcf2prot_hip.h"   /* v2p_chunk */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct v2p_cohort v2p_cohort;
typedef struct v2p_hapbuf v2p_hapbuf;

enum { V2P_ALT_MISSENSE = 0, V2P_ALT_INSERTION = 1, V2P_ALT_DELETION = 2, V2P_ALT_FRAMESHIFT = 3,
       V2P_ALT_STOP_GAINED = 4, V2P_ALT_STOP_LOST = 5, V2P_ALT_KINDS = 6 };

typedef struct {
    uint64_t seed_proteome;    /* 7 in every preset                                             */
    uint64_t seed_cohort;
    uint32_t n_samples;        /* haplotypes = 2 * n_samples                                    */
    uint32_t n_transcripts;
    double   mean_len;
    uint32_t len_model;        /* 0: max(50, round(N(L, L/4)))   1: log-normal, median L         */
    uint32_t fixed_len;        /* != 0: every transcript has this length                         */
    uint32_t altered_per_hap;  /* 0: every transcript is altered in every haplotype              */
    uint32_t alts_fixed;       /* != 0: this many alterations per altered transcript             */
    double   alts_poisson;     /* else 1 + Poisson(lambda)                                       */
    double   mix[V2P_ALT_KINDS];
    uint32_t max_ins, max_del, max_fs_tail, max_sl_ext;
    double   p_start_lost;     /* altered transcript is a start_lost (empty record) instead      */
    double   p_empty_hap;      /* haplotype carries nothing                                      */
} v2p_cohort_params;

/* One haplotype as the executor receives it.  Pointers stay valid until the next
 * v2p_cohort_generate() on the same hapbuf. */
typedef struct {
    uint64_t n_tasks;
    const uint8_t*  code;
    const uint64_t* start_pos;
    const uint64_t* length;
    const uint64_t* start_pos_res;
    uint64_t n_alt;  const uint8_t* alt;          /* alt tape, 1 byte per residue                  */
    uint64_t n_res;                                /* result tape length                            */
    uint64_t n_ref;                                /* private ref tape length                       */
    uint64_t n_seg;                                /* transcripts concatenated in the ref tape      */
    const uint64_t* seg_ref_begin;                 /* [n_seg + 1]                                   */
    const uint64_t* seg_proteome_off;              /* [n_seg]                                       */
    uint64_t n_tx;                                 /* annotation: altered transcripts incl. empty ones */
    const uint32_t* tx_id;                         /* [n_tx]                                        */
    const uint64_t* tx_res_begin;                  /* [n_tx]                                        */
    const uint64_t* tx_res_end;                    /* [n_tx]                                        */
} v2p_hap_view;

typedef struct {
    uint64_t* desc;          uint64_t n_desc;
    v2p_chunk* chunks;       uint64_t n_chunks;
    uint8_t*  payload;       uint64_t n_payload;
    uint64_t* hap_out_begin; uint64_t n_haps;      /* [n_haps + 1] */
    uint64_t  n_tasks;       /* N: reference Task descriptors consumed (zero-length ones included) */
    uint64_t  n_copy_bytes;  /* A: residues written = Sum task.length                              */
    uint64_t  max_chunk_tasks; /* largest descriptor count of a chunk: v2p_stitch_launch needs ceil(this/256) per lane */
} v2p_packed_image;

/* "C1".."C5" (BASELINE.json configs, concretised in SURVEY.md section 8d) */
int  v2p_cohort_preset(const char* name, v2p_cohort_params* out);
int  v2p_cohort_create(const v2p_cohort_params* p, v2p_cohort** out);
void v2p_cohort_destroy(v2p_cohort* c);

uint64_t        v2p_cohort_n_haplotypes(const v2p_cohort* c);
uint32_t        v2p_cohort_n_transcripts(const v2p_cohort* c);
uint64_t        v2p_cohort_proteome_len(const v2p_cohort* c);
const uint8_t*  v2p_cohort_proteome(const v2p_cohort* c);      /* transcripts back to back         */
const uint64_t* v2p_cohort_tx_offsets(const v2p_cohort* c);    /* [n_transcripts + 1]              */

v2p_hapbuf* v2p_hapbuf_create(void);
void        v2p_hapbuf_destroy(v2p_hapbuf* b);
/* haplotype index h = 2 * sample + (0|1) */
int v2p_cohort_generate(const v2p_cohort* c, uint64_t hap, v2p_hapbuf* buf, v2p_hap_view* view);
/* the private ref tape step 5 would build for this haplotype, as Rust chars */
int v2p_cohort_ref_tape_u32(const v2p_cohort* c, const v2p_hap_view* view, uint32_t* out);
/* alterations of one haplotype as text, one per line: "<transcript index>\t<csq type>\t<aa change>\n"
 * (aa change in BCFtools/csq notation, e.g. 13I>13F).  Returns the bytes needed (excluding NUL). */
int64_t v2p_cohort_describe(const v2p_cohort* c, uint64_t hap, char* buf, uint64_t cap);

/* Device image of haplotypes [h0, h1) against the resident proteome, built on n_threads threads.
 * chunk_tasks / chunk_bytes = 0 select the library defaults (sir_pack.hpp). */
#define V2P_PACK_FASTA 1u   /* file-ready arena: ">ENST..._h\n" + residues + "\n" per record (needs the header table resident) */
#define V2P_PACK_PER_BLOCK 16u /* route every chunk to the per-block kernel (implies no fusion) */
#define V2P_PACK_LONG_RUN 32u  /* route every chunk of <= 512 tasks to the long-run kernel, whatever its shape */
#define V2P_PACK_DENSE 64u       /* the image goes to stitch_dense_kernel: chunks of <= 1024 short tasks, fused substitutions allowed */
#define V2P_PACK_WAVE 128u       /* the image goes to stitchw_kernel: one wave per chunk of <= 64 descriptors and <= 10 KiB, fused substitutions allowed */
#define V2P_PACK_NO_LINE_CUT 0x800000u /* wave images: no 128-byte fallback cut (A/B runs) */
#define V2P_PACK_NO_DOUBLE 8u /* dense images: one substitution per fused descriptor only (A/B runs) */
#define V2P_PACK_NO_FUSE 4u /* one descriptor per task: no fused substitutions (A/B runs, per-block kernel) */
#define V2P_PACK_NO_IMM 2u  /* keep short alt payloads in the payload arena instead of inside their descriptors (A/B runs) */
int  v2p_cohort_pack(const v2p_cohort* c, uint64_t h0, uint64_t h1, int n_threads,
                     uint32_t chunk_tasks, uint32_t chunk_bytes, uint32_t flags, v2p_packed_image* out);
/* result tape length (residues) of haplotypes [h0, h1): what SURVEY 8e's byte-balanced sharding cuts its prefix sum over */
int  v2p_cohort_result_sizes(const v2p_cohort* c, uint64_t h0, uint64_t h1, int n_threads, uint64_t* out);
/* The same haplotypes one step EARLIER: the per-transcript GIRs of step 4b, un-rebased (offsets relative to the transcript and to
 * its own alt tape), concatenated -- the input of v2p_batch_build_on_device (its field order; free with v2p_txstream_free). */
typedef struct {
    uint64_t n_haps, n_tx, n_tasks, n_alt;
    uint64_t* hap_tx_begin; uint64_t* tx_proteome_off; uint32_t* tx_ref_len; uint32_t* tx_res_len;
    uint64_t* tx_task_begin; uint64_t* tx_alt_begin;
    uint8_t* code; uint32_t* start_pos; uint32_t* length; uint32_t* start_pos_res; uint8_t* alt;
    uint64_t* tx_header_off; uint32_t* tx_header_len;      /* NULL from v2p_cohort_txstream (plain result tapes) */
} v2p_txstream_buf;
int  v2p_cohort_txstream(const v2p_cohort* c, uint64_t h0, uint64_t h1, int n_threads, v2p_txstream_buf* out);
void v2p_txstream_free(v2p_txstream_buf* s);
/* Host image of haplotypes [h0, h1) cut on a fixed result grid (what the device builder produces): v2p_cohort_pack with
 * ImageBuilder::grid_bytes = window_bytes; kernel 1 = long-run routing, 2 = per block */
int  v2p_cohort_pack_grid(const v2p_cohort* c, uint64_t h0, uint64_t h1, uint32_t window_bytes, int kernel, v2p_packed_image* out);
/* Host ROWS image of a transcript stream (vcf2prot_amd/csrc/rows_image.hpp) -- what v2p_batch_build_on_device(kernel 6: wave, 7: dense)
 * must reproduce byte for byte: whole descriptors in result order (nothing cut), chunks cut afterwards on 1 KiB rows of the arena
 * (head skip / row clip in the chunk records, include/vcf2prot_hip.h), chunk table in arena order.  mode 1 = wave image, 2 = dense.
 * emulate_k = 0: the sequential restatement of the packer's state machine; a power of two <= 64: the device kernel's tiles of that
 * many transcripts, its 64-item windows and ballot masks emulated lane by lane (the two must agree).  *status (optional): the
 * device-style status word (task << 8 | reason; ~0 = clean). */
int  v2p_txstream_pack_rows(const v2p_txstream_buf* s, uint64_t proteome_len, int mode, uint32_t emulate_k, v2p_packed_image* out, uint64_t* status);
/* PATCH image (round 5; kernel 8 of v2p_batch_build_on_device: segments + patches on a 12 KiB grid, include/vcf2prot_hip.h) of a transcript
 * stream built on the HOST -- the sequential restatement of the device builder's rules (vcf2prot_amd/csrc/patch_image_host.hpp) -- and an
 * interpreter that executes such an image cell by cell.  *status: ~0 or (index << 8 | reason) as the device reports it (reason 9: the
 * format declines the stream). */
typedef struct {
    uint64_t* seg; uint32_t* patch; v2p_chunk* chunks; uint64_t* hap_out_begin;
    uint64_t n_chunks, n_haps, out_bytes, n_seg, n_patch;
} v2p_patch_image;
int  v2p_txstream_pack_patch(const v2p_txstream_buf* s, uint64_t proteome_len, v2p_patch_image* out, uint64_t* status);
void v2p_patch_image_free(v2p_patch_image* im);
int  v2p_patch_interpret(const uint64_t* seg, const uint32_t* patch, const v2p_chunk* chunks, uint64_t n_chunks, const uint8_t* src0, uint64_t src0_len,
                         const uint8_t* src1, uint64_t src1_len, uint8_t* out, uint64_t out_len);
/* FASTA record headers of every transcript and haplotype parity: one leading '\n', then 19 bytes each;
 * header of (transcript t, parity p) at 1 + (2*t + p) * 19.  Returns the bytes needed; fills `out` when cap suffices. */
uint64_t v2p_cohort_fasta_headers(const v2p_cohort* c, uint8_t* out, uint64_t cap);
void v2p_packed_free(v2p_packed_image* img);
/* host-only twin of v2p_stitch_launch_bits() (include/vcf2prot_hip.h): which stitch kernels a chunk table needs -- so that host-side
 * tools and CPU tests need neither hipcc nor a HIP runtime to look at an image */
int  v2p_cohort_launch_bits(const v2p_chunk* chunks, uint64_t n_chunks);

#ifdef __cplusplus
}
#endif
#endif /* V2P_COHORT_H */
