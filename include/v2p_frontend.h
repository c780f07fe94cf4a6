/* v2p_frontend.h -- C ABI of the VCF front-end pieces next to the hot path:
 *
 *   (1) v2p_vcf_index_*   host, linear time: which record lines are supported, where their sample columns
 *                         are, and the flattened consequence table.  Replaces, for the GPU engine,
 *                         readers.rs:151-231 (get_records / return_if_supported) and
 *                         vcf_ds.rs:67-87 (get_consequences_vector).
 *   (2) v2p_decode_*      device (gfx950): BCSQ bitmask decode of every sample column of every record into
 *                         per-haplotype lists of consequence ids.  Replaces the Engine::GPU arm of
 *                         VCFRecords::get_csq_per_patient (vcf_ds.rs:192-211): get_patient_fields
 *                         (vcf_ds.rs:126-190), text_parser::get_bit_mask (text_parser.rs:163-252),
 *                         BitMask::from_string / get_indices (MaskDecoder.rs:33-153), extract_effects and the
 *                         SUP_TYPE filter of decode_back (vcf_ds.rs:213-329).
 *   (3) v2p_groups_*      host, O(n log n) per haplotype: group_muts_per_transcript (vcf_tools.rs:82-96, quadratic
 *                         in the reference) + AltTranscript::drop_replicate (vcf_ds.rs:387-420), in id space.
 *
 * Where the reference aborts (panic!) these calls return a negative status; the binding maps it back to panic!.
 * libvcf2prot_hip.so exports (2); libv2p_cohort.so (plain C++) exports (1) and (3).
 */
#ifndef V2P_FRONTEND_H
#define V2P_FRONTEND_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)      /* the libraries are built with -fvisibility=hidden: what these headers declare is what they export */
#endif

struct v2p_ctx;                                   /* include/vcf2prot_hip.h */

/* status codes of this header (continue the numbering of vcf2prot_hip.h) */
#define V2P_ERR_MASK_NEGATIVE   (-20)   /* text_parser.rs:210,244  "An invalid bit mask was encountered"            */
#define V2P_ERR_MASK_PARSE      (-21)   /* MaskDecoder.rs:41,47    parse::<u32>().unwrap() on a bad word            */
#define V2P_ERR_MASK_INDEX      (-22)   /* vcf_ds.rs:321,324       bit set for a consequence the record lacks       */
#define V2P_ERR_COLUMNS         (-23)   /* vcf_ds.rs:148           record with a different number of sample columns:
                                           more -> the reference panics; fewer -> it silently pairs fields with the
                                           wrong consequences, which this engine refuses                             */
#define V2P_ERR_FIELD_TOO_LONG  (-24)   /* a sample column whose text after the last ':' exceeds 4 KiB             */
#define V2P_ERR_CAPACITY        (-25)   /* raw launcher only: ids / multi-word capacity too small (needed size reported) */
#define V2P_ERR_VCF_FORMAT      (-26)   /* readers.rs:113-150      no "#CHROM" line, fewer than 10 columns, no records */
#define V2P_ERR_DUPLICATE_POS   (-27)   /* vcf_ds.rs:411           two different mutations on one reference position  */

/* ---------------------------------------------------------------------------------------------------------
 * (1) record index (host)
 * ------------------------------------------------------------------------------------------------------- */
typedef struct v2p_vcf_index v2p_vcf_index;

/* `text` = the whole VCF file; it must stay alive and unchanged while the index is used (offsets point into it). */
int  v2p_vcf_index_build(const uint8_t* text, uint64_t n_bytes, v2p_vcf_index** out);
void v2p_vcf_index_destroy(v2p_vcf_index* x);
const char* v2p_vcf_index_error(const v2p_vcf_index* x);          /* message of the failed build ("" if none)      */
uint64_t v2p_vcf_index_n_samples(const v2p_vcf_index* x);
uint64_t v2p_vcf_index_n_records(const v2p_vcf_index* x);        /* supported records (readers.rs:185-231)        */
uint64_t v2p_vcf_index_n_consequences(const v2p_vcf_index* x);
/* sample name i: offset and length in `text` (readers.rs:113-150) */
int  v2p_vcf_index_sample(const v2p_vcf_index* x, uint64_t i, uint64_t* begin, uint64_t* len);
/* [n_records] byte ranges of the sample columns (after the 9th tab, vcf_ds.rs:148) */
const uint64_t* v2p_vcf_index_row_begin(const v2p_vcf_index* x);
const uint64_t* v2p_vcf_index_row_end(const v2p_vcf_index* x);
/* [n_records + 1] first consequence id of each record; ids number the comma-separated BCSQ entries file-wide */
const uint32_t* v2p_vcf_index_csq_begin(const v2p_vcf_index* x);
/* [n_consequences] 1 = type is in Constants::SUP_TYPE (decode_back's filter, vcf_ds.rs:272) */
const uint8_t*  v2p_vcf_index_csq_supported(const v2p_vcf_index* x);
/* [n_consequences] byte range of each consequence string in `text` */
const uint64_t* v2p_vcf_index_csq_text_begin(const v2p_vcf_index* x);
const uint32_t* v2p_vcf_index_csq_text_len(const v2p_vcf_index* x);

/* ---------------------------------------------------------------------------------------------------------
 * (2) BCSQ bitmask decode (device).  Haplotype h of sample s is list 2*s + (h-1).
 * ------------------------------------------------------------------------------------------------------- */
typedef struct v2p_decode v2p_decode;

/* One call = VCFRecords::get_csq_per_patient for all probands.  Host pointers; the text is uploaded once.
 * On success *out holds the per-haplotype id lists on the device.  On a reference abort returns the
 * V2P_ERR_MASK_* / V2P_ERR_COLUMNS code; v2p_last_error_index(ctx) = record * n_samples + sample. */
int  v2p_decode_run(struct v2p_ctx* ctx,
                    const uint8_t* text, uint64_t n_text,
                    const uint64_t* row_begin, const uint64_t* row_end, uint64_t n_records, uint64_t n_samples,
                    const uint32_t* csq_begin /*[n_records+1]*/, const uint8_t* csq_supported /*[csq_begin[n_records]]*/,
                    v2p_decode** out);
/* hap_begin[2*n_samples + 1]: exclusive prefix sums of the list lengths */
int  v2p_decode_counts(const v2p_decode* d, uint64_t* hap_begin);
/* ids[hap_begin[2*n_samples]]: the lists back to back, each in record order, then mask-word order, then bit order
 * (the order decode_back pushes them, vcf_ds.rs:262-292) */
int  v2p_decode_download(v2p_decode* d, uint32_t* ids);
/* device pointers of the same two arrays, for engines that keep going on the GPU */
int  v2p_decode_device(const v2p_decode* d, const uint64_t** d_hap_begin, const uint32_t** d_ids);
/* kernel time of the last run in milliseconds (HIP events around the four kernels), for benches */
int  v2p_decode_timing(const v2p_decode* d, float* ms_parse, float* ms_count, float* ms_scan, float* ms_emit);
void v2p_decode_destroy(v2p_decode* d);

/* Raw launcher on device buffers the caller owns (benches, profilers).  All pointers are device pointers.
 *   d_text            the file text; 16 readable bytes before d_text and after d_text + n_text
 *   d_sup_pairs       [n_records] pair mask of the first mask word: bits 2j and 2j+1 set iff consequence j of the
 *                     record exists and is supported (j < 16)
 *   d_sup_bits        bitset over consequence ids (bit i of word i/32), supported flags
 *   d_workspace       v2p_decode_workspace_bytes(n_records, n_samples, ovf_words) bytes, 256-byte aligned
 *   d_hap_begin       [2*n_samples + 1] out
 *   d_ids / ids_capacity   out; if the lists need more than ids_capacity entries nothing is written to d_ids and
 *                     V2P_ERR_CAPACITY is reported through d_status (d_hap_begin is still valid)
 *   d_status          [2] u64: [0] min over offending fields of (record*n_samples + sample) << 8 | reason, ~0 = clean;
 *                     [1] multi-word words needed
 * phases: bit 0 parse, bit 1 count, bit 2 scan, bit 3 emit (15 = all); kernels are enqueued on hip_stream. */
uint64_t v2p_decode_workspace_bytes(uint64_t n_records, uint64_t n_samples, uint64_t ovf_words);
int  v2p_decode_launch(void* hip_stream,
                       const uint8_t* d_text, uint64_t n_text,
                       const uint64_t* d_row_begin, const uint64_t* d_row_end, uint64_t n_records, uint64_t n_samples,
                       const uint32_t* d_csq_begin, const uint32_t* d_sup_pairs, const uint32_t* d_sup_bits,
                       uint8_t* d_workspace, uint64_t ovf_words,
                       uint64_t* d_hap_begin, uint32_t* d_ids, uint64_t ids_capacity,
                       uint64_t* d_status, unsigned phases);

/* ---------------------------------------------------------------------------------------------------------
 * (3) grouping per transcript (host)
 * ------------------------------------------------------------------------------------------------------- */
typedef struct v2p_groups v2p_groups;

/* parsed consequence (mutation_ds.rs:78-131); valid = 0 where Mutation::new(...) is Err */
typedef struct v2p_mutation {
    uint32_t transcript;        /* rank of the transcript id among the file's sorted unique ids, ~0u if none */
    uint16_t ref_aa_position;   /* 0-based (mutation_ds.rs:96-97) */
    uint16_t mut_aa_position;
    uint8_t  type;              /* index into Constants::SUP_TYPE */
    uint8_t  valid;
    uint8_t  pad_[2];
} v2p_mutation;

/* Groups every haplotype's consequence ids by transcript the way vcf_tools.rs:82-96 does (sorted unique
 * transcript ids; a consequence joins every group whose id occurs anywhere in its text; Mutation::new failures
 * dropped), sorts each group by mut_aa_position and applies drop_replicate.  n_threads = 0 -> hardware threads. */
int  v2p_groups_build(const v2p_vcf_index* x, const uint8_t* text,
                      const uint64_t* hap_begin, const uint32_t* ids, uint64_t n_haps, uint32_t n_threads,
                      v2p_groups** out);
void v2p_groups_destroy(v2p_groups* g);
const char* v2p_groups_error(const v2p_groups* g);
int64_t v2p_groups_error_haplotype(const v2p_groups* g);
uint64_t v2p_groups_n_transcripts(const v2p_groups* g);          /* unique transcript ids of the file, sorted */
int  v2p_groups_transcript(const v2p_groups* g, uint64_t rank, uint64_t* begin, uint64_t* len);   /* id text */
const v2p_mutation* v2p_groups_mutations(const v2p_groups* g);   /* [n_consequences] */
/* CSR: haplotype h owns groups [hap_group_begin[h], hap_group_begin[h+1]); group k is transcript group_transcript[k]
 * with members member_ids[group_member_begin[k] .. group_member_begin[k+1]) (consequence ids, final order) */
const uint64_t* v2p_groups_hap_group_begin(const v2p_groups* g);
const uint32_t* v2p_groups_group_transcript(const v2p_groups* g);
const uint64_t* v2p_groups_group_member_begin(const v2p_groups* g);
const uint32_t* v2p_groups_member_ids(const v2p_groups* g);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
