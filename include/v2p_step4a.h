/*
 * v2p_step4a.h -- step 4a of vcf2prot restated (plain C ABI, host only): the sorted mutations of one
 * AltTranscript -> its Instruction list, i.e. TranscriptInstruction::from_alt_transcript
 * (/root/reference/src/data_structures/InternalRep/transcript_instructions.rs:33-160) with
 * Instruction::from_mutation and every interpret_* / validate_s_state of
 * /root/reference/src/data_structures/InternalRep/instruction.rs:20-760.
 *
 * With include/v2p_frontend.h in front (index, GPU bitmask decode, grouping) and include/v2p_step4b.h +
 * v2p_batch_add_transcript() (include/vcf2prot_hip.h) behind, a host without Rust goes from VCF text to FASTA bytes.
 * Lives in libv2p_cohort.so.
 */
#ifndef V2P_STEP4A_H
#define V2P_STEP4A_H
#include <stdint.h>
#include "v2p_step4b.h"
#include "v2p_frontend.h"
#ifdef __cplusplus
extern "C" {
#endif

/* mutation_ds.rs:78-131; ref_aa / mut_aa are the non-digit characters of the two halves of the amino-acid field, "*" if
 * there were none (text_parser.rs:118-145) */
typedef struct v2p_mutation_view {
    uint8_t     type;               /* index into Constants::SUP_TYPE (Constants.rs:3-8) */
    uint16_t    ref_aa_position;    /* 0-based */
    uint16_t    mut_aa_position;
    const char* ref_aa; uint32_t ref_aa_len;
    const char* mut_aa; uint32_t mut_aa_len;
} v2p_mutation_view;

#define V2P_4A_OK        0
#define V2P_4A_SKIP      1   /* Err(..): no supported instruction left (:52-55) or, with PANIC_INSPECT_ERR off, duplicates/overlaps */
#define V2P_4A_PANIC     2   /* the reference aborts: a panic! of an interpret_* function, or the INSPECT_INS_GEN checks (:57-144) */
#define V2P_4A_CAPACITY  3

#define V2P_4A_INSPECT_INS_GEN    1u   /* both are on unless NO_TEST is exported (cli.rs:337-368) */
#define V2P_4A_PANIC_INSPECT_ERR  2u

/* muts: sorted by mut_aa_position (AltTranscript::sort_alterations).  out[i].data points into muts[..].mut_aa. */
int v2p_transcript_instructions(const v2p_mutation_view* muts, uint64_t n, uint32_t flags,
                                v2p_instruction* out, uint64_t cap, uint64_t* n_out);

/* the parsed form of consequence `csq_id` held by a v2p_groups (valid while g lives); -1 if Mutation::new failed */
int v2p_groups_mutation_view(const v2p_groups* g, uint32_t csq_id, v2p_mutation_view* out);

#ifdef __cplusplus
}
#endif
#endif
