/*
 * v2p_step4b.h -- step 4b of vcf2prot restated (plain C ABI, host only): the Instruction list of
 * one transcript -> its Vec<Task>, alt tape and result length, i.e.
 * TranscriptInstruction::get_g_rep (/root/reference/src/data_structures/InternalRep/
 * transcript_instructions.rs:335-427) with to_task :452-505, add_till_next_ins :508-629,
 * add_last_instruction :633-651, get_task_from_* :654-780, build_base_instruction :713-736 and
 * compute_expected_results_array_size :214-321.
 *
 * Together with v2p_batch_add_transcript() (include/vcf2prot_hip.h) a non-Rust host can go from
 * Instructions to FASTA bytes.  Producing Instructions from mutations (instruction.rs) is not part
 * of this library.
 */
#ifndef V2P_STEP4B_H
#define V2P_STEP4B_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* instruction.rs:6-15 */
typedef struct {
    char        code;       /* 'M','N','I','J','D','C','F','R','K','B','Y','G','X','A','T','L','W','Q','Z','P','2','3','0','U' */
    uint8_t     s_state;
    uint64_t    pos_ref, pos_res, len;
    const char* data;       /* payload residues (not NUL terminated) */
    uint64_t    data_len;
} v2p_instruction;

#define V2P_4B_OK            0
#define V2P_4B_MUST_BE_LAST  1   /* Err(..) of to_task :499: the reference skips the transcript                     */
#define V2P_4B_UNSUPPORTED   2   /* panic!("Instruction .. is not supported") :305,:479                             */
#define V2P_4B_ARITHMETIC    3   /* usize underflow in a task length (a debug build panics, e.g. catch_unwind :533) */
#define V2P_4B_CAPACITY      4   /* output arrays too small                                                         */

/* Tasks carry offsets relative to the transcript (step 5 / v2p_batch_add_transcript rebases them).
 * A transcript with a '0' or 'U' instruction, or none, yields the empty GIR: 0 tasks, res_len 0. */
int v2p_transcript_g_rep(const v2p_instruction* ins, uint64_t n_ins, uint64_t ref_len,
                         uint8_t* code, uint64_t* start_pos, uint64_t* length, uint64_t* start_pos_res,
                         uint64_t cap_tasks, uint64_t* n_tasks,
                         uint8_t* alt, uint64_t cap_alt, uint64_t* n_alt, uint64_t* res_len);

/* The INSPECT_TXP validation at the end of get_g_rep (transcript_instructions.rs:386-421; on unless NO_TEST is exported,
 * cli.rs:337-368): every task must start where the previous one ended and the lengths must add up to the expected
 * result size.  Returns 0, or 1 (not contiguous at task *first_bad: panic :404) or 2 (size mismatch: panic :413). */
#define V2P_4B_INSPECT_OK            0
#define V2P_4B_INSPECT_NOT_CONTIGUOUS 1
#define V2P_4B_INSPECT_SIZE_MISMATCH  2
int v2p_inspect_transcript_tasks(const uint64_t* length, const uint64_t* start_pos_res, uint64_t n_tasks, uint64_t res_len,
                                 int64_t* first_bad);

#ifdef __cplusplus
}
#endif
#endif
