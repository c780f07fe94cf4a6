/*
 * sir_oracle.c -- CPU restatement of vcf2prot's step-6 SIR executor.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP engine.
 * Nothing under vcf2prot_amd/ may include, link, dlopen or call it; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the
 * checker / the reported CPU baseline -- never as the thing shipped or measured
 * as "the engine".
 *
 * Parity status: PINNED.  The restatement is checked (tests/test_oracle_golden.py)
 * against (a) the reference's own known-answer tests -- task.rs:118-144,
 * gir.rs:172-196, transcript_instructions.rs:884-1594 -- and (b) Task vectors and
 * FASTA records produced by the reference's prebuilt CPU binary
 * (/root/reference/bins/Linux/vcf2prot v0.1.2) run in the build container by
 * oracle/make_golden.py; the harvested vectors live in tests/golden/.
 * The Rust sources cannot be compiled here (no cargo/rustc), so there is no
 * oracle/_ref build; see DESIGN.md "Oracle".
 *
 * What is restated (all paths relative to /root/reference/src/data_structures/InternalRep):
 *   task.rs:2-9      struct Task {exe_code:u8, start_pos, length, start_pos_res: usize}
 *   task.rs:38-50    Task::execute  (slice copy, panics on out-of-bounds)
 *   gir.rs:197-241   GIR::execute   (optional DEBUG_CPU_EXEC contiguity check, then
 *                                    the sequential task loop; Engine::GPU arm panics)
 *   haplotype_instruction.rs:78     result tape pre-filled with '.'
 *   parts/exec.rs:34-40             samples spread over a thread pool (Rayon)
 *
 * `char` in the reference is a 4-byte Unicode scalar; the faithful entry points
 * below therefore move uint32_t.  The *_u8 entry points are the "best-effort
 * CPU" flavour (1 byte per amino acid, memcpy) reported beside it.
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <stdlib.h>
#include <pthread.h>
#include <time.h>

/* task.rs:2-9.  Rust's default repr orders/pads this to 32 bytes; the field
 * order here is irrelevant to the semantics, the 32-byte AoS footprint is kept
 * so the CPU baseline moves the same descriptor bytes as the reference. */
typedef struct {
    uint64_t start_pos;      /* offset in the source tape                    */
    uint64_t length;         /* number of chars to copy                      */
    uint64_t start_pos_res;  /* offset in the result tape                    */
    uint8_t  exe_code;       /* 0 = reference tape, anything else = alt tape */
    uint8_t  _pad[7];
} sir_task;

/* panic-equivalent status codes (Rust slice indexing panics; task.rs:38-50) */
enum {
    SIR_OK = 0,
    SIR_PANIC_RES_OOB = 1,     /* results_tape[start_pos_res..end] out of range */
    SIR_PANIC_SRC_OOB = 2,     /* ref_tape/alt_tape[start_pos..end] out of range */
    SIR_PANIC_NOT_CONTIGUOUS = 3, /* gir.rs:208-226 DEBUG_CPU_EXEC predicate failed */
    SIR_PANIC_OVERFLOW = 4     /* start + length overflows usize (debug-build panic) */
};

size_t sir_sizeof_task(void) { return sizeof(sir_task); }

/* task.rs:38-50 */
int sir_task_execute(const sir_task *t,
                     uint32_t *res, uint64_t n_res,
                     const uint32_t *ref, uint64_t n_ref,
                     const uint32_t *alt, uint64_t n_alt)
{
    uint64_t end_res = t->start_pos_res + t->length;   /* task.rs:40 */
    uint64_t end_src = t->start_pos + t->length;       /* task.rs:41 */
    if (end_res < t->start_pos_res || end_src < t->start_pos) return SIR_PANIC_OVERFLOW;
    const uint32_t *src; uint64_t n_src;
    if (t->exe_code == 0) { src = ref; n_src = n_ref; }  /* task.rs:42-45 */
    else                  { src = alt; n_src = n_alt; }  /* task.rs:46-49 */
    /* Rust evaluates the destination slice first, then the source slice */
    if (end_res > n_res) return SIR_PANIC_RES_OOB;
    if (end_src > n_src) return SIR_PANIC_SRC_OOB;
    memcpy(res + t->start_pos_res, src + t->start_pos, (size_t)t->length * sizeof(uint32_t));
    return SIR_OK;
}

/* gir.rs:208-226: first idx >= 1 with start_pos_res[idx] != start_pos_res[idx-1] + length[idx-1];
 * -1 if the vector passes. */
int64_t sir_validate_contiguity(const sir_task *tasks, uint64_t n)
{
    for (uint64_t i = 1; i < n; ++i)
        if (tasks[i].start_pos_res != tasks[i - 1].start_pos_res + tasks[i - 1].length)
            return (int64_t)i;
    return -1;
}

/* haplotype_instruction.rs:78 */
void sir_fill_dots(uint32_t *res, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i) res[i] = (uint32_t)'.';
}

/* gir.rs:197-241, ST/MT arm.  On a panic-equivalent the index of the offending
 * task is stored in *bad_idx and the result tape holds whatever the tasks before
 * it wrote (as a Rust unwinding panic would leave it). */
int sir_gir_execute(const sir_task *tasks, uint64_t n_tasks,
                    const uint32_t *ref, uint64_t n_ref,
                    const uint32_t *alt, uint64_t n_alt,
                    uint32_t *res, uint64_t n_res,
                    int debug_cpu_exec, int64_t *bad_idx)
{
    if (bad_idx) *bad_idx = -1;
    if (debug_cpu_exec) {                                   /* gir.rs:203-229 */
        int64_t b = sir_validate_contiguity(tasks, n_tasks);
        if (b >= 0) { if (bad_idx) *bad_idx = b; return SIR_PANIC_NOT_CONTIGUOUS; }
    }
    for (uint64_t i = 0; i < n_tasks; ++i) {                /* gir.rs:233 */
        int rc = sir_task_execute(&tasks[i], res, n_res, ref, n_ref, alt, n_alt);
        if (rc != SIR_OK) { if (bad_idx) *bad_idx = (int64_t)i; return rc; }
    }
    return SIR_OK;
}

/* Convenience for callers that hold the SoA shape of gir.rs:283-299. */
int sir_gir_execute_soa(const uint8_t *code, const uint64_t *start_pos,
                        const uint64_t *length, const uint64_t *start_pos_res,
                        uint64_t n_tasks,
                        const uint32_t *ref, uint64_t n_ref,
                        const uint32_t *alt, uint64_t n_alt,
                        uint32_t *res, uint64_t n_res,
                        int debug_cpu_exec, int64_t *bad_idx)
{
    sir_task *t = (sir_task *)malloc((size_t)(n_tasks ? n_tasks : 1) * sizeof(sir_task));
    if (!t) return -1;
    for (uint64_t i = 0; i < n_tasks; ++i) {
        memset(&t[i], 0, sizeof(sir_task));
        t[i].exe_code = code[i]; t[i].start_pos = start_pos[i];
        t[i].length = length[i]; t[i].start_pos_res = start_pos_res[i];
    }
    int rc = sir_gir_execute(t, n_tasks, ref, n_ref, alt, n_alt, res, n_res, debug_cpu_exec, bad_idx);
    free(t);
    return rc;
}

/* ------------------------------------------------------------------------- *
 * 1-byte-per-residue flavour ("CPU-best" in BASELINE.md section 3).  Same     *
 * semantics, uint8_t tapes.  Used (a) as the honest upper bound of a CPU      *
 * engine and (b) to produce byte-level expectations for the device arena.     *
 * ------------------------------------------------------------------------- */
int sir_gir_execute_u8(const sir_task *tasks, uint64_t n_tasks,
                       const uint8_t *ref, uint64_t n_ref,
                       const uint8_t *alt, uint64_t n_alt,
                       uint8_t *res, uint64_t n_res, int64_t *bad_idx)
{
    if (bad_idx) *bad_idx = -1;
    for (uint64_t i = 0; i < n_tasks; ++i) {
        const sir_task *t = &tasks[i];
        uint64_t end_res = t->start_pos_res + t->length, end_src = t->start_pos + t->length;
        const uint8_t *src = t->exe_code == 0 ? ref : alt;
        uint64_t n_src = t->exe_code == 0 ? n_ref : n_alt;
        if (end_res < t->start_pos_res || end_src < t->start_pos) { if (bad_idx) *bad_idx = (int64_t)i; return SIR_PANIC_OVERFLOW; }
        if (end_res > n_res) { if (bad_idx) *bad_idx = (int64_t)i; return SIR_PANIC_RES_OOB; }
        if (end_src > n_src) { if (bad_idx) *bad_idx = (int64_t)i; return SIR_PANIC_SRC_OOB; }
        memcpy(res + t->start_pos_res, src + t->start_pos, (size_t)t->length);
    }
    return SIR_OK;
}

/* ------------------------------------------------------------------------- *
 * "Rayon-MT equivalent" driver (parts/exec.rs:34-40 + gir.rs:230-234):       *
 * haplotypes are independent jobs pulled from a shared counter by a pool of   *
 * worker threads; inside a haplotype the tasks run sequentially, exactly as   *
 * the reference's MT engine does.  Each job fills its result tape with '.'    *
 * (haplotype_instruction.rs:78) and then runs the task loop.                  *
 * ------------------------------------------------------------------------- */
typedef struct {
    const sir_task *tasks; uint64_t n_tasks;
    const void *ref; uint64_t n_ref;
    const void *alt; uint64_t n_alt;
    void *res; uint64_t n_res;
} sir_job;

size_t sir_sizeof_job(void) { return sizeof(sir_job); }

typedef struct {
    const sir_job *jobs; uint64_t n_jobs; int wide; int reps;
    uint64_t *next;                 /* one job counter per pass */
    int status; pthread_mutex_t mu; pthread_barrier_t bar;
} sir_pool;

static void sir_run_job(sir_pool *p, const sir_job *jb)
{
    int rc;
    if (p->wide) {
        sir_fill_dots((uint32_t *)jb->res, jb->n_res);
        rc = sir_gir_execute(jb->tasks, jb->n_tasks, (const uint32_t *)jb->ref, jb->n_ref,
                             (const uint32_t *)jb->alt, jb->n_alt, (uint32_t *)jb->res, jb->n_res, 0, NULL);
    } else {
        memset(jb->res, '.', (size_t)jb->n_res);
        rc = sir_gir_execute_u8(jb->tasks, jb->n_tasks, (const uint8_t *)jb->ref, jb->n_ref,
                                (const uint8_t *)jb->alt, jb->n_alt, (uint8_t *)jb->res, jb->n_res, NULL);
    }
    if (rc != SIR_OK) { pthread_mutex_lock(&p->mu); p->status = rc; pthread_mutex_unlock(&p->mu); }
}

/* One worker of the persistent pool: threads are created once per sir_mt_execute() call and pull jobs
 * pass after pass (a barrier between passes), as a Rayon pool keeps its workers across par_iter calls. */
static void *sir_worker(void *arg)
{
    sir_pool *p = (sir_pool *)arg;
    for (int r = 0; r < p->reps; ++r) {
        for (;;) {
            uint64_t j = __atomic_fetch_add(&p->next[r], 1, __ATOMIC_RELAXED);
            if (j >= p->n_jobs) break;
            sir_run_job(p, &p->jobs[j]);
        }
        pthread_barrier_wait(&p->bar);
    }
    return NULL;
}

/* Runs all jobs `reps` times on `n_threads` threads; returns wall seconds for
 * the whole run in *seconds (thread creation excluded: the clock starts when every worker
 * stands at the first barrier).  wide = 1: uint32_t tapes (reference-faithful);
 * wide = 0: uint8_t tapes (CPU-best). */
int sir_mt_execute(const sir_job *jobs, uint64_t n_jobs, int n_threads, int wide, int reps, double *seconds)
{
    if (n_threads < 1) n_threads = 1;
    if (reps < 1) reps = 1;
    sir_pool p; p.jobs = jobs; p.n_jobs = n_jobs; p.wide = wide; p.status = SIR_OK; p.reps = reps + 1;
    p.next = (uint64_t *)calloc((size_t)reps + 1, sizeof(uint64_t));
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    if (!p.next || !th) { free(p.next); free(th); return SIR_PANIC_OVERFLOW; }
    p.next[0] = n_jobs;             /* pass 0 is empty: it only lines the workers up at the barrier */
    pthread_mutex_init(&p.mu, NULL);
    pthread_barrier_init(&p.bar, NULL, (unsigned)n_threads + 1u);
    for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, sir_worker, &p);
    struct timespec t0, t1;
    pthread_barrier_wait(&p.bar);
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < reps; ++r) pthread_barrier_wait(&p.bar);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
    if (seconds) *seconds = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    free(th); free(p.next);
    pthread_barrier_destroy(&p.bar);
    pthread_mutex_destroy(&p.mu);
    return p.status;
}

/* Position-sensitive, order-independent digest of a byte range; the same
 * function is implemented on the device (csrc/stitch_kernels.hip, digest kernel)
 * so full-size runs can be compared haplotype by haplotype without moving the
 * arena over PCIe.  Round 5 definition (one multiplier per 8-byte word instead of
 * one per byte: the device kernel had become 8 x an execute):
 *     digest = sum_i (byte_i + 1) * 2^(8 * (i mod 8)) * mix(i div 8)   (mod 2^64),
 * i relative to the range start, mix = splitmix64 finaliser.  Word by word that is
 * sum_k mix(k) * (little-endian word k + 0x0101..01 over the bytes that exist). */
static inline uint64_t sir_mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
uint64_t sir_digest_u8(const uint8_t *p, uint64_t n)
{
    uint64_t s = 0;
    const uint64_t nw = n >> 3;
    for (uint64_t k = 0; k < nw; ++k) {
        uint64_t w;
        memcpy(&w, p + 8 * k, 8);              /* (little-endian host) */
        s += (w + 0x0101010101010101ull) * sir_mix64(k);
    }
    uint64_t t = 0;
    for (uint64_t i = 8 * nw; i < n; ++i) t += ((uint64_t)p[i] + 1ull) << (8 * (i & 7));
    if (n & 7) s += t * sir_mix64(nw);
    return s;
}
uint64_t sir_digest_u32(const uint32_t *p, uint64_t n)
{
    uint64_t s = 0;
    for (uint64_t i = 0; i < n; ++i) s += (((uint64_t)(p[i] & 0xFFu) + 1ull) << (8 * (i & 7))) * sir_mix64(i >> 3);
    return s;
}
