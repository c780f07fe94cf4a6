/* TEST INFRASTRUCTURE ONLY -- C restatement of the reference's BCSQ bitmask decode, used as the CPU baseline of
 * tools/decode_bench.py and as a second checker in tests/.  The product (vcf2prot_amd/) never links this file.
 *
 * Follows the reference's algorithm and threading (Engine::MT arm):
 *   vcf_ds.rs:126-190   get_patient_fields   records split at tabs, first nine columns dropped, transposed per proband
 *                                             (chunks of records in parallel, then concatenated)
 *   vcf_ds.rs:192-211   get_csq_per_patient  probands in parallel
 *   text_parser.rs:163-252, MaskDecoder.rs:33-153, vcf_ds.rs:213-329   per column: last ':' field -> words -> indices ->
 *                                             bounds -> SUP_TYPE filter
 * Strings are not copied (the reference clones every column and every consequence); that only flatters the CPU side.
 * Parity pinned: tests/test_frontend_oracle.py checks it against oracle/frontend_oracle.py, which is pinned on the
 * reference's unit vectors and on FASTA written by the reference binary.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    const char* text;
    const uint64_t* row_begin;
    const uint64_t* row_end;
    uint64_t n_rows, n_samples;
    const uint32_t* csq_begin;      /* [n_rows + 1] */
    const uint8_t* csq_supported;   /* [n consequences] */
} fe_input;

enum { FE_OK = 0, FE_NEGATIVE = 1, FE_PARSE = 2, FE_INDEX = 3, FE_COLUMNS = 4, FE_NOMEM = 5 };

typedef struct { uint32_t* v; uint64_t n, cap; } vec32;
static int push(vec32* a, uint32_t x)
{
    if (a->n == a->cap) {
        uint64_t c = a->cap ? a->cap * 2 : 64;
        uint32_t* p = (uint32_t*)realloc(a->v, c * sizeof(uint32_t));
        if (!p) return -1;
        a->v = p; a->cap = c;
    }
    a->v[a->n++] = x;
    return 0;
}

/* Rust from_str: optional sign, digits; returns 0 on Err */
static int parse_num(const char* b, const char* e, int* neg, uint64_t* val)
{
    *neg = 0;
    if (b < e && (*b == '+' || *b == '-')) { *neg = (*b == '-'); ++b; }
    if (b >= e) return 0;
    uint64_t v = 0;
    for (const char* q = b; q < e; ++q) {
        if (*q < '0' || *q > '9') return 0;
        v = v * 10 + (uint64_t)(*q - '0');
        if (v > (1ull << 40)) v = 1ull << 40;
    }
    *val = v;
    return 1;
}

/* parse_fields + from_string of one element; returns the word, *err on abort */
static uint32_t single_word(const char* b, const char* e, int* err)
{
    int neg; uint64_t v;
    if (!parse_num(b, e, &neg, &v)) return 0;
    if (neg) {
        if (v > (1ull << 31)) return 0;
        *err = v ? FE_NEGATIVE : FE_PARSE;
        return 0;
    }
    return v <= 0x7FFFFFFFull ? (uint32_t)v : 0;
}

/* one column of one record: pushes consequence ids onto h[0] / h[1]; returns FE_* */
static int decode_column(const char* fb, const char* fe, uint32_t c0, uint32_t n_csq, const uint8_t* sup, vec32* h)
{
    const char* colon = NULL;
    for (const char* q = fe; q > fb; --q) if (q[-1] == ':') { colon = q - 1; break; }
    if (!colon) return FE_OK;
    const char* s = colon + 1;
    if (s == fe || (fe - s == 1 && *s == '.')) return FE_OK;
    uint32_t words[64];
    uint32_t n_words = 0;
    int err = 0, has_minus = 0;
    uint32_t n_el = 1, kept = 0;
    const char* es = s;
    const char* first_end = fe;
    for (const char* q = s; q < fe; ++q) {
        if (*q == ',') {
            if (!(q - es == 1 && *es == '0')) kept = n_el;
            if (n_el == 1) first_end = q;
            ++n_el; es = q + 1;
        } else if (*q == '-') has_minus = 1;
    }
    if (!(fe - es == 1 && *es == '0')) kept = n_el;
    int multi = 0;
    if (n_el == 1) words[n_words++] = single_word(s, fe, &err);
    else if (kept == 0) return FE_OK;
    else if (has_minus) return FE_NEGATIVE;
    else if (kept == 1) words[n_words++] = single_word(s, first_end, &err);
    else {
        multi = 1;
        const char* eb = s;
        uint32_t k = 0;
        for (const char* q = s; q <= fe && k < kept; ++q) {
            if (q == fe || *q == ',') {
                int neg; uint64_t v;
                if (!parse_num(eb, q, &neg, &v) || neg || v > 0xFFFFFFFFull) return FE_PARSE;
                if (n_words == 64) return FE_PARSE;      /* more than 960 consequences in one record: not restated */
                words[n_words++] = (uint32_t)v;
                ++k; eb = q + 1;
            }
        }
    }
    if (err) return err;
    /* get_indices (MaskDecoder.rs:95-153) + extract_effects' indexing (vcf_ds.rs:311-327) */
    for (int hap = 0; hap < 2; ++hap)
        for (uint32_t k = 0; k < n_words; ++k) {
            uint32_t m = words[k], idx = 0;
            while (m) {
                if ((m >> hap) & 1u) {
                    const uint32_t i = (multi ? 15u * k : 0u) + idx;
                    if (i >= n_csq) return FE_INDEX;
                }
                m >>= 2; ++idx;
            }
        }
    for (int hap = 0; hap < 2; ++hap)
        for (uint32_t k = 0; k < n_words; ++k) {
            uint32_t m = words[k], idx = 0;
            while (m) {
                if ((m >> hap) & 1u) {
                    const uint32_t i = (multi ? 15u * k : 0u) + idx;
                    if (sup[c0 + i] && push(&h[hap], c0 + i)) return FE_NOMEM;
                }
                m >>= 2; ++idx;
            }
        }
    return FE_OK;
}

typedef struct {
    const fe_input* in;
    uint64_t* field_begin;   /* [n_samples][n_rows] transposed column starts (get_patient_fields) */
    uint32_t* field_len;
    vec32* lists;            /* [2 * n_samples] */
    int n_threads, tid, phase;
    int err; int64_t err_field;
} fe_job;

static void* worker(void* p)
{
    fe_job* j = (fe_job*)p;
    const fe_input* in = j->in;
    const uint64_t R = in->n_rows, S = in->n_samples;
    if (j->phase == 0) {
        /* vcf_ds.rs:151-181: each thread splits a chunk of records */
        const uint64_t r0 = R * (uint64_t)j->tid / (uint64_t)j->n_threads, r1 = R * (uint64_t)(j->tid + 1) / (uint64_t)j->n_threads;
        for (uint64_t r = r0; r < r1; ++r) {
            const char* b = in->text + in->row_begin[r];
            const char* e = in->text + in->row_end[r];
            uint64_t s = 0;
            const char* fb = b;
            for (;;) {
                const char* t = (const char*)memchr(fb, '\t', (size_t)(e - fb));
                const char* fe = t ? t : e;
                if (s >= S) { j->err = FE_COLUMNS; j->err_field = (int64_t)(r * S + S - 1); return NULL; }
                j->field_begin[s * R + r] = (uint64_t)(fb - in->text);
                j->field_len[s * R + r] = (uint32_t)(fe - fb);
                ++s;
                if (!t) break;
                fb = t + 1;
            }
            if (s != S) { j->err = FE_COLUMNS; j->err_field = (int64_t)(r * S + s); return NULL; }
        }
    } else {
        /* vcf_ds.rs:205-209: probands in parallel, records in order */
        for (uint64_t s = (uint64_t)j->tid; s < S; s += (uint64_t)j->n_threads) {
            for (uint64_t r = 0; r < R; ++r) {
                const char* fb = in->text + j->field_begin[s * R + r];
                const int rc = decode_column(fb, fb + j->field_len[s * R + r], in->csq_begin[r], in->csq_begin[r + 1] - in->csq_begin[r],
                                             in->csq_supported, &j->lists[2 * s]);
                if (rc) { if (!j->err) { j->err = rc; j->err_field = (int64_t)(r * S + s); } return NULL; }
            }
        }
    }
    return NULL;
}

/* Returns FE_*; on success *ids_out is malloc'ed (caller frees with fe_free) and hap_begin[2S+1] is filled. */
int fe_decode(const fe_input* in, int n_threads, uint64_t* hap_begin, uint32_t** ids_out, int64_t* err_field)
{
    const uint64_t R = in->n_rows, S = in->n_samples;
    if (n_threads < 1) n_threads = 1;
    uint64_t* fbeg = (uint64_t*)malloc(R * S * sizeof(uint64_t));
    uint32_t* flen = (uint32_t*)malloc(R * S * sizeof(uint32_t));
    vec32* lists = (vec32*)calloc(2 * S, sizeof(vec32));
    fe_job* jobs = (fe_job*)calloc((size_t)n_threads, sizeof(fe_job));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    int rc = FE_OK;
    if (!fbeg || !flen || !lists || !jobs || !th) rc = FE_NOMEM;
    for (int phase = 0; phase < 2 && rc == FE_OK; ++phase) {
        for (int t = 0; t < n_threads; ++t) {
            jobs[t].in = in; jobs[t].field_begin = fbeg; jobs[t].field_len = flen; jobs[t].lists = lists;
            jobs[t].n_threads = n_threads; jobs[t].tid = t; jobs[t].phase = phase; jobs[t].err = 0; jobs[t].err_field = -1;
            pthread_create(&th[t], NULL, worker, &jobs[t]);
        }
        for (int t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
        for (int t = 0; t < n_threads; ++t)
            if (jobs[t].err && (rc == FE_OK || jobs[t].err_field < *err_field)) { rc = jobs[t].err; if (err_field) *err_field = jobs[t].err_field; }
    }
    if (rc == FE_OK) {
        hap_begin[0] = 0;
        for (uint64_t h = 0; h < 2 * S; ++h) hap_begin[h + 1] = hap_begin[h] + lists[h].n;
        uint32_t* ids = (uint32_t*)malloc((hap_begin[2 * S] + 1) * sizeof(uint32_t));
        if (!ids) rc = FE_NOMEM;
        else {
            for (uint64_t h = 0; h < 2 * S; ++h) if (lists[h].n) memcpy(ids + hap_begin[h], lists[h].v, lists[h].n * sizeof(uint32_t));
            *ids_out = ids;
        }
    }
    if (lists) for (uint64_t h = 0; h < 2 * S; ++h) free(lists[h].v);
    free(lists); free(fbeg); free(flen); free(jobs); free(th);
    return rc;
}

void fe_free(void* p) { free(p); }
