/* oracle/gsa_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * C API of the CPU restatement of GSAlign's hot path (see gsa_oracle.cpp).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load libgsa_oracle.so; the product (gsalign_amd/) never links or calls it.
 *
 * The getters deliberately have the same shape as the gsref_* functions that
 * oracle/ref_glue.cpp exports around the real reference, so one Python helper
 * can read either side and the parity tests compare like with like.
 */
#ifndef GSA_ORACLE_H
#define GSA_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_ctx ora_ctx;

/* hdr = {primary, L2[1], L2[2], L2[3], L2[4]} exactly as stored at the head of
 * the .bwt file; bwt = the interleaved Occ+BWT words that follow; sa = n_sa
 * values with sa[0] = (uint64_t)-1 (loader convention, bwt_index.cpp:40);
 * ref = 2G ASCII bytes (forward strand then reverse complement,
 * bwt_index.cpp:193-209); chr_len = forward lengths of the n_chr reference
 * sequences.  All arrays are copied. */
ora_ctx *ora_create(const uint64_t hdr[5], const uint32_t *bwt, uint64_t bwt_words,
                    const uint64_t *sa, uint64_t n_sa,
                    const char *ref, int64_t G, const int32_t *chr_len, int n_chr);
void ora_destroy(ora_ctx *);

/* tunables as parsed by the reference CLI (main.cpp:202-214,236-299,323) */
void ora_params(ora_ctx *, int slen, int ind, int clr, int alen, int idy, int sen, int one);
void ora_set_query(ora_ctx *, const char *seq, int len);
/* stage numbers: see ref_glue.cpp (1 = seeds+groups ... 8 = final blocks) */
int  ora_run_to(ora_ctx *, int stage);

long long ora_seed_count(ora_ctx *);
void ora_seeds(ora_ctx *, int *qpos, int *qlen, long long *rpos);
int  ora_group_count(ora_ctx *);
void ora_groups(ora_ctx *, int *beg, int *end);
int  ora_block_count(ora_ctx *);
long long ora_frag_total(ora_ctx *);
long long ora_aln_total(ora_ctx *);
void ora_block_meta(ora_ctx *, int *score, int *aln_len, int *bdup, int *nfrag, int *bdir, int *gpos, int *chr);
void ora_frags(ora_ctx *, int *bseed, int *qpos, int *qlen, long long *rpos, int *rlen, int *alnlen);
void ora_frag_aln(ora_ctx *, char *a1, char *a2);

/* function-level entry points */
int  ora_ksw2(const char *s1, int m, const char *s2, int n, char *out1, char *out2); /* out: m+n+1 bytes each */
int  ora_ksw2_ops(const char *s1, int m, const char *s2, int n, char *ops);           /* forward-order M/D/I, m+n bytes */
int  ora_gap_similarity(ora_ctx *, int q1, int q2, long long r1, long long r2);
int  ora_bwt_search(ora_ctx *, int start, int stop, int *len, long long *locs);
long long ora_bwt_sa(ora_ctx *, unsigned long long k);

/* event counters accumulated since ora_set_query:
 * [0] Occ blocks touched by seed extension  [1] LF steps in locate
 * [2] located hits                          [3] seeds written
 * [4] DP cells (sum m*n)                    [5] DP jobs
 * [6] sum (m+n) over DP jobs                [7] times the reference's undefined
 *                                               "whole group died" case was hit */
void ora_counters(ora_ctx *, uint64_t out[8]);

#ifdef __cplusplus
}
#endif
#endif
