/*
 * cu2rec_oracle.c -- CPU restatement of the cu2rec hot path.  TEST INFRASTRUCTURE ONLY
 * (see cu2rec_oracle.h for the rules and the pinning status).
 *
 * Build: gcc -std=c99 -O3 -ffp-contract=off -fno-fast-math (see oracle/Makefile).  No FMA is
 * ever formed implicitly; the only fused operations are the explicit fmaf() calls of the
 * TREE16 dot order, which mirrors the HIP kernels.
 */
#define _POSIX_C_SOURCE 200809L
#include "cu2rec_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ================================================================== config */

void orc_config_default(orc_config *c) { /* config.h:23-51 */
    c->cur_iterations = 0;
    c->total_iterations = 5000;
    c->n_factors = 50;
    c->learning_rate = 0.01f;
    c->seed = 42;
    c->P_reg = c->Q_reg = c->user_bias_reg = c->item_bias_reg = 0.02f;
    c->is_train = 1;
    c->n_threads = 32;
    c->check_error = 500;
    c->patience = 2;
    c->learning_rate_decay = 0.2f;
}

int orc_config_read(const char *path, orc_config *c) { /* config.cu:7-13 */
    FILE *fp = fopen(path, "r");
    if (!fp) return -1;
    int n = fscanf(fp, "%d %d %d %f %d %f %f %f %f", &c->cur_iterations, &c->total_iterations,
                   &c->n_factors, &c->learning_rate, &c->seed, &c->P_reg, &c->Q_reg,
                   &c->user_bias_reg, &c->item_bias_reg);
    fclose(fp);
    return n == 9 ? 0 : -2;
}

int orc_config_write(const char *path, const orc_config *c) { /* config.cu:15-22 */
    FILE *fp = fopen(path, "w");
    if (!fp) return -1;
    fprintf(fp, "%d %d %d %g %d %g %g %g %g\n", c->cur_iterations, c->total_iterations, c->n_factors,
            c->learning_rate, c->seed, c->P_reg, c->Q_reg, c->user_bias_reg, c->item_bias_reg);
    fclose(fp);
    return 0;
}

/* ================================================================== CSV + CSR */

int orc_read_csv(const char *path, orc_ratings *out) { /* util.cu:17-45 */
    memset(out, 0, sizeof(*out));
    FILE *fp = fopen(path, "r");
    if (!fp) return -1;
    /* ratingsFile.ignore(1000, '\n') */
    for (int k = 0; k < 1000; ++k) {
        int ch = fgetc(fp);
        if (ch == EOF || ch == '\n') break;
    }
    int cap = 1024;
    out->user = (int *)malloc(sizeof(int) * cap);
    out->item = (int *)malloc(sizeof(int) * cap);
    out->rating = (float *)malloc(sizeof(float) * cap);
    int max_row = 0, max_col = 0;
    double sum = 0.0;
    int u, it;
    char d1, d2;
    float r;
    /* operator>> skips leading whitespace before every field, the char fields included */
    while (fscanf(fp, " %d %c %d %c %f", &u, &d1, &it, &d2, &r) == 5) {
        if (out->n == cap) {
            cap *= 2;
            out->user = (int *)realloc(out->user, sizeof(int) * cap);
            out->item = (int *)realloc(out->item, sizeof(int) * cap);
            out->rating = (float *)realloc(out->rating, sizeof(float) * cap);
        }
        out->user[out->n] = u - 1;
        out->item[out->n] = it - 1;
        out->rating[out->n] = r;
        out->n++;
        if (u > max_row) max_row = u;
        if (it > max_col) max_col = it;
        sum += r;
    }
    fclose(fp);
    out->rows = max_row;
    out->cols = max_col;
    out->global_bias = (float)(sum / (1.0 * out->n));
    return 0;
}

void orc_ratings_free(orc_ratings *r) {
    free(r->user);
    free(r->item);
    free(r->rating);
    memset(r, 0, sizeof(*r));
}

int orc_build_csr(const orc_ratings *r, int rows, int *indptr, int *indices, float *data) {
    /* util.cu:152-179: push i once for every user id stepped over, then nnz at the end */
    int last_user = -1, filled = 0;
    for (int i = 0; i < r->n; ++i) {
        int u = r->user[i];
        if (u < last_user) return -1; /* reference would spin until int overflow */
        while (last_user != u) {
            if (filled > rows) return -1;
            indptr[filled++] = i;
            last_user++;
        }
        indices[i] = r->item[i];
        data[i] = r->rating[i];
    }
    int rc = 0;
    if (filled > rows) return -1;
    if (filled < rows) rc = -2; /* trailing users without ratings */
    while (filled <= rows) indptr[filled++] = r->n;
    return rc;
}

/* ================================================================== mt19937 + normal */

typedef struct { uint32_t mt[624]; int idx; } orc_mt;

static void mt_seed(orc_mt *s, uint32_t seed) {
    s->mt[0] = seed;
    for (int i = 1; i < 624; ++i) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
    s->idx = 624;
}

static uint32_t mt_next(orc_mt *s) {
    if (s->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (s->mt[i] & 0x80000000u) | (s->mt[(i + 1) % 624] & 0x7fffffffu);
            uint32_t v = s->mt[(i + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            s->mt[i] = v;
        }
        s->idx = 0;
    }
    uint32_t y = s->mt[s->idx++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* libstdc++ std::generate_canonical<float, 24>(mt19937): one 32-bit draw, float(draw) / 2^32,
 * clamped below 1 (bits/random.tcc). */
static float mt_canonical(orc_mt *s) {
    float sum = (float)mt_next(s);
    float ret = sum / 4294967296.0f;
    if (ret >= 1.0f) ret = nextafterf(1.0f, 0.0f);
    return ret;
}

void orc_normal_fill(float *out, size_t size, int n_factors, float mean, float stddev, int seed) {
    /* util.cu:124-132; libstdc++ normal_distribution<float>::operator() (Marsaglia polar) */
    orc_mt g;
    mt_seed(&g, (uint32_t)seed);
    const float sd = stddev / n_factors;
    int saved_ok = 0;
    float saved = 0.f;
    for (size_t i = 0; i < size; ++i) {
        float ret;
        if (saved_ok) {
            saved_ok = 0;
            ret = saved;
        } else {
            float x, y, r2;
            do {
                x = 2.0f * mt_canonical(&g) - 1.0f;
                y = 2.0f * mt_canonical(&g) - 1.0f;
                r2 = x * x + y * y;
            } while (r2 > 1.0f || r2 == 0.0f);
            const float mult = sqrtf(-2 * logf(r2) / r2);
            saved = x * mult;
            saved_ok = 1;
            ret = y * mult;
        }
        out[i] = ret * sd + mean;
    }
}

/* ================================================================== Philox sampler */

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t m0 = (uint64_t)0xD2511F53u * c0;
        uint64_t m1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(m1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)m1;
        uint32_t n2 = (uint32_t)(m0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)m0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

uint32_t orc_draw(uint64_t seed, uint64_t user, uint64_t iteration) {
    uint32_t ctr[4] = {(uint32_t)iteration, (uint32_t)(iteration >> 32), (uint32_t)user, (uint32_t)(user >> 32)};
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    uint32_t out[4];
    orc_philox4x32_10(ctr, key, out);
    return out[0];
}

float orc_uniform(uint32_t x) {
    /* power-of-two scaling is exact, so this equals the fused form bit for bit */
    const float inv = 2.3283064365386963e-10f; /* 2^-32 */
    return inv + (float)x * inv;
}

/* Tier 3 of the parity protocol only (tests/test_oracle_golden.py): the reference CPU twin draws std::uniform_int_distribution(low, high)
 * -- INCLUSIVE of high (mf_sequential.cu:111), so with probability 1 / (n + 1) a user trains on the NEXT user's first rating (the
 * last user: one past the array, clamped here).  Never on in the product's parity tests: the half-open range of sgd.cu:37 is the
 * sampler everywhere else. */
static int g_inclusive_range = 0;
static int g_sample_limit = 0;
void orc_set_inclusive_range(int on, int nnz) {
    g_inclusive_range = on;
    g_sample_limit = nnz;
}

int orc_sample(uint64_t seed, uint64_t user, uint64_t iteration, int low, int high) {
    float u = orc_uniform(orc_draw(seed, user, iteration));
    if (g_inclusive_range) {
        int y = (int)ceilf(u * (float)(high - low + 1)) - 1 + low; /* mf_sequential.cu:111 */
        return y < g_sample_limit ? y : g_sample_limit - 1;
    }
    return (int)ceilf(u * (float)(high - low)) - 1 + low; /* sgd.cu:37 */
}

/* ================================================================== prediction */

static float dot_tree16(int f, const float *p, const float *q) {
    float lane[16];
    const int nslots = (f + 3) / 4;
    const int per_lane = (nslots + 15) / 16; /* slots per lane in the kernels' layout (zero padded) */
    for (int l = 0; l < 16; ++l) {
        /* lane total: slot partials (4-term fmaf chains from +0) added in slot order */
        float acc = 0.0f;
        for (int j = 0; j < (per_lane > 0 ? per_lane : 1); ++j) {
            const int s = l + 16 * j;
            float part = 0.0f;
            for (int c = 0; c < 4; ++c) {
                const int e = 4 * s + c;
                const float qv = e < f ? q[e] : 0.0f, pv = e < f ? p[e] : 0.0f;
                part = fmaf(qv, pv, part);
            }
            acc = j == 0 ? part : acc + part;
        }
        lane[l] = acc;
    }
    for (int m = 1; m < 16; m <<= 1) {
        float nxt[16];
        for (int l = 0; l < 16; ++l) nxt[l] = lane[l] + lane[l ^ m];
        memcpy(lane, nxt, sizeof(lane));
    }
    return lane[0];
}

float orc_predict(int f, const float *p, const float *q, float ub, float ib, float gb, int dot_order) {
    if (dot_order == ORC_DOT_TREE16) {
        float base = (gb + ub) + ib;
        return base + dot_tree16(f, p, q);
    }
    float pred = gb + ub + ib; /* util.cu:200 */
    for (int k = 0; k < f; ++k) pred += q[k] * p[k]; /* util.cu:201-202 */
    return pred;
}

/* ================================================================== SGD */

void orc_sgd_one(const int *indptr, const int *indices, const float *data, int x,
                 float *P, float *Q, float *user_bias, float *item_bias, float global_bias,
                 const orc_hyper *h, int f, uint64_t seed, uint64_t it, int dot_order, int update_items) {
    const int low = indptr[x], high = indptr[x + 1];
    if (low == high) return; /* mf_sequential.cu:108 */
    const int y_i = orc_sample(seed, (uint64_t)x, it, low, high);
    const int y = indices[y_i];
    float *p = &P[(size_t)x * f];
    float *q = &Q[(size_t)y * f];
    const float ub = user_bias[x], ib = item_bias[y];
    const float err = data[y_i] - orc_predict(f, p, q, ub, ib, global_bias, dot_order); /* :122-126 */
    const float lr = h->learning_rate;
    for (int k = 0; k < f; ++k) { /* :129-137 */
        const float p_old = p[k], q_old = q[k];
        p[k] = p_old + lr * (err * q_old - h->P_reg * p_old);
        if (update_items) q[k] = q_old + lr * (err * p_old - h->Q_reg * q_old);
    }
    user_bias[x] = ub + lr * (err - h->user_bias_reg * ub); /* :140 */
    if (update_items) item_bias[y] = ib + lr * (err - h->item_bias_reg * ib); /* :141 */
}

void orc_sgd_iterations(const int *indptr, const int *indices, const float *data, int rows,
                        float *P, float *Q, float *user_bias, float *item_bias, float global_bias,
                        const orc_hyper *h, int f, uint64_t seed, uint64_t iter0, int n_iters,
                        int dot_order, int update_items) {
    for (int i = 0; i < n_iters; ++i)      /* mf_sequential.cu:102 */
        for (int x = 0; x < rows; ++x)     /* :104 */
            orc_sgd_one(indptr, indices, data, x, P, Q, user_bias, item_bias, global_bias, h, f, seed,
                        iter0 + (uint64_t)i, dot_order, update_items);
}

void orc_pingpong_swap(float *Q, float *Q_target, float *item_bias, float *item_bias_target, int cols, int f) {
    /* training.cu:164-165 swaps pointers; the caller's arrays keep their roles here, so exchange the contents */
    for (size_t k = 0; k < (size_t)cols * f; ++k) {
        const float t = Q[k];
        Q[k] = Q_target[k];
        Q_target[k] = t;
    }
    for (int y = 0; y < cols; ++y) {
        const float t = item_bias[y];
        item_bias[y] = item_bias_target[y];
        item_bias_target[y] = t;
    }
}

void orc_sgd_pingpong_iterations(const int *indptr, const int *indices, const float *data, int rows, int cols,
                                 float *P, float *Q, float *Q_target, float *user_bias, float *item_bias,
                                 float *item_bias_target, float global_bias, const orc_hyper *h, int f, uint64_t seed,
                                 uint64_t iter0, int n_iters, int dot_order, int update_items, int swap_last) {
    unsigned char *item_is_updated = (unsigned char *)malloc(cols > 0 ? (size_t)cols : 1); /* training.cu:41-43 */
    const float lr = h->learning_rate;
    for (int i = 0; i < n_iters; ++i) {
        const uint64_t it = iter0 + (uint64_t)i;
        const int start_user = rows > 0 ? (int)((250u * it) % (uint64_t)rows) : 0; /* training.cu:97-98,115 */
        memset(item_is_updated, 0, cols > 0 ? (size_t)cols : 1);                   /* training.cu:168 */
        for (int gid = 0; gid < rows; ++gid) {                                      /* one thread per user, sgd.cu:27 */
            const int x = (gid + start_user) % rows;
            const int low = indptr[x], high = indptr[x + 1];
            if (low == high) continue; /* sgd.cu:34 */
            const int y_i = orc_sample(seed, (uint64_t)x, it, low, high); /* sgd.cu:36-37 */
            const int y = indices[y_i];
            float *p = &P[(size_t)x * f];
            const float *q = &Q[(size_t)y * f];
            const float ub = user_bias[x], ib = item_bias[y];
            const float err = data[y_i] - orc_predict(f, p, q, ub, ib, global_bias, dot_order); /* sgd.cu:45 */
            const int early_bird = !item_is_updated[y]; /* sgd.cu:49-50 */
            item_is_updated[y] = 1;
            for (int k = 0; k < f; ++k) { /* sgd.cu:53-64 */
                const float p_old = p[k], q_old = q[k];
                p[k] = p_old + lr * (err * q_old - h->P_reg * p_old);
                if (update_items && early_bird) Q_target[(size_t)y * f + k] = q_old + lr * (err * p_old - h->Q_reg * q_old);
            }
            user_bias[x] = ub + lr * (err - h->user_bias_reg * ub); /* sgd.cu:67 */
            if (update_items && early_bird) item_bias_target[y] = ib + lr * (err - h->item_bias_reg * ib); /* sgd.cu:71 */
        }
        if (i + 1 < n_iters || swap_last) orc_pingpong_swap(Q, Q_target, item_bias, item_bias_target, cols, f);
    }
    free(item_is_updated);
}

int orc_sgd_iterations_parallel(const int *indptr, const int *indices, const float *data, int rows,
                                float *P, float *Q, float *user_bias, float *item_bias, float global_bias,
                                const orc_hyper *h, int f, uint64_t seed, uint64_t iter0, int n_iters,
                                int dot_order, int n_threads) {
    int used = 1;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    for (int i = 0; i < n_iters; ++i) {
#ifdef _OPENMP
#pragma omp parallel
        {
#pragma omp single
            used = omp_get_num_threads();
#pragma omp for schedule(static)
            for (int x = 0; x < rows; ++x)
                orc_sgd_one(indptr, indices, data, x, P, Q, user_bias, item_bias, global_bias, h, f, seed,
                            iter0 + (uint64_t)i, dot_order, 1);
        }
#else
        (void)n_threads;
        for (int x = 0; x < rows; ++x)
            orc_sgd_one(indptr, indices, data, x, P, Q, user_bias, item_bias, global_bias, h, f, seed, iter0 + (uint64_t)i,
                        dot_order, 1);
#endif
    }
    return used;
}

/* ================================================================== loss */

void orc_loss(const int *indptr, const int *indices, const float *data, int rows, int nnz,
              const float *P, const float *Q, const float *user_bias, const float *item_bias,
              float global_bias, int f, int dot_order, int acc, float *errors_out,
              double *sum_abs, double *sum_sq, float *mae, float *rmse) {
    double dabs = 0.0, dsq = 0.0;
    float fabs_acc = 0.0f, fsq_acc = 0.0f;
    for (int x = 0; x < rows; ++x) { /* loss.cu:24-34, mf_sequential.cu:152-171 */
        const float *p = &P[(size_t)x * f];
        const float ub = user_bias[x];
        for (int k = indptr[x]; k < indptr[x + 1]; ++k) {
            const int item = indices[k];
            const float e = data[k] - orc_predict(f, p, &Q[(size_t)item * f], ub, item_bias[item], global_bias, dot_order);
            if (errors_out) errors_out[k] = e;
            if (acc == ORC_ACC_F32) {
                fabs_acc += fabsf(e);
                fsq_acc += e * e;
            } else {
                dabs += (double)fabsf(e);     /* loss.cu:70 abs(float) widened into double sdata */
                dsq += (double)e * (double)e; /* loss.cu:70 pow(double(e), 2) */
            }
        }
    }
    if (acc == ORC_ACC_F32) {
        /* mf_sequential.cu:173-174: float / float(size_t), sqrt(float) */
        const float n = (float)(size_t)nnz;
        if (mae) *mae = fabs_acc / n;
        if (rmse) *rmse = sqrtf(fsq_acc / n);
        if (sum_abs) *sum_abs = fabs_acc;
        if (sum_sq) *sum_sq = fsq_acc;
    } else {
        if (mae) *mae = (float)(dabs / nnz);          /* loss.cu:189 */
        if (rmse) *rmse = (float)sqrt(dsq / nnz);     /* loss.cu:189 */
        if (sum_abs) *sum_abs = dabs;
        if (sum_sq) *sum_sq = dsq;
    }
}

void orc_error_metrics(const float *errors, int n, float *mae, float *rmse) {
    double a = 0.0, s = 0.0;
    for (int i = 0; i < n; ++i) {
        a += (double)fabsf(errors[i]);
        s += (double)errors[i] * (double)errors[i];
    }
    *mae = (float)(a / n);
    *rmse = (float)sqrt(s / n);
}

/* ================================================================== training schedule */

int orc_train(const int *tr_indptr, const int *tr_indices, const float *tr_data, int rows, int cols, int tr_nnz,
              const int *te_indptr, const int *te_indices, const float *te_data, int te_rows, int te_nnz,
              orc_config *cfg, float *P, float *Q, float *user_bias, float *item_bias, float global_bias,
              int dot_order, int acc, int schedule, orc_log_entry *log, int log_cap) {
    (void)cols;
    const int f = cfg->n_factors;
    int n_log = 0;
    float validation_rmse = FLT_MAX, last_validation_rmse; /* training.cu:102 */
    int current_patience = (int)cfg->patience;             /* training.cu:103 */
    const int total = cfg->total_iterations;
    const uint64_t iter_base = (uint64_t)cfg->cur_iterations;
    for (int i = 0; i < total; ++i) {
        orc_hyper h = {cfg->learning_rate, cfg->P_reg, cfg->Q_reg, cfg->user_bias_reg, cfg->item_bias_reg};
        orc_sgd_iterations(tr_indptr, tr_indices, tr_data, rows, P, Q, user_bias, item_bias, global_bias, &h, f,
                           (uint64_t)(uint32_t)cfg->seed, iter_base + (uint64_t)i, 1, dot_order, cfg->is_train);
        if ((i + 1) % cfg->check_error == 0 || i == 0 || (i + 1) % total == 0) { /* training.cu:118 */
            orc_log_entry e;
            e.iteration = i + 1;
            orc_loss(tr_indptr, tr_indices, tr_data, rows, tr_nnz, P, Q, user_bias, item_bias, global_bias, f,
                     dot_order, acc, NULL, NULL, NULL, &e.train_mae, &e.train_rmse);
            last_validation_rmse = validation_rmse; /* training.cu:129 */
            orc_loss(te_indptr, te_indices, te_data, te_rows, te_nnz, P, Q, user_bias, item_bias, global_bias, f,
                     dot_order, acc, NULL, NULL, NULL, &e.test_mae, &e.test_rmse);
            validation_rmse = e.test_rmse;
            if (schedule == ORC_SCHED_PATIENCE) { /* training.cu:146-155 */
                if (last_validation_rmse < validation_rmse) current_patience--;
                if (current_patience <= 0) {
                    current_patience = (int)cfg->patience;
                    cfg->learning_rate *= cfg->learning_rate_decay;
                }
            }
            e.lr = cfg->learning_rate;
            if (n_log < log_cap) log[n_log++] = e;
        }
        cfg->cur_iterations += 1; /* training.cu:170 */
    }
    return n_log;
}

/* ================================================================== output */

int orc_write_csv(const char *path, const float *data, int rows, int cols) { /* util.cu:86-97 */
    FILE *fp = fopen(path, "w");
    if (!fp) return -1;
    for (int i = 0; i < rows; ++i) {
        for (int j = 0; j < cols - 1; ++j) fprintf(fp, "%f,", data[(size_t)i * cols + j]);
        fprintf(fp, "%f", data[(size_t)i * cols + cols - 1]);
        fprintf(fp, "\n");
    }
    fclose(fp);
    return 0;
}

double orc_now_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
