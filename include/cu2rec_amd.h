/*
 * cu2rec_amd.h -- C ABI of the MI355X-native cu2rec hot path (libcu2rec_amd.so).
 *
 * Plain C: pointers, sizes, PODs.  No torch / HIP types in any signature (streams travel as
 * void*, device buffers as raw pointers).  Every entry point returns CU2REC_OK (0) or a
 * negative cu2rec_status; cu2rec_last_error() gives the message of the calling thread's last
 * failure.  The reference reports failures by throwing std::runtime_error from CHECK_CUDA
 * (util.h:27-34) or by printing to stderr and returning an empty result (util.cu:41-44); the
 * C++ wrappers in cu2rec_amd/csrc/cu2rec.hpp turn a non-zero status back into that exception.
 *
 * Each declaration names the reference interface it replaces
 * (paths relative to nickgreenquist/cu2rec matrix_factorization/).
 *
 * Device data layout ("padded rows"): dense factor matrices are row-major float32 with a row
 * stride `ld` that is a multiple of 4 floats (16 B) and >= n_factors; the padding floats must
 * be zero (they then stay zero under every kernel here).  ld == n_factors is the reference's
 * own layout (matrix.h:21-28) whenever n_factors % 4 == 0.  Recommended for Q, and what cu2rec_model uses: a stride of
 * a multiple of 32 floats, i.e. item rows that are whole 128-byte lines -- several XCDs read and write them, and rows
 * that share a line cost coherence misses and partial-line write-backs.  CSR is exactly the reference's:
 * int32 indptr[rows+1], int32 indices[nnz], float32 data[nnz] (matrix.h:11-19).
 */
#ifndef CU2REC_AMD_H
#define CU2REC_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CU2REC_AMD_VERSION 100 /* 0.1.0 */

typedef enum cu2rec_status {
    CU2REC_OK = 0,
    CU2REC_EINVAL = -1,   /* bad argument (null pointer, bad shape, unsorted ratings, ...) */
    CU2REC_EIO = -2,      /* file cannot be opened / parsed */
    CU2REC_EHIP = -3,     /* a HIP runtime call failed (the reference's CHECK_CUDA throw) */
    CU2REC_ENODEVICE = -4,/* no usable GPU: the hot path never falls back to the CPU */
    CU2REC_ENOMEM = -5,
    CU2REC_EUNSUPPORTED = -6 /* e.g. n_factors above the compiled kernel range */
} cu2rec_status;

const char *cu2rec_last_error(void);
int cu2rec_version(void);
/* number of visible HIP devices, 0 if none (never fails) */
int cu2rec_device_count(void);
/* select the device used by subsequent calls of this thread (hipSetDevice) */
int cu2rec_set_device(int device);

/* ------------------------------------------------------------------------------------------
 * Config -- replaces config::Config (config.h:20-58) and its file format (config.cu:7-22).
 * The first nine fields are the file's nine whitespace separated fields, in file order.
 * The `__constant__` mirror (config.h:9-18, config.cu:24-48) has no equivalent: hyper-parameters
 * are passed to kernels by value (cu2rec_hyper below).
 * ---------------------------------------------------------------------------------------- */
typedef struct cu2rec_config {
    int cur_iterations;        /* config.h:23 */
    int total_iterations;      /* config.h:25  (default 5000) */
    int n_factors;             /* config.h:27  (default 50) */
    float learning_rate;       /* config.h:29  (default 0.01) */
    int seed;                  /* config.h:31  (default 42): sampler seed, training.cu:88 */
    float P_reg;               /* config.h:33  (default 0.02) */
    float Q_reg;               /* config.h:35 */
    float user_bias_reg;       /* config.h:37 */
    float item_bias_reg;       /* config.h:39 */
    int is_train;              /* config.h:41: 0 freezes Q and item_bias (honoured here; the
                                  reference never uploads it, config.cu:24-35) */
    int n_threads;             /* config.h:43 (default 32): kept for schema compatibility and
                                  printed; launch geometry is chosen by the library */
    int check_error;           /* config.h:45 (default 500) */
    float patience;            /* config.h:48 (default 2) */
    float learning_rate_decay; /* config.h:51 (default 0.2) */
} cu2rec_config;

int cu2rec_config_default(cu2rec_config *cfg);
int cu2rec_config_read(const char *path, cu2rec_config *cfg);        /* Config::read_config  config.cu:7-13 */
int cu2rec_config_write(const char *path, const cu2rec_config *cfg); /* Config::write_config config.cu:15-22 */
int cu2rec_config_print(const cu2rec_config *cfg);                   /* Config::print_config config.cu:50-64 (stdout) */

/* ------------------------------------------------------------------------------------------
 * Ratings file -> host CSR.  Replaces readCSV (util.cu:17-45) and the host half of
 * createSparseMatrix (util.cu:152-179).
 * ---------------------------------------------------------------------------------------- */
typedef struct cu2rec_ratings cu2rec_ratings; /* opaque, host memory */

/* "userId,itemId,rating" with one header line; ids 1-based, converted to 0-based.  rows/cols
 * = max id seen, global_bias = float(double sum / n).  A file that cannot be opened is
 * CU2REC_EIO (the reference prints "ERROR: The file isnt open." and returns nothing).
 * Files above ~1 MB whose records sit one per line are parsed on all host cores (CU2REC_READER_THREADS
 * overrides the count); anything else takes the sequential path, whose record grammar is the reference's
 * (`int char int char float` separated by arbitrary whitespace, first malformed record ends the file). */
int cu2rec_ratings_read_csv(const char *path, cu2rec_ratings **out);
/* Binary cache of a parsed file (header + three COO columns): loading it replaces the text parse. */
int cu2rec_ratings_save_binary(const cu2rec_ratings *r, const char *path);
int cu2rec_ratings_load_binary(const char *path, cu2rec_ratings **out);
int cu2rec_ratings_info(const cu2rec_ratings *r, int *n, int *rows, int *cols, float *global_bias);
/* raw COO views (length n), valid until cu2rec_ratings_free */
int cu2rec_ratings_view(const cu2rec_ratings *r, const int **user, const int **item, const float **rating);
void cu2rec_ratings_free(cu2rec_ratings *r);
/* Build the CSR into caller memory: indptr[rows+1], indices[n], data[n].  Users without
 * ratings get repeated pointers (tests/test_util.cu:146-189).  Ratings must be grouped by
 * ascending user (CU2REC_EINVAL otherwise -- the reference would not terminate); item order
 * inside a user is kept as in the file.  rows must be >= the largest user id. */
int cu2rec_csr_build(const cu2rec_ratings *r, int rows, int *indptr, int *indices, float *data);

/* initialize_normal_array (util.cu:124-144): std::mt19937(seed) +
 * std::normal_distribution<float>(mean, stddev / n_factors), sequential fill. */
int cu2rec_init_normal(float *out, size_t size, int n_factors, float mean, float stddev, int seed);

/* writeCSV / writeToFile (util.cu:86-103): "%f" comma joined rows;
 * file name <parent_dir>/<base>_f<factors>_<component>.csv */
int cu2rec_write_csv(const char *path, const float *data, int rows, int cols);
int cu2rec_write_component(const char *parent_dir, const char *base, const char *component, const float *data,
                           int rows, int cols, int factors);
/* read_array (util.cu:52-81): comma separated float rows; *out is malloc'ed -> cu2rec_free */
int cu2rec_read_array(const char *path, float **out, int *rows, int *cols);
void cu2rec_free(void *p);

/* ------------------------------------------------------------------------------------------
 * Sampler.  Replaces initCurand + curand_uniform (sgd.cu:11-16,36-37).  Counter based, no
 * state array, no init kernel: the draw of (user, iteration) is the first word of
 * rocRAND's Philox4x32-10 stream rocrand_init(seed, subsequence=user, offset=4*iteration),
 * mapped to (0,1] as rocrand_uniform does, then y_i = ceil(u * n) - 1 + low  (sgd.cu:37).
 * Host copy of the device function, for tests and for callers that need the schedule.
 * ---------------------------------------------------------------------------------------- */
uint32_t cu2rec_sampler_draw(uint64_t seed, uint64_t user, uint64_t iteration);
int cu2rec_sampler_index(uint64_t seed, uint64_t user, uint64_t iteration, int low, int high);

/* ------------------------------------------------------------------------------------------
 * Hot path on raw device pointers.  These are what a maintainer binds when the buffers
 * already live on the GPU (the reference's kernel-level interface, sgd.h:12-16, loss.h:18-24).
 * All pointers are device pointers on the current device unless marked host.  `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous unless they
 * return host results.
 * ---------------------------------------------------------------------------------------- */
typedef struct cu2rec_hyper { /* the kernel-visible subset of config (config.h:9-18) */
    float learning_rate, P_reg, Q_reg, user_bias_reg, item_bias_reg;
} cu2rec_hyper;

typedef enum cu2rec_sgd_mode {
    /* One 16-lane group per user, all users of an iteration in flight at once; P, user_bias
     * updated in place by their owner, Q / item_bias rows updated in place with plain stores
     * (Hogwild: concurrent updates of one item overwrite each other, sgd.cu:18-21).  The
     * throughput path. */
    CU2REC_SGD_HOGWILD = 0,
    /* One group walks the users in ascending order: mf_sequential.cu:102-143 exactly
     * (in-place Gauss-Seidel).  Bit-identical to the CPU oracle in its TREE16 dot order.
     * For parity tests; orders of magnitude slower. */
    CU2REC_SGD_SERIAL = 1,
    /* The sequential result at GPU speed, deterministic, no races: the sample stream is value independent,
     * so each iteration is scheduled ahead of time as independent per-item chains whose updates run in
     * ascending user order with the item row held in registers (cu2rec_amd/csrc/ordered.hip).
     * Bit-identical to CU2REC_SGD_SERIAL and to the CPU oracle.  Needs a cu2rec_schedule workspace
     * (the object layer creates one per cu2rec_csr on first use; there the schedule lives in windows of up to 64 iterations
     * that outlive the call -- a call of any length, continuing the last one or not, runs out of the window that holds its
     * iterations, and the window behind it is scheduled beside them; with raw pointers no window outlives the call: the
     * arrays are the caller's). */
    CU2REC_SGD_ORDERED = 2,
    /* The reference GPU kernel's OWN semantics (sgd.cu:22-75 with the loop of training.cu:107-171), for callers that
     * want those trajectories rather than mf_sequential.cu's: every user reads the item side as it was at the start
     * of the iteration, the FIRST user to claim an item (sgd.cu:49-50) writes its new row / bias into a second buffer
     * pair, the pairs swap after every iteration (training.cu:164-165), an item nobody sampled falls back to its value
     * of two iterations ago.  "First" -- a race on a non-atomic flag in the reference -- is the lowest thread index
     * here (thread gid handles user (gid + 250 * iteration) % rows, training.cu:97-98,115), decided with a 64-bit
     * atomicMin, so the mode is deterministic and race free; the reference's surplus threads (grid = rows / n_threads
     * + 1 blocks, which update a few users twice) are not reproduced.  Object layer only (cu2rec_model_sgd,
     * cu2rec_train, bin/mf -m pingpong) or cu2rec_sgd_update_pingpong below. */
    CU2REC_SGD_PINGPONG = 3,
    /* mf_sequential.cu:102-143's semantics at Hogwild-class speed: the schedule of CU2REC_SGD_ORDERED, but the chain of
     * a popular item (thousands of dependent updates per iteration) is solved in blocks of 64 updates -- a block's errors
     * are the solution of a unit lower triangular 64 x 64 system built from the Gram matrix of its 64 user rows (fp32
     * matrix cores); its inverse, formed tile-wise ahead of the chain (two 32 x 32 triangular inversions + two matrix
     * products), turns 64 dependent row updates into three mat-vecs per block (cu2rec_amd/csrc/blocksolve.hip).  This is
     * the default of bin/mf and of bench.py, the mode certified against the north star's 1e-4 RMSE bar.  Equal to the
     * sequential result up to float rounding (sums are associated differently), NOT bit for bit: tests pin it at
     * |test RMSE - oracle| <= 1e-4 after 1,000 iterations of the ML-20M shape.  n_factors <= 252.  Needs a
     * cu2rec_schedule like CU2REC_SGD_ORDERED. */
    CU2REC_SGD_BLOCKSOLVE = 4
} cu2rec_sgd_mode;

/* Replaces sgd_update (sgd.cu:22-75) + the per-iteration launch loop of train()
 * (training.cu:107-115): runs iterations [iter0, iter0 + n_iters), one update per user with
 * at least one rating per iteration.  update_items == 0 is is_train == false.
 * user_offset: global id of row 0 (0 unless the rows are one shard of a user-sharded set); it only
 * enters the sampler, so a shard draws exactly what the unsharded run draws for the same users. */
int cu2rec_sgd_update(const int *indptr, const int *indices, const float *data, int n_rows, int n_cols,
                      float *P, int ldp, float *Q, int ldq, float *user_bias, float *item_bias, float global_bias,
                      int n_factors, const cu2rec_hyper *hyper, uint64_t seed, uint64_t iter0, int n_iters,
                      int mode, int update_items, int user_offset, void *stream);

/* sgd_update (sgd.h:12-16) with its Q_target / item_bias_target / item_is_updated arguments, n_iters iterations with the
 * swap of training.cu:164-165 after each one: CU2REC_SGD_PINGPONG on raw device pointers.  Q_target / item_bias_target:
 * second buffers of the same shapes, initialised by the caller as copies of Q / item_bias (training.cu:37,69-70);
 * claim: n_cols 64-bit words of workspace (the role of item_is_updated; no per-iteration memset, reset by the call).
 * The buffers swap ROLES like the reference's pointers: *swapped = 1 means the current item side is now in
 * Q_target / item_bias_target (an odd number of swaps).  swap_last == 0 leaves out the last iteration's swap, so that
 * a loss evaluated right after the call sees what the reference's does (training.cu:121-137 run before the swap). */
int cu2rec_sgd_update_pingpong(const int *indptr, const int *indices, const float *data, int n_rows, int n_cols,
                               float *P, int ldp, float *Q, float *Q_target, int ldq, float *user_bias,
                               float *item_bias, float *item_bias_target, unsigned long long *claim,
                               float global_bias, int n_factors, const cu2rec_hyper *hyper, uint64_t seed,
                               uint64_t iter0, int n_iters, int update_items, int user_offset, int swap_last,
                               int *swapped, void *stream);

/* Optional sample array for the Hogwild path: [nnz] x {int32 item, float32 rating} = indices[k] and data[k] of the
 * CSR side by side (8 bytes per rating, device memory, 8-byte aligned), so that drawing a rating (sgd.cu:36-44) is
 * one 8-byte gather instead of two 4-byte gathers in different arrays.  The reference's CSR stays the input
 * format; this is a derived copy the caller owns: allocate cu2rec_sample_pairs_bytes(nnz), fill it once with
 * cu2rec_sample_pairs_build, pass it to cu2rec_sgd_update_ex (NULL = gather from indices / data; results are
 * identical either way).  Used by the resident launches (below); the owned-object layer (cu2rec_model_sgd,
 * cu2rec_train) keeps one per cu2rec_csr by itself. */
size_t cu2rec_sample_pairs_bytes(int nnz);
int cu2rec_sample_pairs_build(const int *indices, const float *data, int nnz, void *pairs, void *stream);
int cu2rec_sgd_update_ex(const int *indptr, const int *indices, const float *data, int n_rows, int n_cols,
                         float *P, int ldp, float *Q, int ldq, float *user_bias, float *item_bias, float global_bias,
                         int n_factors, const cu2rec_hyper *hyper, uint64_t seed, uint64_t iter0, int n_iters,
                         int mode, int update_items, int user_offset, const void *sample_pairs, void *stream);

/* Resident Hogwild launches (process-wide; replaces the launch loop of training.cu:107-113 with ONE persistent
 * launch per cu2rec_sgd_update call).  The iterations of a call keep the reference's cadence -- every user's update
 * of iteration i is visible to iteration i+1 -- but the boundary between them is a grid-wide barrier instead of a
 * kernel boundary, and every user's row stays in the register file for the whole call (no P traffic between
 * iterations).  Applies when all user rows of the CSR fit the register file + LDS (e.g. 204,000 users at f <= 128);
 * otherwise the call runs one streaming launch per iteration as before.
 * policy: 0 never, 1 auto (default: when it fits and a call covers >= 4 iterations), 2 whenever it fits;
 * CU2REC_RESIDENT=0|1|2 in the environment sets the initial value.  Returns the previous policy; other values query.
 * A resident launch assumes the GPU to itself (one workgroup per CU, all co-resident).  It is a cooperative launch: a
 * grid the runtime finds too large to be co-resident is refused up front and that call runs one streaming launch per
 * iteration instead (cu2rec_hogwild_resident_refusals counts them) -- nothing is lost.  What a launch-time check cannot
 * see (another process taking CUs while the grid runs) ends at the barrier's 3 s timeout: the next call into the library
 * returns CU2REC_EHIP (once: the model state is undefined after that, but the library stays usable, e.g. with policy 0). */
int cu2rec_hogwild_resident(int policy);
/* resident launches refused at launch time on the current device since the process started (those calls streamed) */
int cu2rec_hogwild_resident_refusals(void);
/* Waits for the current device and reports a resident launch that gave up at its barrier (CU2REC_EHIP, once), without
 * doing any other work: the same check every SGD / loss entry point makes on its way in. */
int cu2rec_check_faults(void);
/* 1 if a Hogwild cu2rec_sgd_update call of n_iters iterations on n_rows users would be one resident launch on the
 * current device under the current policy (then *blocks = workgroups, one per CU, and *users_per_group = rows each
 * 16-lane group keeps in registers; both may be NULL), 0 if it would stream, < 0 on error. */
int cu2rec_hogwild_resident_plan(int n_rows, int n_factors, int n_iters, int *blocks, int *users_per_group);

/* The arithmetic behind cu2rec_hogwild_resident_plan, without a device: 1 if the rows of n_rows users of n_factors floats
 * fit the registers + LDS of n_cus CUs (then *blocks = workgroups, *users_per_group = rows per 16-lane group, of which
 * *lds_rows live in LDS; all may be NULL), 0 if they do not.  Policy and call length are not considered. */
int cu2rec_hogwild_resident_geometry(int n_rows, int n_factors, int n_cus, int *blocks, int *users_per_group,
                                     int *lds_rows);
/* Partial residency (round 4; n_factors <= 256: every row width with a compiled partial form): a set too large for the chip still
 * runs as ONE launch per call -- of the users_per_group users of every group (cu2rec_hogwild_resident_geometry's figure INCLUDES the
 * streamed ones) the first users_per_group - s are resident as above, the other s streamed through the same pipeline, row in,
 * update, row out (the reference's loop is sgd.cu:22-75 either way).  Returns s for the geometry above: 0 = fully resident, > 0 = that
 * many streamed rows per group, -1 = no compiled form holds the set (one launch per iteration). */
int cu2rec_hogwild_resident_streamed_rows(int n_rows, int n_factors, int n_cus);

/* Workspace of CU2REC_SGD_ORDERED for one device CSR: item popularity ranks, key/value buffers of the
 * per-iteration schedule, sort scratch.  indptr / indices are device pointers (read once at creation). */
typedef struct cu2rec_schedule cu2rec_schedule;
int cu2rec_schedule_create(const int *indptr, const int *indices, int n_rows, int n_cols, int nnz,
                           cu2rec_schedule **out);
void cu2rec_schedule_destroy(cu2rec_schedule *s);
/* cu2rec_sgd_update in CU2REC_SGD_ORDERED mode on raw device pointers (same arguments + the workspace). */
int cu2rec_sgd_update_ordered(cu2rec_schedule *schedule, const int *indptr, const int *indices, const float *data,
                              int n_rows, int n_cols, float *P, int ldp, float *Q, int ldq, float *user_bias,
                              float *item_bias, float global_bias, int n_factors, const cu2rec_hyper *hyper,
                              uint64_t seed, uint64_t iter0, int n_iters, int update_items, int user_offset,
                              void *stream);

/* cu2rec_sgd_update in CU2REC_SGD_BLOCKSOLVE mode on raw device pointers (same arguments as the ordered form).
 * Launch topology: an iteration is phases 1-3 on `stream` and the other items' chains on a side stream of the schedule's.  Two forms,
 * same kernels, same bytes, same results (tests/test_gpu_blocksolve.py runs both): DEVICE -- fork and join by a gate kernel, a signal
 * kernel and one waiting workgroup, no event on the main stream; needs kernels of the two streams running SIDE BY SIDE -- and EVENTS
 * (3-6 us per iteration slower; works wherever HIP works).  The library picks per device, on the first block-solve call:
 *   1. CU2REC_BS_GATE=0 / 1 in the environment forces events / device;
 *   2. under a counter pass of rocprofv3 (--pmc; the tool exports ROCPROF_COUNTER_COLLECTION: kernels are serialised across streams):
 *      events -- so a --pmc pass is for counters, never for timing;
 *   3. otherwise a two-stream handshake probe on the streams the iterations will use (a wait kernel queued first on one, satisfied by a
 *      signal kernel on the other, 10 ms bound, both directions): streams sharing one hardware queue (GPU_MAX_HW_QUEUES=1), a tool that
 *      serialises dispatches, another tenant holding the compute units all end in events;
 *   4. a join that gives up mid-run (CU2REC_EHIP from the call or cu2rec_check_faults: the model state is undefined) switches the
 *      device to events for every later call of the process, and the error text says so.
 * cu2rec_blocksolve_topology reports what is in force.
 * Environment knobs the library reads (all optional, read once per process): CU2REC_RESIDENT (Hogwild launch form), CU2REC_BLOCKSOLVE_RATE,
 * CU2REC_BLOCKSOLVE_LOOKAHEAD, CU2REC_BS_GATE, CU2REC_BS_WAIT_S (block-solve mode), CU2REC_READER_THREADS, CU2REC_RATINGS_CACHE (ingest),
 * CU2REC_COMM_TIMEOUT_S, CU2REC_MERGE_ADAPTIVE_C, CU2REC_RCCL_WORLD1 (sharded driver; the last one a debugging aid: a real one-rank
 * communicator) -- ten in all; fault injection exists only in the test builds (make test-hooks). */
int cu2rec_sgd_update_blocksolve(cu2rec_schedule *schedule, const int *indptr, const int *indices, const float *data,
                                 int n_rows, int n_cols, float *P, int ldp, float *Q, int ldq, float *user_bias,
                                 float *item_bias, float global_bias, int n_factors, const cu2rec_hyper *hyper,
                                 uint64_t seed, uint64_t iter0, int n_iters, int update_items, int user_offset,
                                 void *stream);
/* Which items CU2REC_SGD_BLOCKSOLVE solves block-wise: those expected to receive at least `rate` updates per
 * iteration (sum over the item's raters of 1 / the rater's number of ratings); the other chains are walked update by
 * update.  Process-wide, read when a schedule is created; default 240 per 131,072 rating users, scaled with the set (no
 * less than 30: one GPU's shard of a strong-scaling run has shorter chains and the same launch overheads; the Netflix
 * shape's 480,189 users give 880); a value set here or by CU2REC_BLOCKSOLVE_RATE in the environment is taken as it is.
 * rate > 0 sets an explicit threshold, rate < 0 returns to the automatic one, rate == 0 only queries.  Returns what was in
 * force before the call: the explicit threshold, or -1 for automatic (so passing a returned value back restores it). */
float cu2rec_blocksolve_min_rate(float rate);
/* The longest chains in the look-ahead form (block-solve mode, n_factors <= 116; round 4; default: items expected to collect 24 blocks
 * and more per iteration -- the two or three longest chains of the ML-20M shape: -2.7 % per iteration).  Items EXPECTED to collect at
 * least `blocks` x 64 updates per iteration (the leading popularity ranks): phase 1 also builds, for every block of their chains
 * but the first, the 64 x 64 block of lr L that couples it to the block before it, and phase 2 runs the chain with
 *     e_i = M_i (pre_i - N_i e_(i-1))
 * -- two 64 x 64 matrix-vector products on ONE wavefront as the only dependent work per block, the item row following one block
 * behind on other wavefronts (no meeting points; DESIGN.md section 4, "The look-ahead form").  Same results as the plain form
 * up to float rounding.  Process-wide, read when a schedule is created (like cu2rec_blocksolve_min_rate); 0 = off;
 * CU2REC_BLOCKSOLVE_LOOKAHEAD in the environment sets the initial value.  Returns the previous value; blocks < 0 only queries. */
int cu2rec_blocksolve_lookahead_blocks(int blocks);
/* The fork / join form in force on the current device (see cu2rec_sgd_update_blocksolve): 2 device-side gate and join, 0 events,
 * -1 not decided yet (no block-solve call so far).  `why` (may be NULL): the reason, NUL-terminated, truncated to `cap` bytes. */
int cu2rec_blocksolve_topology(char *why, size_t cap);
/* Development aid: while `buffer` (device memory, 8 * (1 + 8 * capacity) bytes, zeroed by the caller) is set, every
 * wavefront of the block-solve kernels writes {kernel, id, start, end, 4 marks (trace builds)} in ticks of the 100 MHz device clock into record
 * kernel * (capacity / 8) + id (kernel: 1 gram, 2 solver, 3 loader, 4 update, 7 cross blocks; records never written stay
 * zero).  NULL switches it off. */
int cu2rec_debug_blocksolve_stamps(void *buffer, int capacity);

/* Replaces calculate_loss_gpu + get_error_metrics_gpu (loss.cu:19-49,150-200) in ONE pass:
 * residual e = r - (gb + ub + ib + p.q) per rating, sum |e| and sum e^2 accumulated in
 * double, MAE = S1/n, RMSE = sqrt(S2/n) returned as float (loss.cu:185-190).
 * errors_out (device, nnz floats) may be NULL; when given it receives the residuals like the
 * reference's error_d array (loss.cu:31).  workspace: device memory of at least
 * cu2rec_loss_workspace_bytes() bytes.  Synchronises `stream` (host results).
 * ALWAYS size the workspace by calling cu2rec_loss_workspace_bytes(): it is NOT a constant of the ABI.  It grew by 16 bytes in
 * round 3 (the per-block partial sums are added on the device and the two results live at the workspace's tail: 2 x 4,096 + 2
 * doubles); a buffer sized by an older formula would be written past its end. */
size_t cu2rec_loss_workspace_bytes(void);
int cu2rec_loss(const int *indptr, const int *indices, const float *data, int n_rows, int nnz,
                const float *P, int ldp, const float *Q, int ldq, const float *user_bias, const float *item_bias,
                float global_bias, int n_factors, float *errors_out, void *workspace, double *sum_abs,
                double *sum_sq, float *mae, float *rmse, void *stream);

/* Replaces total_loss_kernel<B> + calculate_error_metric_gpu x2 (loss.cu:58-128,150-200) on an
 * explicit device array of residuals (tests/test_loss.cu:106-147). */
int cu2rec_error_metrics(const float *errors, int n, void *workspace, float *mae, float *rmse, void *stream);

/* ------------------------------------------------------------------------------------------
 * Owned device objects.  Replace CudaCSRMatrix / CudaDenseMatrix (matrix.h:11-28,
 * matrix.cu:12-46): constructor = allocate + H2D, destroy = free, download = to_host.
 * ---------------------------------------------------------------------------------------- */
typedef struct cu2rec_csr cu2rec_csr;     /* device CSR */
typedef struct cu2rec_model cu2rec_model; /* device P, Q, user_bias, item_bias + global_bias */

int cu2rec_csr_create(int rows, int cols, int nnz, const int *indptr, const int *indices, const float *data,
                      cu2rec_csr **out); /* host arrays in, CudaCSRMatrix ctor matrix.cu:28-40 */
int cu2rec_csr_info(const cu2rec_csr *m, int *rows, int *cols, int *nnz);
int cu2rec_csr_device_ptrs(const cu2rec_csr *m, const int **indptr, const int **indices, const float **data);
void cu2rec_csr_destroy(cu2rec_csr *m);
/* How many items CU2REC_SGD_BLOCKSOLVE solves block-wise on this rating matrix (creates its schedule on first use); 0 = the
 * mode is the ordered walk there.  < 0 on error. */
int cu2rec_csr_blocksolve_items(const cu2rec_csr *train);

/* Host arrays are dense (row stride n_factors), the reference's layout; any of P/Q/biases may
 * be NULL = initialise like the reference: initialize_normal_array(..., seed 42)
 * (training.cu:28,54,212-213). */
int cu2rec_model_create(int rows, int cols, int n_factors, const float *P, const float *Q, const float *user_bias,
                        const float *item_bias, float global_bias, cu2rec_model **out);
int cu2rec_model_info(const cu2rec_model *m, int *rows, int *cols, int *n_factors, int *ld, float *global_bias);
/* *ld above is the row stride of P (n_factors rounded up to 4 floats); this is the row stride of Q: n_factors rounded up
 * to 32 floats, so that every item row is a whole number of 128-byte cache lines (see "Device data layout"). */
int cu2rec_model_item_stride(const cu2rec_model *m);
/* The device arrays of the model (padded rows: P with stride ld, Q with cu2rec_model_item_stride).  Q / item_bias always name the CURRENT item side: after a
 * CU2REC_SGD_PINGPONG call they may be other buffers than before (the mode swaps two pairs), so ask again. */
int cu2rec_model_device_ptrs(const cu2rec_model *m, float **P, float **Q, float **user_bias, float **item_bias);
/* dense host arrays out (any may be NULL): CudaDenseMatrix::to_host + bias copies, training.cu:180-185 */
int cu2rec_model_download(const cu2rec_model *m, float *P, float *Q, float *user_bias, float *item_bias);
void cu2rec_model_destroy(cu2rec_model *m);

int cu2rec_model_sgd(cu2rec_model *m, const cu2rec_csr *train, const cu2rec_hyper *hyper, uint64_t seed,
                     uint64_t iter0, int n_iters, int mode, int update_items);
int cu2rec_model_loss(const cu2rec_model *m, const cu2rec_csr *ratings, double *sum_abs, double *sum_sq,
                      float *mae, float *rmse);

/* ------------------------------------------------------------------------------------------
 * Scoring and ranking for MANY users at once -- replaces predict_ratings (predict.cu:17-30) and get_recommendations
 * (predict.cu:49-65), which the reference runs on the host for its one user.
 * scores[u * cols + i] = ((gb + ub[u]) + ib[i]) + p_u . q_i for every user of the model and every item: one dense
 * product on the matrix cores (f32 in, f32 accumulate).  The sum over the factors is associated differently from the
 * reference's sequential loop: equal up to float rounding.
 * ---------------------------------------------------------------------------------------- */
int cu2rec_model_scores(const cu2rec_model *m, float *scores_device /* rows * cols floats */, void *stream);
int cu2rec_model_scores_host(const cu2rec_model *m, float *scores_host /* rows * cols floats */);
/* For every user the k best items it has NOT rated, best predicted rating first.  rated: a CSR whose row u holds user
 * u's ratings (NULL: nothing is excluded).  items_out / scores_out: host arrays of rows * k entries; a user with fewer
 * than k unrated items gets -1 / NaN padding. */
int cu2rec_model_recommend(const cu2rec_model *m, const cu2rec_csr *rated, int k, int *items_out, float *scores_out);

/* ------------------------------------------------------------------------------------------
 * train() -- replaces both overloads of train (training.h:12-15, training.cu:21-217):
 * total_iterations iterations; loss on train and test when i == 0, (i+1) % check_error == 0
 * or last (training.cu:118) printed as "TRAIN: Iteration %d GPU MAE: %f RMSE: %f" / "TEST: ..."
 * (training.cu:135,137) when `verbose`; patience / learning-rate decay on the test RMSE
 * (training.cu:101-103,146-155, "New Learning Rate" line); "Time taken for %d of iterations
 * is %lf" (training.cu:177).  cfg->learning_rate and cfg->cur_iterations are updated as the
 * reference does (training.cu:152,170).  losses (host, total_iterations floats, may be NULL):
 * losses[i] = test RMSE at the checked iterations (training.cu:158), NaN elsewhere.
 * The model is updated in place; download it afterwards.
 * ---------------------------------------------------------------------------------------- */
typedef struct cu2rec_train_stats {
    double seconds_total;  /* wall time of the loop, loss checks included (training.cu:106,172-177) */
    double seconds_sgd;    /* GPU time of the SGD launches only (HIP events) */
    double updates;        /* users-with-ratings x iterations */
    int n_checks;
    float last_train_mae, last_train_rmse, last_test_mae, last_test_rmse;
} cu2rec_train_stats;

int cu2rec_train(const cu2rec_csr *train, const cu2rec_csr *test, cu2rec_config *cfg, cu2rec_model *model,
                 int mode, int verbose, float *losses, cu2rec_train_stats *stats);

/* ------------------------------------------------------------------------------------------
 * Sharding by user (multi-GPU; the reference is single device).  Host-side planner:
 * user_begin[nranks+1] with contiguous ranges of (almost) equal user count; work per SGD
 * iteration is one update per user.
 * ---------------------------------------------------------------------------------------- */
int cu2rec_shard_plan(int rows, int nranks, int *user_begin);
/* Slice a host CSR to users [u0,u1): indptr_out[u1-u0+1] rebased to 0; returns the slice's
 * nnz in *nnz_out and the offset of its first rating in *offset_out (indices/data slices are
 * then plain sub-arrays). */
int cu2rec_csr_slice(const int *indptr, int rows, int u0, int u1, int *indptr_out, int *offset_out, int *nnz_out);
/* Device helpers for the item-factor exchange (one fused buffer of n_cols*(ld+1) floats):
 *   pack:   buf = [Q - Q_base | item_bias - item_bias_base]
 *   apply:  Q = Q_base + scale * buf_Q ; item_bias likewise ; then Q_base = Q (new snapshot)
 * between them the caller all-reduces `buf` (RCCL via torch.distributed or ncclAllReduce). */
int cu2rec_items_delta_pack(const float *Q, const float *item_bias, const float *Q_base, const float *ib_base,
                            int n_cols, int ldq, float *buf, void *stream);
/* Overlapped exchange: the all-reduce of `buf` runs while training continues.  *_snap hold Q / item_bias as they were
 * when `buf` was packed; when the reduced `buf` arrives:
 *     merged = Q_base + scale * buf;   Q = merged + (Q - Q_snap)   (local progress made meanwhile is kept);
 *     Q_base = merged                  (so the next delta is exactly that local progress), same for item_bias. */
int cu2rec_items_delta_apply_overlapped(float *Q, float *item_bias, float *Q_base, float *ib_base, const float *Q_snap,
                                        const float *ib_snap, int n_cols, int ldq, const float *buf, float scale,
                                        void *stream);
/* Same with a per-item weight (device, n_cols floats) applied to the item's delta row and bias delta: ranks pack
 * w_k[y] * delta_k[y] with sum_k w_k[y] == 1, all-reduce(SUM), apply with scale 1 -- a per-item weighted average. */
int cu2rec_items_delta_pack_weighted(const float *Q, const float *item_bias, const float *Q_base, const float *ib_base,
                                     const float *item_weight, int n_cols, int ldq, float *buf, void *stream);
/* Expected number of SGD updates per iteration that the rows of a host CSR make on each item:
 * rate[y] = sum over users u that rated y of 1 / degree(u)   (each user samples uniformly among its ratings). */
int cu2rec_item_update_rates(const int *indptr, const int *indices, int n_rows, int n_cols, double *rate);
int cu2rec_items_delta_apply(float *Q, float *item_bias, float *Q_base, float *ib_base, int n_cols, int ldq,
                             const float *buf, float scale, void *stream);

/* ------------------------------------------------------------------------------------------
 * User-sharded training on several GPUs, ONE PROCESS PER GPU (the reference is single device; the interface extended
 * is train(), training.h:12-15).  Each rank owns a contiguous user range -- its CSR slice (cu2rec_csr_slice), a model
 * whose P / user_bias cover the local users and whose Q / item_bias are a replica of the whole item side -- and every
 * sync_every iterations the replicas are reconciled by ONE sum all-reduce of the item deltas in their wire format,
 * n_cols * (n_factors + 1) floats without row padding: Q = Q_base + scale * sum_k w_k (Q_k - Q_base).  Sampler draws
 * are keyed by the global user id (user_offset), so a shard draws what the unsharded run draws.
 *
 * The communicator is RCCL over xGMI (resolved at run time with dlopen: a copy already in the process -- PyTorch's -- is
 * reused): created here from an ncclUniqueId the caller distributes (cu2rec_comm_unique_id on rank 0, 128 bytes), or an
 * ncclComm_t the caller already owns, handed in as void*.  For tests there is a callback form: the caller's own
 * in-place sum all-reduce on a device buffer (count elements of float, or double when is_double), ordered on `stream`
 * or synchronising it; return 0 on success.  Every rank must make the same sequence of job calls.
 * ---------------------------------------------------------------------------------------- */
typedef struct cu2rec_comm cu2rec_comm;
typedef int (*cu2rec_allreduce_fn)(void *ctx, void *device_buf, size_t count, int is_double, void *stream);
int cu2rec_comm_unique_id(void *id_out /* 128 bytes */);
int cu2rec_comm_create(const void *unique_id, int rank, int nranks, cu2rec_comm **out); /* ncclCommInitRank on the current device */
int cu2rec_comm_from_nccl(void *nccl_comm, int rank, int nranks, cu2rec_comm **out);    /* not owned */
int cu2rec_comm_from_callback(cu2rec_allreduce_fn fn, void *ctx, int rank, int nranks, cu2rec_comm **out);
void cu2rec_comm_destroy(cu2rec_comm *comm);
/* What the communicator really is, for a bench line / a preflight to PROVE the collective's world: rank / nranks as given at
 * creation; rccl_nranks / rccl_rank / rccl_device from ncclCommCount / ncclCommUserRank / ncclCommCuDevice of the attached
 * ncclComm_t (all 0 / -1 / -1 when no RCCL communicator is attached: one rank, or the callback form); rccl_version from
 * ncclGetVersion (0 if RCCL was never loaded).  Any pointer may be NULL. */
typedef struct cu2rec_comm_info_t {
    int rank, nranks;
    int rccl_nranks, rccl_rank, rccl_device;
    int rccl_version;
    int is_callback; /* 1: the caller's all-reduce (tests), no RCCL on the data path */
} cu2rec_comm_info_t;
int cu2rec_comm_info(const cu2rec_comm *comm, cu2rec_comm_info_t *out);

typedef enum cu2rec_merge {
    CU2REC_MERGE_MEAN = 0,     /* scale 1 / nranks */
    CU2REC_MERGE_WEIGHTED = 1, /* per item: w_k[y] = rank k's share of the item's expected updates per iteration */
    CU2REC_MERGE_SUM = 2,      /* scale 1: diverges when many ranks update the same items many times per period */
    CU2REC_MERGE_ADAPTIVE = 3  /* the sum, scaled per item by phi(r_total) / sum_k phi(r_k), phi(r) = 1 - exp(-c r), r = expected
                                * updates of the item per iteration, c = 6 * min(1, sync_every / epoch) (what saturates a row is its
                                * updates per PERIOD; CU2REC_MERGE_ADAPTIVE_C overrides c): the sum for rarely updated items, the mean
                                * for items every rank updates all the time.  ML-20M shape, 8 shards, 1,000 iterations: test RMSE 3e-4
                                * from the unsharded sequential result (mean / weighted: 1e-3) */
} cu2rec_merge;
typedef struct cu2rec_shard_options {
    int sync_every; /* iterations between exchanges; 0 = one epoch = round(nnz / users) of the whole population */
    int merge;      /* cu2rec_merge */
} cu2rec_shard_options;

typedef struct cu2rec_shard_job cu2rec_shard_job;
/* model and train must outlive the job; options may be NULL (epoch cadence, CU2REC_MERGE_ADAPTIVE: the default of bin/mf
 * and bench.py). Collective: all ranks call it. */
int cu2rec_shard_job_create(cu2rec_comm *comm, cu2rec_model *model, const cu2rec_csr *train, int user_offset,
                            const cu2rec_shard_options *options, cu2rec_shard_job **out);
void cu2rec_shard_job_destroy(cu2rec_shard_job *job);
/* n_iters iterations on the local shard, exchanging every sync_every iterations (the cadence runs across calls) */
int cu2rec_shard_job_run(cu2rec_shard_job *job, const cu2rec_hyper *hyper, uint64_t seed, uint64_t iter0, int n_iters,
                         int mode, int update_items);
/* an exchange out of cadence (if anything happened since the last one): every replica equal afterwards */
int cu2rec_shard_job_exchange(cu2rec_shard_job *job);
/* MAE / RMSE over all ranks' slices of `ratings` (all-reduce of sum |e|, sum e^2, n) */
int cu2rec_shard_job_loss(cu2rec_shard_job *job, const cu2rec_csr *ratings, double *sum_abs, double *sum_sq,
                          double *n_total, float *mae, float *rmse);
int cu2rec_shard_job_info(const cu2rec_shard_job *job, int *sync_every, int *exchanges, double *users_total,
                          double *nnz_total, size_t *wire_bytes);
/* Device time of the exchanges so far: every exchange (items_wire_pack -> all-reduce -> items_wire_apply) is bracketed by a pair
 * of events on its stream; `timed` of them have completed and been read (an exchange whose events were still in flight when
 * their slot was needed again is not counted), `seconds` is their sum on THIS rank -- it includes the wait for the slowest
 * peer.  Call after the stream has been synchronised to have every exchange counted.  Pointers may be NULL. */
int cu2rec_shard_job_exchange_stats(cu2rec_shard_job *job, int *timed, double *seconds, double *max_seconds);
/* cu2rec_train over all ranks: same schedule, same stdout lines (rank 0 prints), patience / learning-rate decay on the
 * GLOBAL test RMSE; `test` is the rank's slice of the test set; stats->updates counts the whole population. */
int cu2rec_train_sharded(cu2rec_shard_job *job, const cu2rec_csr *test, cu2rec_config *cfg, int mode, int verbose,
                         float *losses, cu2rec_train_stats *stats);

#ifdef __cplusplus
}
#endif
#endif /* CU2REC_AMD_H */
