// Device side of the C ABI: raw-pointer hot-path entry points, owned device objects
// (the roles of CudaCSRMatrix / CudaDenseMatrix, matrix.h:11-28) and the item-factor exchange
// helpers.  Host code calls HIP directly; there is no CPU fallback anywhere in this file.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <vector>

#include "device.hpp"
#include "hip_check.hpp"
#include "kernels.hpp"
#include "resident.hpp"

namespace cu2rec {

// what a persistent / concurrent launch could not finish is reported at the next entry point (resident.hip, blocksolve.hip)
static void check_faults() {
    resident_check_fault();
    bs_check_fault();
}


namespace {

hipStream_t as_stream(void *s) { return static_cast<hipStream_t>(s); }


void check_ld(int ld, int n_factors, const char *what) {
    if (ld < n_factors || (ld & 3) != 0)
        fail(CU2REC_EINVAL, std::string(what) + ": row stride must be a multiple of 4 floats and >= n_factors");
}

void check_aligned(const void *p, const char *what) {
    if ((reinterpret_cast<uintptr_t>(p) & 15u) != 0)
        fail(CU2REC_EINVAL, std::string(what) + ": factor matrix must be 16-byte aligned");
}

// Per-block partial sums come back to the host and are added there in block order, so the result
// does not depend on scheduling (the reference also sums its 256 block partials on the host,
// loss.cu:183-189).
struct PartialSums {
    double sum_abs = 0.0, sum_sq = 0.0;
};

// The per-block partial sums are added ON THE DEVICE (one more tiny launch, 6 us) and two doubles come back, into pinned memory:
// the reference copies every block's partial sum back and adds them on the host, twice per metric pair (loss.cu:183-190), and
// so did rounds 1-2 here (64 KB through a pageable staging copy).
PartialSums collect_partials(double *device_partials, int blocks, hipStream_t stream) {
    // two doubles of pinned memory per host thread, released with the thread (the runtime may be gone by then: errors ignored)
    struct Pinned {
        double *p = nullptr;
        ~Pinned() {
            if (p) (void)hipHostFree(p);
        }
    };
    thread_local Pinned holder;
    if (!holder.p) CU2REC_HIP(hipHostMalloc(reinterpret_cast<void **>(&holder.p), 2 * sizeof(double), hipHostMallocDefault));
    double *pinned = holder.p;
    launch_partials_reduce(device_partials, blocks, stream);
    CU2REC_HIP(hipGetLastError());
    CU2REC_HIP(hipMemcpyAsync(pinned, device_partials + 2 * kMaxPartialBlocks, 2 * sizeof(double), hipMemcpyDeviceToHost, stream));
    CU2REC_HIP(hipStreamSynchronize(stream));
    PartialSums s;
    s.sum_abs = pinned[0];
    s.sum_sq = pinned[1];
    return s;
}

}  // namespace

SgdArgs make_sgd_args(const int *indptr, const int *indices, const float *data, int n_rows, int n_cols, float *P,
                      int ldp, float *Q, int ldq, float *user_bias, float *item_bias, float global_bias, int n_factors,
                      const cu2rec_hyper &hyper, uint64_t seed, int update_items, int user_offset) {
    require(indptr && P && Q && user_bias && item_bias, "cu2rec_sgd_update: null device pointer");
    require(n_rows >= 0 && n_cols >= 0 && n_factors > 0, "cu2rec_sgd_update: bad shape");
    require(n_rows == 0 || (indices && data), "cu2rec_sgd_update: null device pointer");
    check_ld(ldp, n_factors, "P");
    check_ld(ldq, n_factors, "Q");
    check_aligned(P, "P");
    check_aligned(Q, "Q");
    SgdArgs a{};
    a.indptr = indptr;
    a.indices = indices;
    a.data = data;
    a.P = P;
    a.Q = Q;
    a.user_bias = user_bias;
    a.item_bias = item_bias;
    a.n_rows = n_rows;
    a.ldp = ldp;
    a.ldq = ldq;
    a.nslots = (n_factors + 3) / 4;
    a.global_bias = global_bias;
    a.h = SgdHyper{hyper.learning_rate, hyper.P_reg, hyper.Q_reg, hyper.user_bias_reg, hyper.item_bias_reg};
    a.seed = seed;
    a.update_items = update_items ? 1 : 0;
    a.user_offset = user_offset;
    return a;
}

void sgd_update(const int *indptr, const int *indices, const float *data, int n_rows, int n_cols, float *P, int ldp,
                float *Q, int ldq, float *user_bias, float *item_bias, float global_bias, int n_factors,
                const cu2rec_hyper &hyper, uint64_t seed, uint64_t iter0, int n_iters, int mode, int update_items,
                int user_offset, hipStream_t stream, const void *sample_pairs) {
    require(mode == CU2REC_SGD_HOGWILD || mode == CU2REC_SGD_SERIAL,
            "cu2rec_sgd_update: unknown mode (CU2REC_SGD_ORDERED needs cu2rec_sgd_update_ordered and a schedule)");
    require(n_iters >= 0, "cu2rec_sgd_update: bad iteration count");
    SgdArgs a = make_sgd_args(indptr, indices, data, n_rows, n_cols, P, ldp, Q, ldq, user_bias, item_bias, global_bias,
                              n_factors, hyper, seed, update_items, user_offset);
    if (n_rows == 0 || n_iters == 0) return;
    require_device();
    check_faults();
    require((reinterpret_cast<uintptr_t>(sample_pairs) & 7u) == 0, "cu2rec_sgd_update: sample pairs must be 8-byte aligned");
    a.pairs = static_cast<const uint2 *>(sample_pairs);
    if (mode == CU2REC_SGD_SERIAL) {
        // one launch walks all iterations in order
        a.iter0 = iter0;
        a.iters = n_iters;
        launch_sgd(a, mode, stream);
    } else {
        // one launch per reference iteration (training.cu:107-113): the kernel boundary is the
        // point where every user's update of iteration i is visible to iteration i+1
        // frozen items (is_train == false, predict.cu:105,126): no update crosses users, so the whole call is ONE launch
        // with every user's row in registers for all its iterations -- the same results as one launch per iteration
        const int block = update_items ? 1 : n_iters;
        // the same iterations in ONE persistent launch (grid barrier where the kernel boundary was, user rows
        // resident in registers) whenever the rows fit and the policy allows it: resident.hip
        // (n_cols > 0: in a resident launch users without ratings read item row 0 and write to a sink)
        if (block == 1 && n_cols > 0 && resident_launch(a, iter0, n_iters, stream)) return;
        for (int i = 0; i < n_iters; i += block) {
            a.iter0 = iter0 + static_cast<uint64_t>(i);
            a.iters = std::min(block, n_iters - i);
            launch_sgd(a, mode, stream);
        }
    }
    CU2REC_HIP(hipGetLastError());
}

bool sgd_update_pingpong(const int *indptr, const int *indices, const float *data, int n_rows, int n_cols, float *P,
                         int ldp, float *Q, float *Q_target, int ldq, float *user_bias, float *item_bias,
                         float *item_bias_target, unsigned long long *claim, float global_bias, int n_factors,
                         const cu2rec_hyper &hyper, uint64_t seed, uint64_t iter0, int n_iters, int update_items,
                         int user_offset, bool swap_last, hipStream_t stream) {
    require(n_iters >= 0, "cu2rec_sgd_update_pingpong: bad iteration count");
    require(Q_target && item_bias_target && claim, "cu2rec_sgd_update_pingpong: null device pointer");
    check_aligned(Q_target, "Q_target");
    SgdArgs a = make_sgd_args(indptr, indices, data, n_rows, n_cols, P, ldp, Q, ldq, user_bias, item_bias, global_bias,
                              n_factors, hyper, seed, update_items, user_offset);
    if (n_rows == 0 || n_iters == 0) return false;
    require_device();
    check_faults();
    // claims carry the iteration number; a call may restart from an earlier one, so they start from "nobody"
    CU2REC_HIP(hipMemsetAsync(claim, 0xFF, static_cast<size_t>(std::max(n_cols, 1)) * sizeof(unsigned long long), stream));
    PingPongArgs pp{Q_target, item_bias_target, claim};
    a.iters = 1;
    bool swapped = false;
    for (int i = 0; i < n_iters; ++i) {  // one claim + one update launch per iteration (training.cu:107-113)
        a.iter0 = iter0 + static_cast<uint64_t>(i);
        launch_sgd_pingpong(a, pp, stream);
        if (i + 1 < n_iters || swap_last) {  // training.cu:164-165
            std::swap(a.Q, pp.Q_target);
            std::swap(a.item_bias, pp.item_bias_target);
            swapped = !swapped;
        }
    }
    CU2REC_HIP(hipGetLastError());
    return swapped;
}

void sample_pairs_build(const int *indices, const float *data, int nnz, void *pairs, hipStream_t stream) {
    require(nnz >= 0, "cu2rec_sample_pairs_build: bad size");
    if (nnz == 0) return;
    require(indices && data && pairs, "cu2rec_sample_pairs_build: null device pointer");
    require((reinterpret_cast<uintptr_t>(pairs) & 7u) == 0, "cu2rec_sample_pairs_build: pairs must be 8-byte aligned");
    require_device();
    launch_sample_pairs_build(indices, data, static_cast<size_t>(nnz), static_cast<uint2 *>(pairs), stream);
    CU2REC_HIP(hipGetLastError());
}

void sgd_update_ordered(OrderedSchedule &schedule, const int *indptr, const int *indices, const float *data,
                        int n_rows, int n_cols, float *P, int ldp, float *Q, int ldq, float *user_bias,
                        float *item_bias, float global_bias, int n_factors, const cu2rec_hyper &hyper, uint64_t seed,
                        uint64_t iter0, int n_iters, int update_items, int user_offset, hipStream_t stream,
                        bool blocksolve) {
    if (!update_items) {
        // frozen items: no two updates of an iteration share a written row, so the parallel schedule IS the
        // sequential result; no chains needed
        sgd_update(indptr, indices, data, n_rows, n_cols, P, ldp, Q, ldq, user_bias, item_bias, global_bias, n_factors,
                   hyper, seed, iter0, n_iters, CU2REC_SGD_HOGWILD, 0, user_offset, stream);
        return;
    }
    require(n_iters >= 0, "cu2rec_sgd_update_ordered: bad iteration count");
    require(schedule.n_rows == n_rows && schedule.n_cols == n_cols, "cu2rec_sgd_update_ordered: schedule built for another CSR");
    SgdArgs a = make_sgd_args(indptr, indices, data, n_rows, n_cols, P, ldp, Q, ldq, user_bias, item_bias, global_bias,
                              n_factors, hyper, seed, update_items, user_offset);
    if (n_rows == 0 || n_iters == 0) return;
    require_device();
    check_faults();
    schedule.run(a, iter0, n_iters, stream, blocksolve);
}

void loss(const int *indptr, const int *indices, const float *data, int n_rows, int nnz, const float *P, int ldp,
          const float *Q, int ldq, const float *user_bias, const float *item_bias, float global_bias, int n_factors,
          float *errors_out, void *workspace, double *sum_abs, double *sum_sq, float *mae, float *rmse,
          hipStream_t stream) {
    require(n_rows >= 0 && nnz >= 0 && n_factors > 0, "cu2rec_loss: bad shape");
    PartialSums s;
    if (nnz > 0) {
        require(indptr && indices && data && P && Q && user_bias && item_bias && workspace,
                "cu2rec_loss: null device pointer");
        check_ld(ldp, n_factors, "P");
        check_ld(ldq, n_factors, "Q");
        check_aligned(P, "P");
        check_aligned(Q, "Q");
        require_device();
        check_faults();
        LossArgs a{};
        a.indptr = indptr;
        a.indices = indices;
        a.data = data;
        a.P = P;
        a.Q = Q;
        a.user_bias = user_bias;
        a.item_bias = item_bias;
        a.n_rows = n_rows;
        a.nnz = nnz;
        a.ldp = ldp;
        a.ldq = ldq;
        a.nslots = (n_factors + 3) / 4;
        a.global_bias = global_bias;
        a.errors_out = errors_out;
        a.partials = static_cast<double *>(workspace);
        const int blocks = loss_blocks(nnz);
        launch_loss(a, blocks, stream);
        CU2REC_HIP(hipGetLastError());
        s = collect_partials(a.partials, blocks, stream);
        check_faults();  // the stream has been drained: a resident launch before this pass has reported by now
    }
    if (sum_abs) *sum_abs = s.sum_abs;
    if (sum_sq) *sum_sq = s.sum_sq;
    // loss.cu:189: total / n and sqrt(total / n), narrowed to float by the return type
    if (mae) *mae = static_cast<float>(s.sum_abs / nnz);
    if (rmse) *rmse = static_cast<float>(std::sqrt(s.sum_sq / nnz));
}

void error_metrics(const float *errors, int n, void *workspace, float *mae, float *rmse, hipStream_t stream) {
    require(n >= 0, "cu2rec_error_metrics: bad size");
    PartialSums s;
    if (n > 0) {
        require(errors && workspace, "cu2rec_error_metrics: null device pointer");
        require_device();
        const int blocks = error_metrics_blocks(n);
        launch_error_metrics(errors, n, static_cast<double *>(workspace), blocks, stream);
        CU2REC_HIP(hipGetLastError());
        s = collect_partials(static_cast<double *>(workspace), blocks, stream);
    }
    if (mae) *mae = static_cast<float>(s.sum_abs / n);
    if (rmse) *rmse = static_cast<float>(std::sqrt(s.sum_sq / n));
}

// ---------------------------------------------------------------------------------- DeviceCsr
DeviceCsr::DeviceCsr(int rows_, int cols_, int nnz_, const int *h_indptr, const int *h_indices, const float *h_data)
    : rows(rows_), cols(cols_), nnz(nnz_) {
    require(rows >= 0 && cols >= 0 && nnz >= 0 && h_indptr, "cu2rec_csr_create: bad argument");
    require(nnz == 0 || (h_indices && h_data), "cu2rec_csr_create: null indices/data");
    // the reference trusts its input (Appendix: a test file with a larger id than train reads out of
    // bounds, mf.cu:50-51); a malformed CSR would fault the GPU, so check it here on the host
    require(h_indptr[0] == 0 && h_indptr[rows] == nnz, "cu2rec_csr_create: indptr must run from 0 to nnz");
    for (int u = 0; u < rows; ++u) require(h_indptr[u] <= h_indptr[u + 1], "cu2rec_csr_create: indptr must be non-decreasing");
    max_item = -1;
    for (int k = 0; k < nnz; ++k) {
        require(h_indices[k] >= 0, "cu2rec_csr_create: negative item id");
        if (h_indices[k] > max_item) max_item = h_indices[k];
    }
    require(max_item < cols, "cu2rec_csr_create: item id >= cols");
    users_with_ratings = 0;
    for (int u = 0; u < rows; ++u) users_with_ratings += h_indptr[u + 1] > h_indptr[u];
    require_device();
    indptr.allocate(static_cast<size_t>(rows) + 1);
    indptr.upload(h_indptr, static_cast<size_t>(rows) + 1);
    indices.allocate(std::max(nnz, 1));
    data.allocate(std::max(nnz, 1));
    if (nnz) {
        indices.upload(h_indices, nnz);
        data.upload(h_data, nnz);
    }
}

// ---------------------------------------------------------------------------------- DeviceModel
namespace {
// dense host rows (stride f) -> padded device rows (stride ld, zero fill)
void upload_padded(DeviceBuffer<float> &dst, const float *host, int rows, int f, int ld) {
    dst.allocate(std::max<size_t>(static_cast<size_t>(rows) * ld, 4));
    dst.zero();
    if (rows == 0) return;
    CU2REC_HIP(hipMemcpy2D(dst.ptr, static_cast<size_t>(ld) * sizeof(float), host, static_cast<size_t>(f) * sizeof(float),
                           static_cast<size_t>(f) * sizeof(float), rows, hipMemcpyHostToDevice));
}

void download_padded(const DeviceBuffer<float> &src, float *host, int rows, int f, int ld) {
    if (rows == 0) return;
    CU2REC_HIP(hipMemcpy2D(host, static_cast<size_t>(f) * sizeof(float), src.ptr, static_cast<size_t>(ld) * sizeof(float),
                           static_cast<size_t>(f) * sizeof(float), rows, hipMemcpyDeviceToHost));
}
}  // namespace

DeviceModel::DeviceModel(int rows_, int cols_, int f_, const float *hP, const float *hQ, const float *hub,
                         const float *hib, float gb)
    : rows(rows_), cols(cols_), n_factors(f_), ld((f_ + 3) & ~3), ldq((f_ + 31) & ~31), global_bias(gb) {
    require(rows >= 0 && cols >= 0 && n_factors > 0, "cu2rec_model_create: bad shape");
    if (slots_per_lane(ld / 4) > 8) fail(CU2REC_EUNSUPPORTED, "n_factors above 512 is not compiled in");
    require_device();
    // training.cu:28,54,212-213: every array starts from initialize_normal_array(size, n_factors) with seed 42
    std::vector<float> tmp;
    auto init_if_null = [&](const float *given, size_t n) -> const float * {
        if (given) return given;
        tmp.resize(n);
        cu2rec_init_normal(tmp.data(), n, n_factors, 0.f, 1.f, 42);
        return tmp.data();
    };
    upload_padded(P, init_if_null(hP, static_cast<size_t>(rows) * n_factors), rows, n_factors, ld);
    upload_padded(Q, init_if_null(hQ, static_cast<size_t>(cols) * n_factors), cols, n_factors, ldq);
    user_bias.allocate(std::max(rows, 1));
    item_bias.allocate(std::max(cols, 1));
    if (rows) user_bias.upload(init_if_null(hub, rows), rows);
    if (cols) item_bias.upload(init_if_null(hib, cols), cols);
    workspace.allocate(static_cast<size_t>(2) * kMaxPartialBlocks + 2);
}

void DeviceModel::download(float *hP, float *hQ, float *hub, float *hib) const {
    require(!swap_pending, "cu2rec_model_download: a ping-pong swap is pending (finish_swap)");
    CU2REC_HIP(hipDeviceSynchronize());
    check_faults();
    if (hP) download_padded(P, hP, rows, n_factors, ld);
    if (hQ) download_padded(Q, hQ, cols, n_factors, ldq);
    if (hub && rows) user_bias.download(hub, rows);
    if (hib && cols) item_bias.download(hib, cols);
}

void DeviceModel::finish_swap() {
    if (!swap_pending) return;
    Q.swap(Q_other);
    item_bias.swap(item_bias_other);
    swap_pending = false;
}

void DeviceModel::sgd(const DeviceCsr &train, const cu2rec_hyper &h, uint64_t seed, uint64_t iter0, int n_iters, int mode,
                      int update_items, hipStream_t stream, bool defer_last_swap, int user_offset) {
    require(train.rows <= rows && train.max_item < cols, "cu2rec_model_sgd: ratings exceed the model's shape");
    finish_swap();
    if (mode == CU2REC_SGD_PINGPONG) {
        if (!Q_other.ptr) {
            Q_other.allocate(Q.count);
            item_bias_other.allocate(item_bias.count);
            claim.allocate(std::max(cols, 1));
        }
        if (!other_in_sync) {  // training.cu:37,69-70: the targets start as copies of Q / item_bias
            CU2REC_HIP(hipMemcpyAsync(Q_other.ptr, Q.ptr, Q.count * sizeof(float), hipMemcpyDeviceToDevice, stream));
            CU2REC_HIP(hipMemcpyAsync(item_bias_other.ptr, item_bias.ptr, item_bias.count * sizeof(float),
                                      hipMemcpyDeviceToDevice, stream));
            other_in_sync = true;
        }
        const bool swapped = sgd_update_pingpong(train.indptr.ptr, train.indices.ptr, train.data.ptr, train.rows, cols, P.ptr,
                                                 ld, Q.ptr, Q_other.ptr, ldq, user_bias.ptr, item_bias.ptr,
                                                 item_bias_other.ptr, claim.ptr, global_bias, n_factors, h, seed, iter0,
                                                 n_iters, update_items, user_offset, !defer_last_swap, stream);
        if (swapped) {  // Q / item_bias always name the current item side
            Q.swap(Q_other);
            item_bias.swap(item_bias_other);
        }
        swap_pending = defer_last_swap && n_iters > 0 && train.rows > 0;
        return;
    }
    other_in_sync = false;  // another mode moves Q on without the second pair
    if (mode == CU2REC_SGD_ORDERED || mode == CU2REC_SGD_BLOCKSOLVE) {
        if (!train.schedule || train.schedule->n_cols != cols) {
            train.schedule.reset(new OrderedSchedule(train.indptr.ptr, train.indices.ptr, train.rows, cols, train.nnz));
            train.schedule->speculate = true;  // the schedule is destroyed with the arrays it reads
        }
        sgd_update_ordered(*train.schedule, train.indptr.ptr, train.indices.ptr, train.data.ptr, train.rows, cols, P.ptr,
                           ld, Q.ptr, ldq, user_bias.ptr, item_bias.ptr, global_bias, n_factors, h, seed, iter0, n_iters,
                           update_items, user_offset, stream, mode == CU2REC_SGD_BLOCKSOLVE);
        return;
    }
    const void *pairs = nullptr;
    if (mode == CU2REC_SGD_HOGWILD && update_items && train.nnz > 0 &&
        resident_plan(train.rows, n_factors, n_iters, nullptr, nullptr)) {
        if (!train.pairs.ptr) {  // first resident call on this CSR: 8 bytes per rating, built once
            train.pairs.allocate(train.nnz);
            sample_pairs_build(train.indices.ptr, train.data.ptr, train.nnz, train.pairs.ptr, stream);
        }
        pairs = train.pairs.ptr;
    }
    sgd_update(train.indptr.ptr, train.indices.ptr, train.data.ptr, train.rows, cols, P.ptr, ld, Q.ptr, ldq,
               user_bias.ptr, item_bias.ptr, global_bias, n_factors, h, seed, iter0, n_iters, mode, update_items, user_offset,
               stream, pairs);
}

void DeviceModel::loss(const DeviceCsr &ratings, double *sum_abs, double *sum_sq, float *mae, float *rmse,
                       hipStream_t stream) const {
    // Appendix quirk 6: a test file may have fewer users than train, never more
    require(ratings.rows <= rows && ratings.max_item < cols, "cu2rec_model_loss: ratings exceed the model's shape");
    cu2rec::loss(ratings.indptr.ptr, ratings.indices.ptr, ratings.data.ptr, ratings.rows, ratings.nnz, P.ptr, ld, Q.ptr,
                 ldq, user_bias.ptr, item_bias.ptr, global_bias, n_factors, nullptr, workspace.ptr, sum_abs, sum_sq, mae,
                 rmse, stream);
}

}  // namespace cu2rec

using namespace cu2rec;

struct cu2rec_schedule {
    OrderedSchedule impl;
    template <class... A>
    explicit cu2rec_schedule(A &&...a) : impl(std::forward<A>(a)...) {}
};
struct cu2rec_csr {
    DeviceCsr impl;
    template <class... A>
    explicit cu2rec_csr(A &&...a) : impl(std::forward<A>(a)...) {}
};
struct cu2rec_model {
    DeviceModel impl;
    template <class... A>
    explicit cu2rec_model(A &&...a) : impl(std::forward<A>(a)...) {}
};

namespace cu2rec {
DeviceCsr &unwrap(cu2rec_csr *m) { return m->impl; }
const DeviceCsr &unwrap(const cu2rec_csr *m) { return m->impl; }
DeviceModel &unwrap(cu2rec_model *m) { return m->impl; }
const DeviceModel &unwrap(const cu2rec_model *m) { return m->impl; }
}  // namespace cu2rec

extern "C" {

int cu2rec_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int cu2rec_set_device(int device) {
    return guarded([&] {
        require_device();
        CU2REC_HIP(hipSetDevice(device));
    });
}

int cu2rec_sgd_update(const int *indptr, const int *indices, const float *data, int n_rows, int n_cols, float *P,
                      int ldp, float *Q, int ldq, float *user_bias, float *item_bias, float global_bias, int n_factors,
                      const cu2rec_hyper *hyper, uint64_t seed, uint64_t iter0, int n_iters, int mode, int update_items,
                      int user_offset, void *stream) {
    return guarded([&] {
        require(hyper, "cu2rec_sgd_update: hyper is null");
        sgd_update(indptr, indices, data, n_rows, n_cols, P, ldp, Q, ldq, user_bias, item_bias, global_bias, n_factors,
                   *hyper, seed, iter0, n_iters, mode, update_items, user_offset, as_stream(stream));
    });
}

int cu2rec_sgd_update_ex(const int *indptr, const int *indices, const float *data, int n_rows, int n_cols, float *P,
                         int ldp, float *Q, int ldq, float *user_bias, float *item_bias, float global_bias, int n_factors,
                         const cu2rec_hyper *hyper, uint64_t seed, uint64_t iter0, int n_iters, int mode, int update_items,
                         int user_offset, const void *sample_pairs, void *stream) {
    return guarded([&] {
        require(hyper, "cu2rec_sgd_update_ex: hyper is null");
        sgd_update(indptr, indices, data, n_rows, n_cols, P, ldp, Q, ldq, user_bias, item_bias, global_bias, n_factors,
                   *hyper, seed, iter0, n_iters, mode, update_items, user_offset, as_stream(stream), sample_pairs);
    });
}

size_t cu2rec_sample_pairs_bytes(int nnz) { return nnz > 0 ? static_cast<size_t>(nnz) * 8 : 0; }

int cu2rec_sample_pairs_build(const int *indices, const float *data, int nnz, void *pairs, void *stream) {
    return guarded([&] { sample_pairs_build(indices, data, nnz, pairs, as_stream(stream)); });
}

int cu2rec_sgd_update_pingpong(const int *indptr, const int *indices, const float *data, int n_rows, int n_cols,
                               float *P, int ldp, float *Q, float *Q_target, int ldq, float *user_bias,
                               float *item_bias, float *item_bias_target, unsigned long long *claim,
                               float global_bias, int n_factors, const cu2rec_hyper *hyper, uint64_t seed,
                               uint64_t iter0, int n_iters, int update_items, int user_offset, int swap_last,
                               int *swapped, void *stream) {
    return guarded([&] {
        require(hyper, "cu2rec_sgd_update_pingpong: hyper is null");
        const bool s = sgd_update_pingpong(indptr, indices, data, n_rows, n_cols, P, ldp, Q, Q_target, ldq, user_bias,
                                           item_bias, item_bias_target, claim, global_bias, n_factors, *hyper, seed, iter0,
                                           n_iters, update_items, user_offset, swap_last != 0, as_stream(stream));
        if (swapped) *swapped = s ? 1 : 0;
    });
}

int cu2rec_hogwild_resident(int policy) { return resident_policy(policy); }

int cu2rec_hogwild_resident_refusals(void) { return resident_refusals(); }

int cu2rec_check_faults(void) {
    return guarded([&] {
        require_device();
        CU2REC_HIP(hipDeviceSynchronize());
        check_faults();
    });
}

int cu2rec_hogwild_resident_plan(int n_rows, int n_factors, int n_iters, int *blocks, int *users_per_group) {
    int yes = 0;
    const int rc = guarded([&] {
        require(n_rows >= 0 && n_factors > 0, "cu2rec_hogwild_resident_plan: bad shape");
        require_device();
        yes = resident_plan(n_rows, n_factors, n_iters, blocks, users_per_group) ? 1 : 0;
    });
    return rc == CU2REC_OK ? yes : rc;
}

int cu2rec_hogwild_resident_geometry(int n_rows, int n_factors, int n_cus, int *blocks, int *users_per_group,
                                     int *lds_rows) {
    return resident_geometry(n_rows, n_factors, n_cus, blocks, users_per_group, lds_rows) ? 1 : 0;
}

int cu2rec_hogwild_resident_streamed_rows(int n_rows, int n_factors, int n_cus) { return resident_streamed_rows(n_rows, n_factors, n_cus); }

int cu2rec_schedule_create(const int *indptr, const int *indices, int n_rows, int n_cols, int nnz,
                           cu2rec_schedule **out) {
    return guarded([&] {
        require(out, "out is null");
        *out = nullptr;
        *out = new cu2rec_schedule(indptr, indices, n_rows, n_cols, nnz);
    });
}

void cu2rec_schedule_destroy(cu2rec_schedule *s) { delete s; }

int cu2rec_sgd_update_ordered(cu2rec_schedule *schedule, const int *indptr, const int *indices, const float *data,
                              int n_rows, int n_cols, float *P, int ldp, float *Q, int ldq, float *user_bias,
                              float *item_bias, float global_bias, int n_factors, const cu2rec_hyper *hyper,
                              uint64_t seed, uint64_t iter0, int n_iters, int update_items, int user_offset,
                              void *stream) {
    return guarded([&] {
        require(schedule && hyper, "cu2rec_sgd_update_ordered: null argument");
        sgd_update_ordered(schedule->impl, indptr, indices, data, n_rows, n_cols, P, ldp, Q, ldq, user_bias, item_bias,
                           global_bias, n_factors, *hyper, seed, iter0, n_iters, update_items, user_offset,
                           as_stream(stream));
    });
}

int cu2rec_sgd_update_blocksolve(cu2rec_schedule *schedule, const int *indptr, const int *indices, const float *data,
                                 int n_rows, int n_cols, float *P, int ldp, float *Q, int ldq, float *user_bias,
                                 float *item_bias, float global_bias, int n_factors, const cu2rec_hyper *hyper,
                                 uint64_t seed, uint64_t iter0, int n_iters, int update_items, int user_offset,
                                 void *stream) {
    return guarded([&] {
        require(schedule && hyper, "cu2rec_sgd_update_blocksolve: null argument");
        sgd_update_ordered(schedule->impl, indptr, indices, data, n_rows, n_cols, P, ldp, Q, ldq, user_bias, item_bias,
                           global_bias, n_factors, *hyper, seed, iter0, n_iters, update_items, user_offset,
                           as_stream(stream), true);
    });
}

float cu2rec_blocksolve_min_rate(float rate) { return blocksolve_min_rate(rate); }
int cu2rec_blocksolve_lookahead_blocks(int blocks) { return blocksolve_lookahead_blocks(blocks); }

int cu2rec_blocksolve_topology(char *why, size_t cap) {
    std::string text;
    const int mode = bs_topology_query(&text);
    if (why && cap) {
        const size_t n = std::min(text.size(), cap - 1);
        std::memcpy(why, text.data(), n);
        why[n] = 0;
    }
    return mode;
}

int cu2rec_csr_blocksolve_items(const cu2rec_csr *train) {
    int n = -1;
    const int rc = guarded([&] {
        require(train, "cu2rec_csr_blocksolve_items: null argument");
        const DeviceCsr &m = unwrap(train);
        if (!m.schedule) {
            m.schedule.reset(new OrderedSchedule(m.indptr.ptr, m.indices.ptr, m.rows, m.cols, m.nnz));
            m.schedule->speculate = true;
        }
        n = m.schedule->n_hot_bs;
    });
    return rc == CU2REC_OK ? n : -1;
}

int cu2rec_debug_blocksolve_stamps(void *buffer, int capacity) {
    return guarded([&] {
        require(capacity >= 0, "cu2rec_debug_blocksolve_stamps: bad capacity");
        bs_set_stamps(static_cast<unsigned long long *>(buffer), capacity);
    });
}

size_t cu2rec_loss_workspace_bytes(void) { return sizeof(double) * (2 * kMaxPartialBlocks + 2); }

int cu2rec_loss(const int *indptr, const int *indices, const float *data, int n_rows, int nnz, const float *P, int ldp,
                const float *Q, int ldq, const float *user_bias, const float *item_bias, float global_bias,
                int n_factors, float *errors_out, void *workspace, double *sum_abs, double *sum_sq, float *mae,
                float *rmse, void *stream) {
    return guarded([&] {
        loss(indptr, indices, data, n_rows, nnz, P, ldp, Q, ldq, user_bias, item_bias, global_bias, n_factors,
             errors_out, workspace, sum_abs, sum_sq, mae, rmse, as_stream(stream));
    });
}

int cu2rec_error_metrics(const float *errors, int n, void *workspace, float *mae, float *rmse, void *stream) {
    return guarded([&] { error_metrics(errors, n, workspace, mae, rmse, as_stream(stream)); });
}

int cu2rec_csr_create(int rows, int cols, int nnz, const int *indptr, const int *indices, const float *data,
                      cu2rec_csr **out) {
    return guarded([&] {
        require(out, "out is null");
        *out = nullptr;
        *out = new cu2rec_csr(rows, cols, nnz, indptr, indices, data);
    });
}

int cu2rec_csr_info(const cu2rec_csr *m, int *rows, int *cols, int *nnz) {
    return guarded([&] {
        require(m, "csr is null");
        if (rows) *rows = m->impl.rows;
        if (cols) *cols = m->impl.cols;
        if (nnz) *nnz = m->impl.nnz;
    });
}

int cu2rec_csr_device_ptrs(const cu2rec_csr *m, const int **indptr, const int **indices, const float **data) {
    return guarded([&] {
        require(m, "csr is null");
        if (indptr) *indptr = m->impl.indptr.ptr;
        if (indices) *indices = m->impl.indices.ptr;
        if (data) *data = m->impl.data.ptr;
    });
}

void cu2rec_csr_destroy(cu2rec_csr *m) { delete m; }

int cu2rec_model_create(int rows, int cols, int n_factors, const float *P, const float *Q, const float *user_bias,
                        const float *item_bias, float global_bias, cu2rec_model **out) {
    return guarded([&] {
        require(out, "out is null");
        *out = nullptr;
        *out = new cu2rec_model(rows, cols, n_factors, P, Q, user_bias, item_bias, global_bias);
    });
}

int cu2rec_model_item_stride(const cu2rec_model *m) { return m ? m->impl.ldq : 0; }

int cu2rec_model_info(const cu2rec_model *m, int *rows, int *cols, int *n_factors, int *ld, float *global_bias) {
    return guarded([&] {
        require(m, "model is null");
        if (rows) *rows = m->impl.rows;
        if (cols) *cols = m->impl.cols;
        if (n_factors) *n_factors = m->impl.n_factors;
        if (ld) *ld = m->impl.ld;
        if (global_bias) *global_bias = m->impl.global_bias;
    });
}

int cu2rec_model_device_ptrs(const cu2rec_model *m, float **P, float **Q, float **user_bias, float **item_bias) {
    return guarded([&] {
        require(m, "model is null");
        if (P) *P = m->impl.P.ptr;
        if (Q) *Q = m->impl.Q.ptr;
        if (user_bias) *user_bias = m->impl.user_bias.ptr;
        if (item_bias) *item_bias = m->impl.item_bias.ptr;
    });
}

int cu2rec_model_download(const cu2rec_model *m, float *P, float *Q, float *user_bias, float *item_bias) {
    return guarded([&] {
        require(m, "model is null");
        m->impl.download(P, Q, user_bias, item_bias);
    });
}

void cu2rec_model_destroy(cu2rec_model *m) { delete m; }

int cu2rec_model_sgd(cu2rec_model *m, const cu2rec_csr *train, const cu2rec_hyper *hyper, uint64_t seed, uint64_t iter0,
                     int n_iters, int mode, int update_items) {
    return guarded([&] {
        require(m && train && hyper, "null argument");
        m->impl.sgd(train->impl, *hyper, seed, iter0, n_iters, mode, update_items, nullptr);
    });
}

int cu2rec_model_loss(const cu2rec_model *m, const cu2rec_csr *ratings, double *sum_abs, double *sum_sq, float *mae,
                      float *rmse) {
    return guarded([&] {
        require(m && ratings, "null argument");
        m->impl.loss(ratings->impl, sum_abs, sum_sq, mae, rmse, nullptr);
    });
}

int cu2rec_items_delta_pack(const float *Q, const float *item_bias, const float *Q_base, const float *ib_base,
                            int n_cols, int ldq, float *buf, void *stream) {
    return guarded([&] {
        require(Q && item_bias && Q_base && ib_base && buf && n_cols >= 0 && ldq > 0, "bad argument");
        require_device();
        if (n_cols == 0) return;
        launch_items_delta_pack(Q, item_bias, Q_base, ib_base, n_cols, ldq, buf, as_stream(stream));
        CU2REC_HIP(hipGetLastError());
    });
}

int cu2rec_items_delta_apply_overlapped(float *Q, float *item_bias, float *Q_base, float *ib_base, const float *Q_snap,
                                        const float *ib_snap, int n_cols, int ldq, const float *buf, float scale,
                                        void *stream) {
    return guarded([&] {
        require(Q && item_bias && Q_base && ib_base && Q_snap && ib_snap && buf && n_cols >= 0 && ldq > 0, "bad argument");
        require_device();
        if (n_cols == 0) return;
        launch_items_delta_apply_overlapped(Q, item_bias, Q_base, ib_base, Q_snap, ib_snap, n_cols, ldq, buf, scale,
                                            as_stream(stream));
        CU2REC_HIP(hipGetLastError());
    });
}

int cu2rec_items_delta_pack_weighted(const float *Q, const float *item_bias, const float *Q_base, const float *ib_base,
                                     const float *item_weight, int n_cols, int ldq, float *buf, void *stream) {
    return guarded([&] {
        require(Q && item_bias && Q_base && ib_base && item_weight && buf && n_cols >= 0 && ldq > 0, "bad argument");
        require_device();
        if (n_cols == 0) return;
        launch_items_delta_pack_weighted(Q, item_bias, Q_base, ib_base, item_weight, n_cols, ldq, buf, as_stream(stream));
        CU2REC_HIP(hipGetLastError());
    });
}

int cu2rec_items_delta_apply(float *Q, float *item_bias, float *Q_base, float *ib_base, int n_cols, int ldq,
                             const float *buf, float scale, void *stream) {
    return guarded([&] {
        require(Q && item_bias && Q_base && ib_base && buf && n_cols >= 0 && ldq > 0, "bad argument");
        require_device();
        if (n_cols == 0) return;
        launch_items_delta_apply(Q, item_bias, Q_base, ib_base, n_cols, ldq, buf, scale, as_stream(stream));
        CU2REC_HIP(hipGetLastError());
    });
}

}  // extern "C"
