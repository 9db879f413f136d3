// The reference's training schedule (training.cu:101-177) on the GPU: train_schedule_core.hpp's loop with HIP events on the
// launch stream as its clock.
#pragma once

#include "device.hpp"
#include "train_schedule_core.hpp"

namespace cu2rec {

struct HipClock {  // the SGD stretch between two loss checks, by events on the stream it is queued on
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    HipClock() {
        CU2REC_HIP(hipEventCreate(&ev_start));
        CU2REC_HIP(hipEventCreate(&ev_stop));
    }
    HipClock(const HipClock &) = delete;
    HipClock &operator=(const HipClock &) = delete;
    ~HipClock() {
        if (ev_start) (void)hipEventDestroy(ev_start);
        if (ev_stop) (void)hipEventDestroy(ev_stop);
    }
    void start(hipStream_t s) { CU2REC_HIP(hipEventRecord(ev_start, s)); }
    void stop(hipStream_t s) { CU2REC_HIP(hipEventRecord(ev_stop, s)); }
    float elapsed_ms() {
        float ms = 0.f;
        CU2REC_HIP(hipEventElapsedTime(&ms, ev_start, ev_stop));
        return ms;
    }
    bool done() {  // has everything up to stop() finished?  (the sharded driver's exchange timers: read without waiting)
        const hipError_t q = hipEventQuery(ev_stop);
        if (q == hipErrorNotReady) return false;
        CU2REC_HIP(q);
        return true;
    }
    void drain() { CU2REC_HIP(hipDeviceSynchronize()); }
};

template <class Sgd, class Loss, class After>
void train_schedule(cu2rec_config &cfg, bool verbose, bool print, double updates_per_iteration, float *losses,
                    cu2rec_train_stats *stats, hipStream_t stream, Sgd sgd, Loss loss, After after_check) {
    train_schedule_with<HipClock>(cfg, verbose, print, updates_per_iteration, losses, stats, stream, sgd, loss, after_check);
}

}  // namespace cu2rec
