// Shared plumbing of libcu2rec_amd: status codes, per-thread error text, HIP call checking.
// The reference's CHECK_CUDA (util.h:27-34) throws std::runtime_error with file:line; here
// the throw is caught at the C boundary and turned into a status + cu2rec_last_error().
#pragma once

#include <cstdio>
#include <exception>
#include <new>
#include <stdexcept>
#include <string>

#include "../../include/cu2rec_amd.h"

namespace cu2rec {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &what) : std::runtime_error(what), code(c) {}
};

void set_last_error(const std::string &msg);

[[noreturn]] inline void fail(int code, const std::string &msg) { throw Error(code, msg); }

inline void require(bool ok, const char *what) {
    if (!ok) fail(CU2REC_EINVAL, what);
}

// Runs `body`, mapping exceptions to status codes. Every extern "C" entry point goes through it.
template <class F>
int guarded(F &&body) noexcept {
    try {
        body();
        return CU2REC_OK;
    } catch (const Error &e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc &) {
        set_last_error("out of host memory");
        return CU2REC_ENOMEM;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return CU2REC_EINVAL;
    }
}

}  // namespace cu2rec
