// bin/mf -- trains a matrix-factorisation model on the GPU and writes its five components.
// Drop-in for the reference CLI (mf.cu:16-99): same arguments, same stdout lines, same output
// files; built on the C ABI of libcu2rec_amd.
//
//   bin/mf [-c config] [-m blocksolve|ordered|hogwild|serial|pingpong] [-g gpus] [-s sync_every]
//          [-w adaptive|mean|weighted|sum] train.csv test.csv
// Default mode: blocksolve -- mf_sequential.cu's result within float rounding (the mode bench.py certifies against the
// 1e-4 RMSE bar); n_factors above 252, where block-solve is not compiled, falls back to `ordered` (the same result bit
// for bit).  `-m hogwild` opts into sgd.cu's own racy semantics (fastest, 1e-3 away from the sequential result while
// the model is still moving).
// -g N (N > 1): one process per GPU -- the program forks N - 1 more ranks before anything touches a GPU, rank r takes
// device r and a contiguous range of users, the ranks train through cu2rec_train_sharded (RCCL all-reduce of the item
// deltas every -s iterations, default one epoch), rank 0 prints the lines and writes the five files.
#include <getopt.h>
#include <hip/hip_runtime_api.h>
#include <signal.h>
#include <sys/prctl.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "cu2rec.hpp"

namespace {

bool g_mode_given = false;  // -m on the command line

// the default mode is block-solve; where it is not compiled (n_factors > 252) the default becomes `ordered`: the same
// sequential result, bit for bit instead of within float rounding
int effective_mode(int mode, int n_factors) {
    return (!g_mode_given && mode == CU2REC_SGD_BLOCKSOLVE && n_factors > 252) ? CU2REC_SGD_ORDERED : mode;
}

// all of fd or an exception
void write_all(int fd, const void *buf, size_t n) {
    const char *p = static_cast<const char *>(buf);
    while (n) {
        const ssize_t k = ::write(fd, p, n);
        if (k <= 0) throw std::runtime_error("pipe write failed");
        p += k;
        n -= static_cast<size_t>(k);
    }
}

void read_all(int fd, void *buf, size_t n) {
    char *p = static_cast<char *>(buf);
    while (n) {
        const ssize_t k = ::read(fd, p, n);
        if (k <= 0) throw std::runtime_error("pipe read failed (a rank died?)");
        p += k;
        n -= static_cast<size_t>(k);
    }
}

// One rank of `bin/mf -g N`.  to_child / from_child: rank 0's pipes to and from every other rank; up / down: this
// rank's pipes to and from rank 0.
int run_rank(int rank, int nranks, int mode, int sync_every, int merge, const std::string &config_path,
             const std::string &train_path, const std::string &test_path, const std::vector<int> &to_child,
             const std::vector<int> &from_child, int up, int down) {
#ifdef CU2REC_TEST_HOOKS
    // fault injection for the failure-propagation test (tests/test_host_abi.py), before anything touches a GPU -- compiled into
    // the test binary build/test/mf_hooks only (make -C cu2rec_amd/csrc test-hooks), never into bin/mf:
    // CU2REC_TEST_RANK_EXIT="r:code" makes rank r end with that exit code, CU2REC_TEST_RANK_HANG="r" makes rank r sit still
    // the way a rank inside a collective whose peer died would
    if (const char *env = std::getenv("CU2REC_TEST_RANK_EXIT")) {
        int r = -1, code = 0;
        if (std::sscanf(env, "%d:%d", &r, &code) == 2 && r == rank) _exit(code);
    }
    if (const char *env = std::getenv("CU2REC_TEST_RANK_HANG"))
        if (std::atoi(env) == rank) std::this_thread::sleep_for(std::chrono::seconds(120));
#endif
    if (cu2rec_device_count() < nranks) throw std::runtime_error("fewer HIP devices than ranks (-g)");
    cu2rec::check(cu2rec_set_device(rank));
    if (rank == 0) {
        size_t free_bytes = 0, total_bytes = 0;  // mf.cu:33-37
        if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess) throw std::runtime_error("hipMemGetInfo failed");
        std::printf("Free memory: %ld\n\n", static_cast<long>(free_bytes));
    }
    cu2rec::HostCsr train = cu2rec::load_ratings(train_path);  // every rank parses the files and keeps its slice
    cu2rec::HostCsr test = cu2rec::load_ratings(test_path);
    if (train.nnz == 0) throw std::runtime_error("no ratings read from " + train_path);
    if (test.rows > train.rows || test.cols > train.cols) throw std::runtime_error("the test file names users / items the training file does not have");
    cu2rec_config cfg;
    cu2rec::check(cu2rec_config_default(&cfg));
    if (!config_path.empty()) cu2rec::check(cu2rec_config_read(config_path.c_str(), &cfg));
    if (rank == 0) cu2rec::check(cu2rec_config_print(&cfg));
    mode = effective_mode(mode, cfg.n_factors);

    std::vector<int> bounds(static_cast<size_t>(nranks) + 1);
    cu2rec::check(cu2rec_shard_plan(train.rows, nranks, bounds.data()));
    const int u0 = bounds[rank], u1 = bounds[rank + 1], f = cfg.n_factors;
    test.indptr.resize(static_cast<size_t>(train.rows) + 1, test.nnz);  // a test file may name fewer users (mf.cu:50-51)
    test.rows = train.rows;
    auto slice = [&](const cu2rec::HostCsr &m) {
        cu2rec::HostCsr s;
        s.rows = u1 - u0;
        s.cols = train.cols;
        s.indptr.resize(static_cast<size_t>(s.rows) + 1);
        int off = 0;
        cu2rec::check(cu2rec_csr_slice(m.indptr.data(), m.rows, u0, u1, s.indptr.data(), &off, &s.nnz));
        s.indices.assign(m.indices.begin() + off, m.indices.begin() + off + s.nnz);
        s.data.assign(m.data.begin() + off, m.data.begin() + off + s.nnz);
        s.global_bias = train.global_bias;
        return s;
    };
    const cu2rec::HostCsr tr = slice(train), te = slice(test);
    // every rank draws the reference's seed-42 initialisation and keeps its slice (training.cu:28,54)
    std::vector<float> P0(static_cast<size_t>(train.rows) * f), ub0(train.rows);
    cu2rec::check(cu2rec_init_normal(P0.data(), P0.size(), f, 0.f, 1.f, 42));
    cu2rec::check(cu2rec_init_normal(ub0.data(), ub0.size(), f, 0.f, 1.f, 42));
    cu2rec::CsrHandle d_train(tr), d_test(te);
    cu2rec::ModelHandle model(u1 - u0, train.cols, f, train.global_bias, P0.data() + static_cast<size_t>(u0) * f, nullptr,
                              ub0.data() + u0, nullptr);

    unsigned char uid[128] = {0};
    if (rank == 0) {
        cu2rec::check(cu2rec_comm_unique_id(uid));
        for (int r = 1; r < nranks; ++r) write_all(to_child[r], uid, sizeof(uid));
    } else {
        read_all(down, uid, sizeof(uid));
    }
    cu2rec_comm *comm = nullptr;
    cu2rec::check(cu2rec_comm_create(uid, rank, nranks, &comm));
    cu2rec_shard_options opt{sync_every, merge};
    cu2rec_shard_job *job = nullptr;
    cu2rec::check(cu2rec_shard_job_create(comm, model.h, d_train.h, u0, &opt, &job));
    std::vector<float> losses(static_cast<size_t>(cfg.total_iterations > 0 ? cfg.total_iterations : 1));
    cu2rec::check(cu2rec_train_sharded(job, d_test.h, &cfg, mode, 1, losses.data(), nullptr));

    std::vector<float> P(static_cast<size_t>(u1 - u0) * f), ub(u1 - u0), Q, ib;
    if (rank == 0) {
        Q.resize(static_cast<size_t>(train.cols) * f);
        ib.resize(train.cols);
    }
    cu2rec::check(cu2rec_model_download(model.h, P.data(), rank == 0 ? Q.data() : nullptr, ub.data(), rank == 0 ? ib.data() : nullptr));
    cu2rec_shard_job_destroy(job);
    cu2rec_comm_destroy(comm);
    if (rank != 0) {  // the user side goes to rank 0 through the pipe
        write_all(up, P.data(), P.size() * sizeof(float));
        write_all(up, ub.data(), ub.size() * sizeof(float));
        return 0;
    }
    std::vector<float> P_all(static_cast<size_t>(train.rows) * f), ub_all(train.rows);
    std::copy(P.begin(), P.end(), P_all.begin());
    std::copy(ub.begin(), ub.end(), ub_all.begin());
    for (int r = 1; r < nranks; ++r) {
        read_all(from_child[r], P_all.data() + static_cast<size_t>(bounds[r]) * f, static_cast<size_t>(bounds[r + 1] - bounds[r]) * f * sizeof(float));
        read_all(from_child[r], ub_all.data() + bounds[r], static_cast<size_t>(bounds[r + 1] - bounds[r]) * sizeof(float));
    }
    std::string parent_dir = ".", filename = train_path;  // mf.cu:65-87
    const size_t slash = train_path.find_last_of('/');
    if (slash != std::string::npos) {
        parent_dir = train_path.substr(0, slash);
        filename = train_path.substr(slash + 1);
    }
    const std::string base = filename.substr(0, filename.find_last_of('.'));
    const float gb = train.global_bias;
    cu2rec::check(cu2rec_write_component(parent_dir.c_str(), base.c_str(), "p", P_all.data(), train.rows, f, f));
    cu2rec::check(cu2rec_write_component(parent_dir.c_str(), base.c_str(), "q", Q.data(), train.cols, f, f));
    cu2rec::check(cu2rec_write_component(parent_dir.c_str(), base.c_str(), "user_bias", ub_all.data(), train.rows, 1, f));
    cu2rec::check(cu2rec_write_component(parent_dir.c_str(), base.c_str(), "item_bias", ib.data(), train.cols, 1, f));
    cu2rec::check(cu2rec_write_component(parent_dir.c_str(), base.c_str(), "global_bias", &gb, 1, 1, f));
    return 0;
}

// bin/mf -g N: forks the other ranks BEFORE any HIP call (a process that has initialised the GPU must not fork into
// another GPU user), wires the pipes, runs rank 0 here and collects the children's exit codes.
// A rank that fails must not leave the others sitting in a collective: every child dies with its parent (PR_SET_PDEATHSIG),
// and while rank 0 trains a watcher thread reaps the children -- the first one that ends abnormally makes the program kill
// the rest and exit non-zero at once (fresh exits only: nothing here re-executes a process that has touched a GPU).
int run_multi_gpu(int nranks, int mode, int sync_every, int merge, const std::string &config_path, const std::string &train_path,
                  const std::string &test_path) {
    std::vector<int> to_child(nranks, -1), from_child(nranks, -1);
    std::vector<pid_t> pids(nranks, 0);
    const pid_t parent = getpid();
    for (int r = 1; r < nranks; ++r) {
        int down[2], up[2];
        if (pipe(down) != 0 || pipe(up) != 0) return 2;
        const pid_t pid = fork();
        if (pid < 0) return 2;
        if (pid == 0) {
            prctl(PR_SET_PDEATHSIG, SIGKILL);
            if (getppid() != parent) _exit(2);  // the parent was gone before the request took effect
            close(down[1]);
            close(up[0]);
            for (int q = 1; q < r; ++q) {
                close(to_child[q]);
                close(from_child[q]);
            }
            int rc = 0;
            try {
                rc = run_rank(r, nranks, mode, sync_every, merge, config_path, train_path, test_path, {}, {}, up[1], down[0]);
            } catch (const std::exception &e) {
                std::cerr << "mf (rank " << r << "): " << e.what() << "\n";
                rc = 2;
            }
            _exit(rc);
        }
        close(down[0]);
        close(up[1]);
        to_child[r] = down[1];
        from_child[r] = up[0];
        pids[r] = pid;
    }
    std::vector<std::atomic<int>> reaped(nranks);  // 0: running, 1: ended cleanly
    for (auto &v : reaped) v.store(0);
    std::atomic<bool> finished{false};
    auto kill_children = [&] {
        for (int r = 1; r < nranks; ++r)
            if (!reaped[r].load()) kill(pids[r], SIGKILL);
    };
    std::thread watcher([&] {
        while (!finished.load()) {
            for (int r = 1; r < nranks; ++r) {
                if (reaped[r].load()) continue;
                int status = 0;
                const pid_t got = waitpid(pids[r], &status, WNOHANG);
                if (got != pids[r]) continue;
                if (WIFEXITED(status) && WEXITSTATUS(status) == 0) {
                    reaped[r].store(1);
                    continue;
                }
                std::cerr << "mf: rank " << r << " ended abnormally (" << (WIFSIGNALED(status) ? "signal " : "exit code ")
                          << (WIFSIGNALED(status) ? WTERMSIG(status) : WEXITSTATUS(status)) << "); stopping the other ranks\n";
                reaped[r].store(1);
                kill_children();
                for (int q = 1; q < nranks; ++q)
                    if (!reaped[q].load()) (void)waitpid(pids[q], &status, 0);  // SIGKILL: at once
                std::fflush(stdout);  // the TRAIN: / TEST: lines rank 0 has printed so far are not lost with the process
                std::fflush(stderr);
                _exit(3);  // rank 0 (this process) may be inside a collective that will never complete
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    });
    int rc = 0;
    try {
        rc = run_rank(0, nranks, mode, sync_every, merge, config_path, train_path, test_path, to_child, from_child, -1, -1);
    } catch (const std::exception &e) {
        std::cerr << "mf: " << e.what() << "\n";
        rc = 2;
    }
    finished.store(true);
    watcher.join();
    if (rc != 0) kill_children();  // rank 0 failed: the others may be waiting for it in a collective
    for (int r = 1; r < nranks; ++r) {
        close(to_child[r]);
        close(from_child[r]);
        if (reaped[r].load()) continue;
        int status = 0;
        if (waitpid(pids[r], &status, 0) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) rc = rc ? rc : 2;
    }
    return rc;
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) return -1;  // mf.cu:17-19
    std::string config_path;
    int mode = CU2REC_SGD_BLOCKSOLVE;  // the mode that meets the north star's 1e-4 bar; -m hogwild is the opt-in
    int gpus = 1, sync_every = 0, merge = CU2REC_MERGE_ADAPTIVE;
    int opt;
    while ((opt = getopt(argc, argv, "c:m:g:s:w:")) != -1) {
        switch (opt) {
            case 'g':
                gpus = std::atoi(optarg);
                break;
            case 's':
                sync_every = std::atoi(optarg);
                break;
            case 'w':
                merge = std::strcmp(optarg, "mean") == 0       ? CU2REC_MERGE_MEAN
                        : std::strcmp(optarg, "sum") == 0      ? CU2REC_MERGE_SUM
                        : std::strcmp(optarg, "adaptive") == 0 ? CU2REC_MERGE_ADAPTIVE
                                                               : CU2REC_MERGE_WEIGHTED;
                break;
            case 'c':
                config_path = optarg;
                break;
            case 'm':
                g_mode_given = true;
                mode = std::strcmp(optarg, "serial") == 0    ? CU2REC_SGD_SERIAL
                       : std::strcmp(optarg, "ordered") == 0 ? CU2REC_SGD_ORDERED
                       : std::strcmp(optarg, "pingpong") == 0 ? CU2REC_SGD_PINGPONG
                       : std::strcmp(optarg, "blocksolve") == 0 ? CU2REC_SGD_BLOCKSOLVE
                                                             : CU2REC_SGD_HOGWILD;
                break;
            default:
                std::cout << "Unknown option.\n";  // mf.cu:27-29
                return 1;
        }
    }
    if (optind + 2 > argc) {
        std::cerr << "usage: mf [-c config] [-m blocksolve|ordered|hogwild|serial|pingpong] [-g gpus] [-s sync_every] "
                     "[-w adaptive|mean|weighted|sum] train.csv test.csv\n";
        return -1;
    }
    if (gpus > 1) {
        if (mode == CU2REC_SGD_PINGPONG) {
            std::cerr << "mf: -m pingpong is single-GPU only\n";
            return 1;
        }
        return run_multi_gpu(gpus, mode, sync_every, merge, config_path, argv[optind], argv[optind + 1]);
    }
    try {
        if (cu2rec_device_count() < 1) throw std::runtime_error("no HIP device available");
        size_t free_bytes = 0, total_bytes = 0;  // mf.cu:33-37
        if (hipMemGetInfo(&free_bytes, &total_bytes) != hipSuccess) throw std::runtime_error("hipMemGetInfo failed");
        std::printf("Free memory: %ld\n\n", static_cast<long>(free_bytes));

        const std::string train_path = argv[optind++];
        const cu2rec::HostCsr train = cu2rec::load_ratings(train_path);  // mf.cu:40-44
        const std::string test_path = argv[optind++];
        const cu2rec::HostCsr test = cu2rec::load_ratings(test_path);    // mf.cu:47-51
        if (train.nnz == 0) throw std::runtime_error("no ratings read from " + train_path);

        cu2rec_config cfg;  // mf.cu:54-57
        cu2rec::check(cu2rec_config_default(&cfg));
        if (!config_path.empty()) cu2rec::check(cu2rec_config_read(config_path.c_str(), &cfg));
        cu2rec::check(cu2rec_config_print(&cfg));
        mode = effective_mode(mode, cfg.n_factors);

        cu2rec::CsrHandle d_train(train), d_test(test);
        cu2rec::ModelHandle model(train.rows, train.cols, cfg.n_factors, train.global_bias);
        std::vector<float> losses(static_cast<size_t>(cfg.total_iterations > 0 ? cfg.total_iterations : 1));
        cu2rec::check(cu2rec_train(d_train.h, d_test.h, &cfg, model.h, mode, 1, losses.data(), nullptr));  // mf.cu:61

        std::vector<float> P(static_cast<size_t>(train.rows) * cfg.n_factors), Q(static_cast<size_t>(train.cols) * cfg.n_factors);
        std::vector<float> user_bias(train.rows), item_bias(train.cols);
        cu2rec::check(cu2rec_model_download(model.h, P.data(), Q.data(), user_bias.data(), item_bias.data()));

        // mf.cu:65-87: outputs go next to the training file, named <base>_f<F>_<component>.csv
        std::string parent_dir = ".", filename = train_path;
        const size_t slash = train_path.find_last_of('/');
        if (slash != std::string::npos) {
            parent_dir = train_path.substr(0, slash);
            filename = train_path.substr(slash + 1);
        }
        const std::string base = filename.substr(0, filename.find_last_of('.'));
        const float gb = train.global_bias;
        const int f = cfg.n_factors;
        cu2rec::check(cu2rec_write_component(parent_dir.c_str(), base.c_str(), "p", P.data(), train.rows, f, f));
        cu2rec::check(cu2rec_write_component(parent_dir.c_str(), base.c_str(), "q", Q.data(), train.cols, f, f));
        cu2rec::check(cu2rec_write_component(parent_dir.c_str(), base.c_str(), "user_bias", user_bias.data(), train.rows, 1, f));
        cu2rec::check(cu2rec_write_component(parent_dir.c_str(), base.c_str(), "item_bias", item_bias.data(), train.cols, 1, f));
        cu2rec::check(cu2rec_write_component(parent_dir.c_str(), base.c_str(), "global_bias", &gb, 1, 1, f));
    } catch (const std::exception &e) {
        std::cerr << "mf: " << e.what() << "\n";
        return 2;
    }
    return 0;
}
