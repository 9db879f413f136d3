// bin/mf -- trains a matrix-factorisation model on the GPU and writes its five components.
// Drop-in for the reference CLI (mf.cu:16-99): same arguments, same stdout lines, same output
// files; built on the C ABI of libcu2rec_amd.
//
//   bin/mf [-c config] [-m hogwild|ordered|blocksolve|serial|pingpong] train.csv test.csv
#include <getopt.h>
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "cu2rec.hpp"

int main(int argc, char **argv) {
    if (argc < 2) return -1;  // mf.cu:17-19
    std::string config_path;
    int mode = CU2REC_SGD_HOGWILD;
    int opt;
    while ((opt = getopt(argc, argv, "c:m:")) != -1) {
        switch (opt) {
            case 'c':
                config_path = optarg;
                break;
            case 'm':
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
        std::cerr << "usage: mf [-c config] [-m hogwild|ordered|blocksolve|serial|pingpong] train.csv test.csv\n";
        return -1;
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
