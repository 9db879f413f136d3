// bin/predict -- partial fit for one new user against a trained model, then recommendations.
// Drop-in for the reference CLI (predict.cu:72-146):
//
//   bin/predict -c config -i item_bias.csv -g global_bias.csv -q q.csv user_ratings.csv
//
// Q, item_bias and global_bias come from the files bin/mf wrote; only the user's factor row and bias are
// trained (is_train = false).  Unlike the reference, the frozen-item flag really reaches the device (there
// `set_cuda_variables` never uploads is_train, config.cu:24-35, so its Q keeps moving).  The list of items to
// recommend is "every item the user has not rated", whatever order the ratings file is in (the reference
// assumes item-sorted ratings and dereferences end(), predict.cu:49-63).
#include <getopt.h>

#include <algorithm>
#include <cstdio>
#include <iostream>
#include <string>
#include <utility>
#include <vector>

#include "cu2rec.hpp"

namespace {

struct FloatTable {
    float *data = nullptr;
    int rows = 0, cols = 0;
    explicit FloatTable(const std::string &path) { cu2rec::check(cu2rec_read_array(path.c_str(), &data, &rows, &cols)); }
    FloatTable(const FloatTable &) = delete;
    FloatTable &operator=(const FloatTable &) = delete;
    ~FloatTable() { cu2rec_free(data); }
};

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) return 2;  // predict.cu:73-75
    std::string config_path, item_bias_path, global_bias_path, q_path;
    int opt;
    while ((opt = getopt(argc, argv, "c:i:g:q:")) != -1) {
        switch (opt) {
            case 'c': config_path = optarg; break;
            case 'i': item_bias_path = optarg; break;
            case 'g': global_bias_path = optarg; break;
            case 'q': q_path = optarg; break;
            default:
                std::cout << "Unknown option.\n";  // predict.cu:96-98
                return 1;
        }
    }
    if (optind >= argc || config_path.empty() || item_bias_path.empty() || global_bias_path.empty() || q_path.empty()) {
        std::cerr << "usage: predict -c config -i item_bias.csv -g global_bias.csv -q q.csv user_ratings.csv\n";
        return 2;
    }
    try {
        cu2rec_config cfg;
        cu2rec::check(cu2rec_config_default(&cfg));
        cu2rec::check(cu2rec_config_read(config_path.c_str(), &cfg));  // predict.cu:103-104
        cfg.is_train = 0;                                               // predict.cu:105

        const FloatTable item_bias(item_bias_path), gb(global_bias_path), Q(q_path);  // predict.cu:110-113
        const int n_items = Q.rows;
        if (n_items <= 0 || Q.cols != cfg.n_factors) throw std::runtime_error("q file does not have n_factors columns");
        if (item_bias.rows * item_bias.cols != n_items) throw std::runtime_error("item_bias and q disagree on the item count");
        if (gb.rows * gb.cols < 1) throw std::runtime_error("empty global_bias file");
        const float global_bias = gb.data[0];

        // the user's ratings: every record becomes user 0 (predict.cu:117-123)
        cu2rec::HostCsr one = cu2rec::load_ratings(argv[optind]);
        if (one.cols > n_items) throw std::runtime_error("user ratings name an item the model does not have");
        cu2rec::HostCsr user;
        user.rows = 1;
        user.cols = n_items;
        user.nnz = one.nnz;
        user.indptr = {0, one.nnz};
        user.indices = one.indices;
        user.data = one.data;

        cu2rec::CsrHandle d_user(user);
        cu2rec::ModelHandle model(1, n_items, cfg.n_factors, global_bias, nullptr, Q.data, nullptr, item_bias.data);
        std::vector<float> losses(static_cast<size_t>(cfg.total_iterations > 0 ? cfg.total_iterations : 1));
        // one user, frozen items: no two updates can conflict, every mode gives the sequential result
        cu2rec::check(cu2rec_train(d_user.h, d_user.h, &cfg, model.h, CU2REC_SGD_SERIAL, 1, losses.data(), nullptr));  // :126

        std::vector<float> P(cfg.n_factors);
        float user_bias = 0.f;
        cu2rec::check(cu2rec_model_download(model.h, P.data(), nullptr, &user_bias, nullptr));

        // predict_ratings, predict.cu:18-30
        std::vector<float> predictions(n_items);
        for (int i = 0; i < n_items; ++i) {
            float pred = global_bias + user_bias + item_bias.data[i];
            for (int f = 0; f < cfg.n_factors; ++f) pred += Q.data[static_cast<size_t>(i) * cfg.n_factors + f] * P[f];
            predictions[i] = pred;
        }
        std::cout << "Predictions: " << "\n" << "[";  // print_predictions, predict.cu:32-39
        for (int i = 0; i < n_items; ++i) std::cout << predictions[i] << ", ";
        std::cout << "]\n";

        std::vector<char> rated(n_items, 0);
        for (int item : user.indices) rated[item] = 1;
        std::vector<std::pair<float, int>> items;  // get_recommendations, predict.cu:50-65
        for (int i = 0; i < n_items; ++i)
            if (!rated[i]) items.emplace_back(predictions[i], i);
        std::stable_sort(items.begin(), items.end(),
                         [](const std::pair<float, int> &l, const std::pair<float, int> &r) { return l.first > r.first; });
        std::cout << "Recommendations:" << std::endl;  // print_recommendations, predict.cu:67-72
        for (size_t i = 0; i < items.size(); ++i)
            std::printf("Rank: %d\tItem: %d\tEstimated rating: %f\n", static_cast<int>(i) + 1, items[i].second, items[i].first);
    } catch (const std::exception &e) {
        std::cerr << "predict: " << e.what() << "\n";
        return 3;
    }
    return 0;
}
