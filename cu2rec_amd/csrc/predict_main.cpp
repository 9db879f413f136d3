// bin/predict -- partial fit for new users against a trained model, then recommendations.
// Drop-in for the reference CLI (predict.cu:72-146):
//
//   bin/predict -c config -i item_bias.csv -g global_bias.csv -q q.csv [-u] [-k K] user_ratings.csv
//
// Q, item_bias and global_bias come from the files bin/mf wrote; only the users' factor rows and biases are
// trained (is_train = false).  Unlike the reference, the frozen-item flag really reaches the device (there
// `set_cuda_variables` never uploads is_train, config.cu:24-35, so its Q keeps moving).  The list of items to
// recommend is "every item the user has not rated", whatever order the ratings file is in (the reference
// assumes item-sorted ratings and dereferences end(), predict.cu:49-63).
// Default = the reference: EVERY record of the file belongs to the one new user, whatever its userId column says
// (predict.cu:119-121).  -u: a file of MANY users -- each distinct userId is a new user of its own; all of them are
// fitted in one batch (frozen items: no update crosses users, so every fit equals the one-user result), scored with one
// dense product on the matrix cores and ranked by one segmented sort on the device (cu2rec_model_scores /
// cu2rec_model_recommend); output is one block per user, headed "User: <id>".  -k K limits the recommendations listed.
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
    bool many_users = false;
    int top_k = 0;  // 0: every unrated item, as the reference prints
    int opt;
    while ((opt = getopt(argc, argv, "c:i:g:q:uk:")) != -1) {
        switch (opt) {
            case 'u': many_users = true; break;
            case 'k': top_k = std::max(0, std::atoi(optarg)); break;
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
        std::cerr << "usage: predict -c config -i item_bias.csv -g global_bias.csv -q q.csv [-u] [-k K] user_ratings.csv\n";
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

        // the ratings: every record becomes user 0 (predict.cu:117-123), or with -u one new user per distinct userId.
        // Read as raw records: the file need not be sorted by user, and its userId column need not start at 1.
        cu2rec_ratings *raw = nullptr;
        cu2rec::check(cu2rec_ratings_read_csv(argv[optind], &raw));
        int n = 0, file_rows = 0, file_cols = 0;
        float file_gb = 0.f;
        const int *ru = nullptr, *ri = nullptr;
        const float *rr = nullptr;
        cu2rec::check(cu2rec_ratings_info(raw, &n, &file_rows, &file_cols, &file_gb));
        cu2rec::check(cu2rec_ratings_view(raw, &ru, &ri, &rr));
        if (file_cols > n_items) throw std::runtime_error("user ratings name an item the model does not have");
        std::vector<int> order(n);
        for (int k = 0; k < n; ++k) order[k] = k;
        std::vector<int> user_ids;  // 1-based ids as in the file, one per new user
        if (many_users) {
            std::stable_sort(order.begin(), order.end(), [&](int l, int r) { return ru[l] < ru[r]; });
            for (int k = 0; k < n; ++k)
                if (user_ids.empty() || user_ids.back() != ru[order[k]] + 1) user_ids.push_back(ru[order[k]] + 1);
        } else {
            user_ids.push_back(1);
        }
        cu2rec::HostCsr users;
        users.rows = static_cast<int>(user_ids.size());
        users.cols = n_items;
        users.nnz = n;
        users.indptr.assign(1, 0);
        for (int k = 0; k < n; ++k) {
            if (many_users && k > 0 && ru[order[k]] != ru[order[k - 1]]) users.indptr.push_back(k);
            users.indices.push_back(ri[order[k]]);
            users.data.push_back(rr[order[k]]);
        }
        users.indptr.push_back(n);
        cu2rec_ratings_free(raw);
        if (static_cast<int>(users.indptr.size()) != users.rows + 1) throw std::runtime_error("internal: bad user grouping");

        cu2rec::CsrHandle d_users(users);
        // every new user starts from the reference's seed-42 draw of ONE user (training.cu:28,54): the same prefix
        const int B = users.rows, f = cfg.n_factors;
        std::vector<float> P0(static_cast<size_t>(B) * f), ub0(B), one_p(f), one_ub(1);
        cu2rec::check(cu2rec_init_normal(one_p.data(), one_p.size(), f, 0.f, 1.f, 42));
        cu2rec::check(cu2rec_init_normal(one_ub.data(), 1, f, 0.f, 1.f, 42));
        for (int u = 0; u < B; ++u) {
            std::copy(one_p.begin(), one_p.end(), P0.begin() + static_cast<size_t>(u) * f);
            ub0[u] = one_ub[0];
        }
        cu2rec::ModelHandle model(B, n_items, f, global_bias, P0.data(), Q.data, ub0.data(), item_bias.data);
        std::vector<float> losses(static_cast<size_t>(cfg.total_iterations > 0 ? cfg.total_iterations : 1));
        // frozen items: no two updates can conflict, every mode gives the sequential result; Hogwild runs all of a call's
        // iterations for all users in one launch
        cu2rec::check(cu2rec_train(d_users.h, d_users.h, &cfg, model.h, B == 1 ? CU2REC_SGD_SERIAL : CU2REC_SGD_HOGWILD, 1,
                                   losses.data(), nullptr));  // predict.cu:126

        // predict_ratings (predict.cu:18-30) and get_recommendations (predict.cu:50-65), on the device
        std::vector<float> predictions(static_cast<size_t>(B) * n_items);
        cu2rec::check(cu2rec_model_scores_host(model.h, predictions.data()));
        const int k_list = top_k > 0 ? std::min(top_k, n_items) : n_items;
        std::vector<int> rec_items(static_cast<size_t>(B) * k_list);
        std::vector<float> rec_scores(static_cast<size_t>(B) * k_list);
        cu2rec::check(cu2rec_model_recommend(model.h, d_users.h, k_list, rec_items.data(), rec_scores.data()));
        for (int u = 0; u < B; ++u) {
            if (many_users) std::cout << "User: " << user_ids[u] << "\n";
            std::cout << "Predictions: " << "\n" << "[";  // print_predictions, predict.cu:32-39
            for (int i = 0; i < n_items; ++i) std::cout << predictions[static_cast<size_t>(u) * n_items + i] << ", ";
            std::cout << "]\n";
            std::cout << "Recommendations:" << std::endl;  // print_recommendations, predict.cu:67-72
            for (int j = 0; j < k_list; ++j) {
                const int item = rec_items[static_cast<size_t>(u) * k_list + j];
                if (item < 0) break;
                std::printf("Rank: %d\tItem: %d\tEstimated rating: %f\n", j + 1, item, rec_scores[static_cast<size_t>(u) * k_list + j]);
            }
        }
    } catch (const std::exception &e) {
        std::cerr << "predict: " << e.what() << "\n";
        return 3;
    }
    return 0;
}
