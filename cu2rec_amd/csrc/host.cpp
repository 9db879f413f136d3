// Host-side substrate of libcu2rec_amd: config schema + file format, ratings reader, CSR builder,
// libstdc++-exact normal initialisation, CSV writers, shard planner.  No GPU is touched here, so
// these entry points also work on a machine without one.
//
// Behaviour follows the reference's host code (config.cu, util.cu, mf.cu) -- cited per function --
// but nothing is shared with it structurally: one pass hand-rolled parser over the whole file
// instead of iostream extraction, explicit validation where the reference would loop forever or
// read out of bounds.
#include <sys/stat.h>

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <sstream>
#include <thread>
#include <vector>

#include "common.hpp"
#include "sampler.hpp"

namespace cu2rec {

namespace {
thread_local std::string g_last_error;
}

void set_last_error(const std::string &msg) { g_last_error = msg; }

struct RatingsFile {  // COO as read, 0-based ids (util.h:19-24 Rating, as three columns)
    std::vector<int> user, item;
    std::vector<float> rating;
    int rows = 0, cols = 0;
    float global_bias = 0.f;
};

namespace {

std::string slurp(const char *path) {
    std::ifstream in(path, std::ios::binary);
    if (!in.is_open()) fail(CU2REC_EIO, std::string("cannot open ") + path);
    std::string text;
    in.seekg(0, std::ios::end);
    const std::streamoff size = in.tellg();
    in.seekg(0, std::ios::beg);
    if (size > 0) {
        text.resize(static_cast<size_t>(size));
        in.read(&text[0], size);
        text.resize(static_cast<size_t>(in.gcount()));
    }
    return text;
}

inline bool is_space(char c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t' || c == '\v' || c == '\f'; }

struct Cursor {
    const char *p, *end;
    void skip_space() {
        while (p < end && is_space(*p)) ++p;
    }
    // `stream >> int`: optional sign then digits
    bool read_int(int &out) {
        skip_space();
        const char *q = p;
        if (q < end && *q == '+') ++q;
        auto res = std::from_chars(q, end, out);
        if (res.ec != std::errc()) return false;
        p = res.ptr;
        return true;
    }
    // `stream >> char`: next non-space character, whatever it is (the reference never checks for ',')
    bool read_char() {
        skip_space();
        if (p >= end) return false;
        ++p;
        return true;
    }
    bool read_float(float &out) {
        skip_space();
        const char *q = p;
        if (q < end && *q == '+') ++q;
        auto res = std::from_chars(q, end, out, std::chars_format::general);
        if (res.ec != std::errc()) return false;
        p = res.ptr;
        return true;
    }
};

}  // namespace

struct ParsedChunk {
    std::vector<int> user, item;
    std::vector<float> rating;
    int rows = 0, cols = 0;
    bool clean = true;  // every record sat alone on its line and the chunk was consumed to its end
};

// One record per line, strictly: used by the threaded reader, which may only split the text at line starts.
// Anything else (records spanning lines, trailing junk) marks the chunk unclean and the caller falls back to
// the sequential reader, whose behaviour is the reference's.
static void parse_lines(const char *begin, const char *end, ParsedChunk &out) {
    Cursor cur{begin, end};
    int u, i;
    float r;
    while (true) {
        cur.skip_space();
        if (cur.p >= cur.end) return;
        const char *line = cur.p;
        if (!(cur.read_int(u) && cur.read_char() && cur.read_int(i) && cur.read_char() && cur.read_float(r))) {
            out.clean = false;
            return;
        }
        for (const char *q = line; q < cur.p; ++q)
            if (*q == '\n') {
                out.clean = false;
                return;
            }
        while (cur.p < cur.end && (*cur.p == ' ' || *cur.p == '\t' || *cur.p == '\r')) ++cur.p;
        if (cur.p < cur.end && *cur.p != '\n') {
            out.clean = false;
            return;
        }
        out.user.push_back(u - 1);
        out.item.push_back(i - 1);
        out.rating.push_back(r);
        out.rows = std::max(out.rows, u);
        out.cols = std::max(out.cols, i);
    }
}

static bool read_ratings_threaded(const char *begin, const char *end, unsigned n_threads, RatingsFile &out) {
    std::vector<const char *> cut(n_threads + 1, end);
    cut[0] = begin;
    for (unsigned k = 1; k < n_threads; ++k) {
        const char *p = begin + (static_cast<size_t>(end - begin) * k) / n_threads;
        while (p < end && *p != '\n') ++p;
        cut[k] = p < end ? p + 1 : end;
    }
    std::vector<ParsedChunk> chunks(n_threads);
    std::vector<std::thread> pool;
    for (unsigned k = 0; k < n_threads; ++k)
        pool.emplace_back([&, k] {
            chunks[k].user.reserve(static_cast<size_t>(cut[k + 1] - cut[k]) / 12 + 16);
            parse_lines(cut[k], cut[k + 1], chunks[k]);
        });
    for (auto &t : pool) t.join();
    size_t total = 0;
    for (const auto &c : chunks) {
        if (!c.clean) return false;
        total += c.user.size();
    }
    out.user.reserve(total);
    out.item.reserve(total);
    out.rating.reserve(total);
    for (const auto &c : chunks) {
        out.user.insert(out.user.end(), c.user.begin(), c.user.end());
        out.item.insert(out.item.end(), c.item.begin(), c.item.end());
        out.rating.insert(out.rating.end(), c.rating.begin(), c.rating.end());
        out.rows = std::max(out.rows, c.rows);
        out.cols = std::max(out.cols, c.cols);
    }
    double sum = 0.0;  // in file order, like the reference's running sum (util.cu:36)
    for (float r : out.rating) sum += r;
    out.global_bias = static_cast<float>(sum / (1.0 * out.user.size()));
    return true;
}

// readCSV, util.cu:17-45
static RatingsFile read_ratings_csv(const char *path) {
    const std::string text = slurp(path);
    Cursor cur{text.data(), text.data() + text.size()};
    // ratingsFile.ignore(1000, '\n'): drop the header line (at most 1000 characters)
    for (int k = 0; k < 1000 && cur.p < cur.end; ++k)
        if (*cur.p++ == '\n') break;
    RatingsFile out;
    // large files: parse line-aligned chunks on all host cores (ML-20M: 20 M records)
    const size_t body = static_cast<size_t>(cur.end - cur.p);
    unsigned n_threads = std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 64u);
    if (const char *env = std::getenv("CU2REC_READER_THREADS")) n_threads = std::max(1, std::atoi(env));
    n_threads = static_cast<unsigned>(std::min<size_t>(n_threads, body / (1u << 20) + 1));
    if (n_threads > 1 && read_ratings_threaded(cur.p, cur.end, n_threads, out)) return out;
    out = RatingsFile();
    const size_t guess = static_cast<size_t>(cur.end - cur.p) / 12 + 16;
    out.user.reserve(guess);
    out.item.reserve(guess);
    out.rating.reserve(guess);
    double sum = 0.0;
    int u, i;
    float r;
    // a record is `int char int char float`; the first malformed record ends the file (util.cu:30)
    while (cur.read_int(u) && cur.read_char() && cur.read_int(i) && cur.read_char() && cur.read_float(r)) {
        out.user.push_back(u - 1);
        out.item.push_back(i - 1);
        out.rating.push_back(r);
        out.rows = std::max(out.rows, u);
        out.cols = std::max(out.cols, i);
        sum += r;
    }
    out.global_bias = static_cast<float>(sum / (1.0 * out.user.size()));
    return out;
}

// createSparseMatrix host half, util.cu:152-179
static void build_csr(const RatingsFile &r, int rows, int *indptr, int *indices, float *data) {
    require(rows >= r.rows, "cu2rec_csr_build: rows is smaller than the largest user id in the file");
    const int n = static_cast<int>(r.user.size());
    int next_user = 0;  // first user whose row pointer is not written yet
    for (int k = 0; k < n; ++k) {
        const int u = r.user[k];
        if (u < 0 || r.item[k] < 0) fail(CU2REC_EINVAL, "cu2rec_csr_build: ids must be >= 1 in the file");
        if (u >= rows) fail(CU2REC_EINVAL, "cu2rec_csr_build: user id beyond rows");
        if (u + 1 < next_user) fail(CU2REC_EINVAL, "cu2rec_csr_build: ratings must be sorted by userId");
        while (next_user <= u) indptr[next_user++] = k;  // users without ratings repeat the pointer
        indices[k] = r.item[k];
        data[k] = r.rating[k];
    }
    while (next_user <= rows) indptr[next_user++] = n;
}

// initialize_normal_array, util.cu:124-132 (the standard library does the work there too)
static void init_normal(float *out, size_t size, int n_factors, float mean, float stddev, int seed) {
    std::mt19937 generator(seed);
    std::normal_distribution<float> distribution(mean, stddev / n_factors);
    for (size_t i = 0; i < size; ++i) out[i] = distribution(generator);
}

// writeCSV, util.cu:86-97
static void write_csv(const char *path, const float *data, int rows, int cols) {
    FILE *fp = std::fopen(path, "w");
    if (!fp) fail(CU2REC_EIO, std::string("cannot write ") + path);
    std::vector<char> line(static_cast<size_t>(cols) * 48 + 2);
    for (int i = 0; i < rows; ++i) {
        size_t at = 0;
        for (int j = 0; j < cols; ++j) {
            at += static_cast<size_t>(std::snprintf(&line[at], 48, "%f", data[static_cast<size_t>(i) * cols + j]));
            line[at++] = j + 1 < cols ? ',' : '\n';
        }
        std::fwrite(line.data(), 1, at, fp);
    }
    std::fclose(fp);
}

}  // namespace cu2rec

using namespace cu2rec;

struct cu2rec_ratings {
    RatingsFile file;
};

extern "C" {

const char *cu2rec_last_error(void) { return g_last_error.c_str(); }
int cu2rec_version(void) { return CU2REC_AMD_VERSION; }

int cu2rec_config_default(cu2rec_config *cfg) {
    return guarded([&] {
        require(cfg, "cfg is null");
        *cfg = cu2rec_config{0, 5000, 50, 0.01f, 42, 0.02f, 0.02f, 0.02f, 0.02f, 1, 32, 500, 2.0f, 0.2f};  // config.h:23-51
    });
}

int cu2rec_config_read(const char *path, cu2rec_config *cfg) {  // config.cu:7-13
    return guarded([&] {
        require(path && cfg, "null argument");
        std::ifstream in(path);
        if (!in.is_open()) fail(CU2REC_EIO, std::string("cannot open config ") + path);
        cu2rec_config c = *cfg;
        in >> c.cur_iterations >> c.total_iterations >> c.n_factors >> c.learning_rate >> c.seed >> c.P_reg >>
            c.Q_reg >> c.user_bias_reg >> c.item_bias_reg;
        if (in.fail()) fail(CU2REC_EIO, std::string("config needs 9 whitespace separated fields: ") + path);
        *cfg = c;
    });
}

int cu2rec_config_write(const char *path, const cu2rec_config *cfg) {  // config.cu:15-22
    return guarded([&] {
        require(path && cfg, "null argument");
        std::ofstream out(path);
        if (!out.is_open()) fail(CU2REC_EIO, std::string("cannot write config ") + path);
        out << cfg->cur_iterations << " " << cfg->total_iterations << " " << cfg->n_factors << " "
            << cfg->learning_rate << " " << cfg->seed << " " << cfg->P_reg << " " << cfg->Q_reg << " "
            << cfg->user_bias_reg << " " << cfg->item_bias_reg << "\n";
    });
}

int cu2rec_config_print(const cu2rec_config *c) {  // config.cu:50-64, same lines (scripts grep them)
    return guarded([&] {
        require(c, "cfg is null");
        std::printf("Hyperparameters:\n");
        std::printf("total_iterations: %d\n", c->total_iterations);
        std::printf("n_factors: %d\n", c->n_factors);
        std::printf("learning_rate: %f\n", c->learning_rate);
        std::printf("P_reg: %f\n", c->P_reg);
        std::printf("Q_reg: %f\n", c->Q_reg);
        std::printf("user_bias_reg: %f\n", c->user_bias_reg);
        std::printf("item_bias_reg: %f\n", c->item_bias_reg);
        std::printf("is_train: %s\n", c->is_train ? "true" : "false");
        std::printf("n_threads: %d\n", c->n_threads);
        std::printf("check_error: %d\n", c->check_error);
        std::printf("patience: %f\n", c->patience);
        std::printf("learning_rate_decay: %f\n", c->learning_rate_decay);
    });
}

int cu2rec_ratings_read_csv(const char *path, cu2rec_ratings **out) {
    return guarded([&] {
        require(path && out, "null argument");
        *out = nullptr;
        auto holder = new cu2rec_ratings{read_ratings_csv(path)};
        *out = holder;
    });
}

// Binary cache of a parsed ratings file: header + the three COO columns, native endianness.
namespace {
struct CacheHeader {
    char magic[8];
    int32_t n, rows, cols;
    float global_bias;
};
const char kCacheMagic[8] = {'C', 'U', '2', 'R', 'C', 'O', 'O', '1'};
}  // namespace

int cu2rec_ratings_save_binary(const cu2rec_ratings *r, const char *path) {
    return guarded([&] {
        require(r && path, "null argument");
        FILE *fp = std::fopen(path, "wb");
        if (!fp) fail(CU2REC_EIO, std::string("cannot write ") + path);
        CacheHeader h;
        std::memcpy(h.magic, kCacheMagic, 8);
        h.n = static_cast<int32_t>(r->file.user.size());
        h.rows = r->file.rows;
        h.cols = r->file.cols;
        h.global_bias = r->file.global_bias;
        bool ok = std::fwrite(&h, sizeof(h), 1, fp) == 1;
        const size_t n = r->file.user.size();
        ok = ok && std::fwrite(r->file.user.data(), sizeof(int), n, fp) == n;
        ok = ok && std::fwrite(r->file.item.data(), sizeof(int), n, fp) == n;
        ok = ok && std::fwrite(r->file.rating.data(), sizeof(float), n, fp) == n;
        std::fclose(fp);
        if (!ok) fail(CU2REC_EIO, std::string("short write to ") + path);
    });
}

int cu2rec_ratings_load_binary(const char *path, cu2rec_ratings **out) {
    return guarded([&] {
        require(path && out, "null argument");
        *out = nullptr;
        FILE *fp = std::fopen(path, "rb");
        if (!fp) fail(CU2REC_EIO, std::string("cannot open ") + path);
        CacheHeader h;
        RatingsFile f;
        bool ok = std::fread(&h, sizeof(h), 1, fp) == 1 && std::memcmp(h.magic, kCacheMagic, 8) == 0 && h.n >= 0;
        if (ok) {
            const size_t n = static_cast<size_t>(h.n);
            f.user.resize(n);
            f.item.resize(n);
            f.rating.resize(n);
            ok = std::fread(f.user.data(), sizeof(int), n, fp) == n && std::fread(f.item.data(), sizeof(int), n, fp) == n &&
                 std::fread(f.rating.data(), sizeof(float), n, fp) == n;
            f.rows = h.rows;
            f.cols = h.cols;
            f.global_bias = h.global_bias;
            // the file is data, not code we wrote a moment ago: every id must lie inside the header's shape, or the CSR
            // build (indptr[rows + 1] in caller memory) and every kernel behind it would run past their arrays
            ok = ok && h.rows >= 0 && h.cols >= 0;
            for (size_t k = 0; ok && k < n; ++k)
                ok = f.user[k] >= 0 && f.user[k] < h.rows && f.item[k] >= 0 && f.item[k] < h.cols;
        }
        std::fclose(fp);
        if (!ok) fail(CU2REC_EIO, std::string("not a cu2rec ratings cache (bad magic, short file or ids outside its header's shape): ") + path);
        *out = new cu2rec_ratings{std::move(f)};
    });
}

int cu2rec_ratings_info(const cu2rec_ratings *r, int *n, int *rows, int *cols, float *global_bias) {
    return guarded([&] {
        require(r, "ratings is null");
        if (n) *n = static_cast<int>(r->file.user.size());
        if (rows) *rows = r->file.rows;
        if (cols) *cols = r->file.cols;
        if (global_bias) *global_bias = r->file.global_bias;
    });
}

int cu2rec_ratings_view(const cu2rec_ratings *r, const int **user, const int **item, const float **rating) {
    return guarded([&] {
        require(r, "ratings is null");
        if (user) *user = r->file.user.data();
        if (item) *item = r->file.item.data();
        if (rating) *rating = r->file.rating.data();
    });
}

void cu2rec_ratings_free(cu2rec_ratings *r) { delete r; }

int cu2rec_csr_build(const cu2rec_ratings *r, int rows, int *indptr, int *indices, float *data) {
    return guarded([&] {
        require(r && indptr, "null argument");
        require(r->file.user.empty() || (indices && data), "null argument");
        build_csr(r->file, rows, indptr, indices, data);
    });
}

int cu2rec_init_normal(float *out, size_t size, int n_factors, float mean, float stddev, int seed) {
    return guarded([&] {
        require(out || size == 0, "out is null");
        require(n_factors > 0, "n_factors must be positive");
        init_normal(out, size, n_factors, mean, stddev, seed);
    });
}

int cu2rec_write_csv(const char *path, const float *data, int rows, int cols) {
    return guarded([&] {
        require(path && data && rows >= 0 && cols > 0, "bad argument");
        write_csv(path, data, rows, cols);
    });
}

int cu2rec_write_component(const char *parent_dir, const char *base, const char *component, const float *data,
                           int rows, int cols, int factors) {  // writeToFile, util.cu:99-103
    return guarded([&] {
        require(parent_dir && base && component && data, "null argument");
        std::ostringstream name;
        name << parent_dir << "/" << base << "_f" << factors << "_" << component << ".csv";
        write_csv(name.str().c_str(), data, rows, cols);
    });
}

int cu2rec_read_array(const char *path, float **out, int *rows, int *cols) {  // read_array, util.cu:52-81
    return guarded([&] {
        require(path && out, "null argument");
        const std::string text = slurp(path);
        std::vector<float> values;
        int n_rows = 0, n_cells = 0;
        std::istringstream all(text);
        std::string line, cell;
        while (std::getline(all, line)) {
            std::istringstream ls(line);
            while (std::getline(ls, cell, ',')) {
                values.push_back(std::stof(cell));
                ++n_cells;
            }
            ++n_rows;
        }
        float *buf = static_cast<float *>(std::malloc(sizeof(float) * std::max<size_t>(values.size(), 1)));
        if (!buf) throw std::bad_alloc();
        std::copy(values.begin(), values.end(), buf);
        *out = buf;
        if (rows) *rows = n_rows;
        // the reference returns the running cell count in n_cols (util.cu:66,77); callers divide by rows
        if (cols) *cols = n_rows ? n_cells / n_rows : 0;
    });
}

void cu2rec_free(void *p) { std::free(p); }

uint32_t cu2rec_sampler_draw(uint64_t seed, uint64_t user, uint64_t iteration) {
    return sampler_draw(seed, user, iteration);
}

int cu2rec_sampler_index(uint64_t seed, uint64_t user, uint64_t iteration, int low, int high) {
    if (high <= low) return low;
    return sampler_index(seed, user, iteration, low, high);
}

int cu2rec_shard_plan(int rows, int nranks, int *user_begin) {
    return guarded([&] {
        require(rows >= 0 && nranks > 0 && user_begin, "bad argument");
        // equal user counts: one SGD iteration is one update per user, so users are the unit of work
        for (int k = 0; k <= nranks; ++k)
            user_begin[k] = static_cast<int>((static_cast<long long>(rows) * k) / nranks);
    });
}

int cu2rec_item_update_rates(const int *indptr, const int *indices, int n_rows, int n_cols, double *rate) {
    return guarded([&] {
        require(indptr && rate && n_rows >= 0 && n_cols >= 0, "bad argument");
        std::fill(rate, rate + n_cols, 0.0);
        for (int u = 0; u < n_rows; ++u) {
            const int lo = indptr[u], hi = indptr[u + 1];
            if (hi <= lo) continue;
            require(indices != nullptr, "indices is null");
            const double w = 1.0 / (hi - lo);
            for (int k = lo; k < hi; ++k) {
                require(indices[k] >= 0 && indices[k] < n_cols, "item id out of range");
                rate[indices[k]] += w;
            }
        }
    });
}

int cu2rec_csr_slice(const int *indptr, int rows, int u0, int u1, int *indptr_out, int *offset_out, int *nnz_out) {
    return guarded([&] {
        require(indptr && indptr_out, "null argument");
        require(0 <= u0 && u0 <= u1 && u1 <= rows, "user range out of bounds");
        const int base = indptr[u0];
        for (int u = u0; u <= u1; ++u) indptr_out[u - u0] = indptr[u] - base;
        if (offset_out) *offset_out = base;
        if (nnz_out) *nnz_out = indptr[u1] - base;
    });
}

}  // extern "C"
