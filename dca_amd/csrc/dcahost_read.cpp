// dcahost_read.cpp -- the count matrix a `dca <input.tsv> <outdir>` run starts from, read natively.
//
// Replaces, on the input side of the hot path, what dca/io.py:59 does for text files:
//     sc.read(filename, first_column_names=True)      (here restated as pandas.read_csv(sep, index_col=0))
// for the plain numeric matrices DCA is fed (one header line of gene / cell names, one name + numbers per line).
// pandas parses such a file on one thread at ~0.15 GB/s: the 2.8 GB text of a 68 579 x 20 000 count matrix takes
// longer to read than the whole training on the GPU.  Here the file is mapped, cut into line-aligned shares and parsed
// by a pool of threads straight into the caller's float32 matrix.
//
// Numbers are parsed by std::from_chars<double> (correctly rounded) and converted to float32 -- what
// `df.values.astype(np.float32)` does after pandas' float64 parse; integer counts, decimals and exponents all take that
// route; an empty field is NaN as in pandas.  Anything this reader does not implement (quoted fields, ragged lines,
// stray text) is reported as DCAHOST_EUNSUPPORTED and the Python caller falls back to pandas.
#include "dcahost.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <charconv>
#include <cmath>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

namespace {

struct TsvFile {
    const char* data = nullptr;
    size_t size = 0;
    int fd = -1;
    char sep = '\t';
    long nrows = 0, ncols = 0;          // data rows / numeric columns
    bool header_has_corner = false;     // header line has ncols + 1 fields (first = label of the name column)
    size_t body = 0;                    // offset of the first data line
    std::vector<size_t> share_begin;    // line-aligned shares of the body
    std::vector<long> share_row0;       // first row index of each share
};

inline const char* line_end(const char* p, const char* end) {
    const char* nl = static_cast<const char*>(memchr(p, '\n', end - p));
    return nl ? nl : end;
}

// number of sep-separated fields of [p, e)
inline long count_fields(const char* p, const char* e, char sep) {
    long n = 1;
    for (const char* q = p; q < e; ++q) n += (*q == sep);
    return n;
}

inline const char* strip_cr(const char* p, const char* e) { return (e > p && e[-1] == '\r') ? e - 1 : e; }

int n_threads(int requested, size_t bytes) {
    int n = requested > 0 ? requested : (int)std::thread::hardware_concurrency();
    if (n > 64) n = 64;
    const int by_size = (int)(bytes / (4u << 20)) + 1;          // >= 4 MB of text per thread
    if (n > by_size) n = by_size;
    return n < 1 ? 1 : n;
}

// one field -> float; false if it is not a plain number (pandas would yield a string column)
inline bool parse_field(const char* p, const char* e, float* out) {
    while (p < e && (*p == ' ')) ++p;
    while (e > p && (e[-1] == ' ')) --e;
    if (p == e) { *out = std::numeric_limits<float>::quiet_NaN(); return true; }
    if (*p == '+') {                                          // from_chars rejects a leading plus, pandas accepts ONE in front of a number
        ++p;
        if (p == e || *p == '+' || *p == '-') return false;
    }
    // the common case first: a short run of digits (counts)
    if (e - p <= 9) {
        unsigned v = 0;
        const char* q = p;
        for (; q < e && (unsigned)(*q - '0') <= 9u; ++q) v = v * 10u + (unsigned)(*q - '0');
        if (q == e) { *out = (float)v; return true; }
    }
    double d;
    const auto r = std::from_chars(p, e, d);
    if (r.ec == std::errc() && r.ptr == e) { *out = (float)d; return true; }
    if (r.ec == std::errc::result_out_of_range && r.ptr == e) {   // overflow / underflow: what strtod returns
        *out = (float)strtod(std::string(p, e).c_str(), nullptr);
        return true;
    }
    // the spellings pandas maps to NaN / inf
    const std::string s(p, e);
    static const char* const kNan[] = {"NA", "N/A", "NaN", "nan", "NULL", "null", "#N/A", "n/a", "-NaN", "-nan", "<NA>", "#NA", "None"};
    for (const char* n : kNan) if (s == n) { *out = std::numeric_limits<float>::quiet_NaN(); return true; }
    if (s == "inf" || s == "Inf" || s == "INF" || s == "infinity" || s == "Infinity") { *out = INFINITY; return true; }
    if (s == "-inf" || s == "-Inf" || s == "-INF" || s == "-infinity" || s == "-Infinity") { *out = -INFINITY; return true; }
    return false;
}

}  // namespace

extern "C" int dcahost_tsv_open(const char* path, char sep, int nthreads, void** handle, long* nrows, long* ncols,
                                long* rowname_bytes, long* colname_bytes) {
    if (!path || !handle || !nrows || !ncols || !rowname_bytes || !colname_bytes) return DCAHOST_EINVAL;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return DCAHOST_EIO;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) { close(fd); return st.st_size == 0 ? DCAHOST_EUNSUPPORTED : DCAHOST_EIO; }
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (m == MAP_FAILED) { close(fd); return DCAHOST_EIO; }
    madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
    auto* f = new TsvFile;
    f->data = static_cast<const char*>(m); f->size = (size_t)st.st_size; f->fd = fd; f->sep = sep;
    const char* const beg = f->data; const char* const end = beg + f->size;
    auto fail = [&](int rc) { munmap(m, f->size); close(fd); delete f; return rc; };
    if (memchr(beg, '"', f->size)) return fail(DCAHOST_EUNSUPPORTED);        // quoted fields: pandas' business

    // header line and the first data line fix the shape
    const char* h_end = line_end(beg, end);
    if (h_end == end) return fail(DCAHOST_EUNSUPPORTED);                         // no data line
    const long hf = count_fields(beg, strip_cr(beg, h_end), sep);
    const char* d0 = h_end + 1;
    while (d0 < end && (*d0 == '\n' || *d0 == '\r')) ++d0;                       // blank lines are skipped (as pandas does)
    if (d0 >= end) return fail(DCAHOST_EUNSUPPORTED);
    const char* d0_end = line_end(d0, end);
    const long df = count_fields(d0, strip_cr(d0, d0_end), sep);
    if (df < 2 || (hf != df && hf != df - 1)) return fail(DCAHOST_EUNSUPPORTED);
    f->ncols = df - 1;
    f->header_has_corner = hf == df;
    f->body = (size_t)(d0 - beg);

    // line-aligned shares of the body, rows per share counted in parallel
    const int T = n_threads(nthreads, f->size - f->body);
    f->share_begin.assign(T + 1, f->size);
    f->share_begin[0] = f->body;
    for (int t = 1; t < T; ++t) {
        size_t pos = f->body + (f->size - f->body) / T * t;
        if (pos <= f->share_begin[t - 1]) pos = f->share_begin[t - 1];
        const char* nl = static_cast<const char*>(memchr(beg + pos, '\n', f->size - pos));
        f->share_begin[t] = nl ? (size_t)(nl + 1 - beg) : f->size;
        if (f->share_begin[t] < f->share_begin[t - 1]) f->share_begin[t] = f->share_begin[t - 1];
    }
    std::vector<long> rows(T, 0), rbytes(T, 0);
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t)
            pool.emplace_back([&, t] {
                const char* p = beg + f->share_begin[t];
                const char* const e = beg + f->share_begin[t + 1];
                long n = 0, nb = 0;
                while (p < e) {
                    const char* le = line_end(p, e);
                    const char* ce = strip_cr(p, le);
                    if (ce > p) {                                                // non-blank line
                        ++n;
                        const char* s1 = static_cast<const char*>(memchr(p, sep, ce - p));
                        nb += (s1 ? s1 - p : ce - p) + 1;
                    }
                    p = le + 1;
                }
                rows[t] = n; rbytes[t] = nb;
            });
        for (auto& th : pool) th.join();
    }
    f->share_row0.assign(T + 1, 0);
    long rb = 0;
    for (int t = 0; t < T; ++t) { f->share_row0[t + 1] = f->share_row0[t] + rows[t]; rb += rbytes[t]; }
    f->nrows = f->share_row0[T];
    *handle = f; *nrows = f->nrows; *ncols = f->ncols;
    *rowname_bytes = rb;                                                         // names joined by '\n'
    *colname_bytes = (long)(strip_cr(beg, h_end) - beg) + 1;
    return DCAHOST_OK;
}

extern "C" int dcahost_tsv_read_f32(void* handle, float* out, long ld, char* rownames, long rowname_cap,
                                    char* colnames, long colname_cap) {
    auto* f = static_cast<TsvFile*>(handle);
    if (!f || !out || ld < f->ncols || !rownames || !colnames) return DCAHOST_EINVAL;
    const char* const beg = f->data;
    const char sep = f->sep;
    // column names: the header fields (without the corner label), joined by '\n'
    {
        const char* h_end = strip_cr(beg, line_end(beg, beg + f->size));
        const char* p = beg;
        if (f->header_has_corner) {
            const char* s1 = static_cast<const char*>(memchr(p, sep, h_end - p));
            p = s1 ? s1 + 1 : h_end;
        }
        long n = 0;
        for (const char* q = p; q < h_end; ++q) {
            if (n + 1 >= colname_cap) return DCAHOST_EINVAL;
            colnames[n++] = (*q == sep) ? '\n' : *q;
        }
        colnames[n] = '\0';
    }
    const int T = (int)f->share_begin.size() - 1;
    // row-name offsets per share: a second light pass would cost a read of the file; instead every share writes its names
    // into a private string and they are concatenated afterwards
    std::vector<std::string> names(T);
    std::atomic<int> bad{0};
    std::vector<std::thread> pool;
    for (int t = 0; t < T; ++t)
        pool.emplace_back([&, t] {
            const char* p = beg + f->share_begin[t];
            const char* const e = beg + f->share_begin[t + 1];
            long row = f->share_row0[t];
            std::string& nm = names[t];
            while (p < e && !bad.load(std::memory_order_relaxed)) {
                const char* le = line_end(p, e);
                const char* ce = strip_cr(p, le);
                if (ce > p) {
                    const char* s1 = static_cast<const char*>(memchr(p, sep, ce - p));
                    if (!s1) { bad = 1; break; }
                    nm.append(p, s1 - p); nm.push_back('\n');
                    float* dst = out + row * ld;
                    const char* q = s1 + 1;
                    long j = 0;
                    for (; j < f->ncols; ++j) {
                        const char* fe = (j + 1 < f->ncols) ? static_cast<const char*>(memchr(q, sep, ce - q)) : ce;
                        if (!fe) break;                                          // too few fields
                        if (j + 1 == f->ncols && memchr(q, sep, ce - q)) { j = -1; break; }   // too many fields
                        if (!parse_field(q, fe, dst + j)) { j = -1; break; }
                        q = fe + 1;
                    }
                    if (j != f->ncols) { bad = 1; break; }
                    ++row;
                }
                p = le + 1;
            }
        });
    for (auto& th : pool) th.join();
    if (bad) return DCAHOST_EUNSUPPORTED;
    long n = 0;
    for (int t = 0; t < T; ++t) {
        if (n + (long)names[t].size() + 1 > rowname_cap) return DCAHOST_EINVAL;
        memcpy(rownames + n, names[t].data(), names[t].size());
        n += (long)names[t].size();
    }
    rownames[n > 0 ? n - 1 : 0] = '\0';                                          // drop the last '\n'
    return DCAHOST_OK;
}

extern "C" void dcahost_tsv_close(void* handle) {
    auto* f = static_cast<TsvFile*>(handle);
    if (!f) return;
    munmap(const_cast<char*>(f->data), f->size);
    close(f->fd);
    delete f;
}
