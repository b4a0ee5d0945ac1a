#include "vtk_io.hpp"

#include "fast_deflate.hpp"

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cmath>
#include <limits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>

#include <omp.h>
#include <zlib.h>

namespace {

std::string lower(std::string s) {
    std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return static_cast<char>(std::tolower(c)); });
    return s;
}

// Cursor over the whole file: whitespace-separated tokens for the ASCII parts, raw big-endian
// blocks for BINARY data sections.
class cursor {
public:
    explicit cursor(std::string data) : buf_(std::move(data)) {}
    bool eof() {
        skip_ws();
        return pos_ >= buf_.size();
    }
    std::string line() {
        const size_t e = buf_.find('\n', pos_);
        std::string s = buf_.substr(pos_, e == std::string::npos ? std::string::npos : e - pos_);
        pos_ = (e == std::string::npos) ? buf_.size() : e + 1;
        if (!s.empty() && s.back() == '\r') s.pop_back();
        return s;
    }
    std::string token() {
        skip_ws();
        const size_t b = pos_;
        while (pos_ < buf_.size() && !std::isspace(static_cast<unsigned char>(buf_[pos_]))) ++pos_;
        if (b == pos_) throw std::runtime_error("vtk: unexpected end of file");
        return buf_.substr(b, pos_ - b);
    }
    std::string peek() {
        const size_t save = pos_;
        std::string t = eof() ? std::string() : token();
        pos_ = save;
        return t;
    }
    void rest_of_line() {
        while (pos_ < buf_.size() && buf_[pos_] != '\n') ++pos_;
        if (pos_ < buf_.size()) ++pos_;
    }
    size_t remaining() const { return buf_.size() - pos_; }
    const unsigned char* raw(size_t n) {
        if (n > buf_.size() - pos_) throw std::runtime_error("vtk: binary section runs past the end of the file");
        const unsigned char* p = reinterpret_cast<const unsigned char*>(buf_.data()) + pos_;
        pos_ += n;
        return p;
    }

private:
    void skip_ws() {
        while (pos_ < buf_.size() && std::isspace(static_cast<unsigned char>(buf_[pos_]))) ++pos_;
    }
    std::string buf_;
    size_t pos_ = 0;
};

size_t type_size(const std::string& t) {
    if (t == "bit") throw std::runtime_error("vtk: bit arrays are not supported");
    if (t == "char" || t == "unsigned_char") return 1;
    if (t == "short" || t == "unsigned_short") return 2;
    if (t == "int" || t == "unsigned_int" || t == "float" || t == "vtktypeint32" || t == "vtktypeuint32") return 4;
    if (t == "long" || t == "unsigned_long" || t == "double" || t == "vtktypeint64" || t == "vtktypeuint64" ||
        t == "vtkidtype")
        return 8;
    throw std::runtime_error("vtk: unknown data type '" + t + "'");
}

template <class T>
T from_big_endian(const unsigned char* p) {
    unsigned char tmp[sizeof(T)];
    for (size_t k = 0; k < sizeof(T); ++k) tmp[k] = p[sizeof(T) - 1 - k];
    T v;
    std::memcpy(&v, tmp, sizeof(T));
    return v;
}

double binary_value(const std::string& t, const unsigned char* p) {
    if (t == "char") return static_cast<double>(static_cast<signed char>(*p));
    if (t == "unsigned_char") return static_cast<double>(*p);
    if (t == "short") return from_big_endian<int16_t>(p);
    if (t == "unsigned_short") return from_big_endian<uint16_t>(p);
    if (t == "int" || t == "vtktypeint32") return from_big_endian<int32_t>(p);
    if (t == "unsigned_int" || t == "vtktypeuint32") return from_big_endian<uint32_t>(p);
    if (t == "float") return from_big_endian<float>(p);
    if (t == "double") return from_big_endian<double>(p);
    if (t == "unsigned_long" || t == "vtktypeuint64") return static_cast<double>(from_big_endian<uint64_t>(p));
    return static_cast<double>(from_big_endian<int64_t>(p));  // long, vtktypeint64, vtkidtype
}

// n values of `type` into doubles (ids up to 2^53 survive the trip)
std::vector<double> read_array(cursor& c, bool binary, const std::string& type_in, size_t n) {
    const std::string type = lower(type_in);
    // A count is a claim of the file: before anything of that size is allocated it must fit in what is left of the
    // file (a binary value takes its size, an ASCII one a character and a separator at least) — a header that says
    // "POINTS 999999999" on a 60 KB file is reported, not obeyed (24 GB of zeros first, found by the sanitizer run).
    if (binary) c.rest_of_line();
    const size_t sz = binary ? type_size(type) : 2;
    if (n > (c.remaining() + 1) / sz)
        throw std::runtime_error(binary ? "vtk: binary section runs past the end of the file" : "vtk: unexpected end of file (an array is shorter than its header says)");
    std::vector<double> out(n);
    if (binary) {
        const unsigned char* p = c.raw(n * sz);
        for (size_t i = 0; i < n; ++i) out[i] = binary_value(type, p + i * sz);
    } else {
        for (size_t i = 0; i < n; ++i) {
            const std::string t = c.token();
            char* end = nullptr;
            out[i] = std::strtod(t.c_str(), &end);
            if (end == t.c_str()) throw std::runtime_error("vtk: expected a number, found '" + t + "'");
        }
    }
    return out;
}

size_t to_count(const std::string& t) {
    char* end = nullptr;
    const long long v = std::strtoll(t.c_str(), &end, 10);
    if (end == t.c_str() || *end != '\0' || v < 0 || v > (1ll << 31)) throw std::runtime_error("vtk: expected a count, found '" + t + "'");
    return static_cast<size_t>(v);
}

// A point id or an offset as it came out of read_array (a double): a whole number below `limit`, or the file is wrong
// (a float -> integer conversion out of range is undefined behaviour, not an error message).
size_t to_index(double v, size_t limit, const char* what) {
    if (!(v >= 0.0) || !(v < static_cast<double>(limit)) || v != std::floor(v)) throw std::runtime_error(std::string("vtk: ") + what);
    return static_cast<size_t>(v);
}

void skip_metadata(cursor& c) {  // METADATA block ends at an empty line
    c.rest_of_line();
    while (!c.eof()) {
        const std::string l = c.line();
        if (l.find_first_not_of(" \t") == std::string::npos) break;
    }
}

}  // namespace

vtk_grid read_legacy_vtk(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open '" + path + "'");
    std::stringstream ss;
    ss << f.rdbuf();
    cursor c(ss.str());

    const std::string magic = c.line();
    if (magic.rfind("# vtk DataFile Version", 0) != 0) throw std::runtime_error("vtk: not a legacy VTK file: " + path);
    c.line();  // title
    const std::string mode = lower(c.token());
    if (mode != "ascii" && mode != "binary") throw std::runtime_error("vtk: expected ASCII or BINARY");
    const bool binary = (mode == "binary");

    vtk_grid g;
    size_t n_cells = 0;
    enum class section { none, cell, point } where = section::none;
    size_t section_count = 0;

    while (!c.eof()) {
        const std::string key = lower(c.token());
        if (key == "dataset") {
            const std::string kind = lower(c.token());
            if (kind != "unstructured_grid") throw std::runtime_error("vtk: DATASET " + kind + " is not an unstructured grid");
        } else if (key == "points") {
            const size_t n = to_count(c.token());
            const std::string type = c.token();
            g.points = read_array(c, binary, type, 3 * n);
        } else if (key == "cells") {
            const size_t a = to_count(c.token());
            const size_t b = to_count(c.token());
            std::vector<double> offsets, conn;
            if (lower(c.peek()) == "offsets") {  // 5.1 layout: CELLS n+1 m / OFFSETS type / CONNECTIVITY type
                c.token();
                offsets = read_array(c, binary, c.token(), a);
                if (lower(c.token()) != "connectivity") throw std::runtime_error("vtk: CONNECTIVITY expected");
                conn = read_array(c, binary, c.token(), b);
                n_cells = a ? a - 1 : 0;
                g.tets.resize(4 * n_cells);
                for (size_t k = 0; k < n_cells; ++k) {
                    const size_t lo = to_index(offsets[k], conn.size() + 1, "a cell offset is out of range");
                    const size_t hi = to_index(offsets[k + 1], conn.size() + 1, "a cell offset is out of range");
                    if (hi < lo || hi - lo < 4) throw std::runtime_error("vtk: a cell has fewer than four points");
                    for (int v = 0; v < 4; ++v)
                        g.tets[4 * k + v] = static_cast<int32_t>(to_index(conn[lo + v], size_t{1} << 31, "cell references a point id out of range"));
                }
            } else {  // classic: CELLS n size, then n records "k id0 ... id(k-1)"
                n_cells = a;
                const std::vector<double> raw = read_array(c, binary, "int", b);
                if (n_cells > raw.size() / 5) throw std::runtime_error("vtk: CELLS section is truncated");  // 5 entries per tetrahedron at least
                g.tets.resize(4 * n_cells);
                size_t p = 0;
                for (size_t k = 0; k < n_cells; ++k) {
                    if (p >= raw.size()) throw std::runtime_error("vtk: CELLS section is truncated");
                    const size_t m = to_index(raw[p], raw.size(), "a cell's point count is out of range");
                    if (m < 4 || p + 1 + m > raw.size()) throw std::runtime_error("vtk: a cell has fewer than four points");
                    for (int v = 0; v < 4; ++v)
                        g.tets[4 * k + v] = static_cast<int32_t>(to_index(raw[p + 1 + v], size_t{1} << 31, "cell references a point id out of range"));
                    p += 1 + m;
                }
            }
        } else if (key == "cell_types") {
            const size_t n = to_count(c.token());
            read_array(c, binary, "int", n);
        } else if (key == "cell_data") {
            where = section::cell;
            section_count = to_count(c.token());
        } else if (key == "point_data") {
            where = section::point;
            section_count = to_count(c.token());
        } else if (key == "scalars") {
            const std::string name = c.token();
            const std::string type = c.token();
            size_t ncomp = 1;
            std::string nxt = c.peek();
            if (!nxt.empty() && std::isdigit(static_cast<unsigned char>(nxt[0]))) ncomp = to_count(c.token());
            if (lower(c.peek()) == "lookup_table") {
                c.token();
                c.token();
            }
            const std::vector<double> v = read_array(c, binary, type, ncomp * section_count);
            if (where == section::cell) {
                std::vector<double>& dst = g.cell_scalars[name];
                dst.resize(section_count);
                for (size_t k = 0; k < section_count; ++k) dst[k] = v[k * ncomp];
            }
        } else if (key == "lookup_table") {
            c.token();
            const size_t n = to_count(c.token());
            if (binary) {
                c.rest_of_line();
                c.raw(4 * n);
            } else {
                if (4 * n > c.remaining()) throw std::runtime_error("vtk: unexpected end of file");
                for (size_t k = 0; k < 4 * n; ++k) c.token();
            }
        } else if (key == "vectors" || key == "normals") {
            c.token();
            read_array(c, binary, c.token(), 3 * section_count);
        } else if (key == "texture_coordinates") {
            c.token();
            const size_t dim = to_count(c.token());
            read_array(c, binary, c.token(), dim * section_count);
        } else if (key == "tensors") {
            c.token();
            read_array(c, binary, c.token(), 9 * section_count);
        } else if (key == "field") {
            c.token();
            const size_t n_arrays = to_count(c.token());
            for (size_t a = 0; a < n_arrays; ++a) {
                std::string name = c.token();
                while (lower(name) == "metadata") {  // 5.x writes METADATA after arrays
                    skip_metadata(c);
                    name = c.token();
                }
                const size_t ncomp = to_count(c.token());
                const size_t ntup = to_count(c.token());
                const std::string type = c.token();
                const std::vector<double> v = read_array(c, binary, type, ncomp * ntup);
                if (where == section::cell && ntup == section_count && ncomp >= 1) {
                    std::vector<double>& dst = g.cell_scalars[name];
                    dst.resize(ntup);
                    for (size_t k = 0; k < ntup; ++k) dst[k] = v[k * ncomp];
                }
            }
        } else if (key == "metadata") {
            skip_metadata(c);
        } else {
            throw std::runtime_error("vtk: unsupported keyword '" + key + "'");
        }
    }
    if (g.points.empty() || g.tets.empty()) throw std::runtime_error("vtk: no POINTS or no CELLS in " + path);
    const int64_t np = g.n_points();
    for (int32_t id : g.tets)
        if (id < 0 || id >= np) throw std::runtime_error("vtk: cell references a point id out of range");
    for (auto& kv : g.cell_scalars)
        if (kv.second.size() != n_cells) throw std::runtime_error("vtk: scalar array '" + kv.first + "' has the wrong length");
    return g;
}

namespace {

// The writer's loops are short: more than a few dozen threads only add fork/join cost (and a box whose CPU
// share is smaller than its core count runs the surplus threads one after the other).
int writer_threads() {
    static const int n = [] {
        long cpus = std::min(omp_get_max_threads(), 32);
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // a container's CPU quota: "<quota|max> <period>"
            char quota[32] = {0};
            long period = 0;
            if (std::fscanf(f, "%31s %ld", quota, &period) == 2 && period > 0 && std::strcmp(quota, "max") != 0)
                cpus = std::min(cpus, std::max(1L, std::atol(quota) / period));
            std::fclose(f);
        }
        return static_cast<int>(std::max(1L, cpus));
    }();
    return n;
}

// base64 of data[0, n) into out (4 * ceil(n / 3) characters); chunks of whole 3-byte groups in parallel
void base64_into(const unsigned char* data, size_t n, char* out) {
    static const char tab[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    const int64_t groups = static_cast<int64_t>(n / 3);
#pragma omp parallel for schedule(static) num_threads(writer_threads())
    for (int64_t g = 0; g < groups; ++g) {
        const unsigned char* d = data + 3 * g;
        const unsigned v = (static_cast<unsigned>(d[0]) << 16) | (static_cast<unsigned>(d[1]) << 8) | d[2];
        char* o = out + 4 * g;
        o[0] = tab[(v >> 18) & 63];
        o[1] = tab[(v >> 12) & 63];
        o[2] = tab[(v >> 6) & 63];
        o[3] = tab[v & 63];
    }
    const size_t i = static_cast<size_t>(groups) * 3;
    char* o = out + 4 * static_cast<size_t>(groups);
    if (i + 1 == n) {
        const unsigned v = static_cast<unsigned>(data[i]) << 16;
        o[0] = tab[(v >> 18) & 63];
        o[1] = tab[(v >> 12) & 63];
        o[2] = '=';
        o[3] = '=';
    } else if (i + 2 == n) {
        const unsigned v = (static_cast<unsigned>(data[i]) << 16) | (static_cast<unsigned>(data[i + 1]) << 8);
        o[0] = tab[(v >> 18) & 63];
        o[1] = tab[(v >> 12) & 63];
        o[2] = tab[(v >> 6) & 63];
        o[3] = '=';
    }
}

std::string base64(const unsigned char* data, size_t n) {
    std::string out((n + 2) / 3 * 4, '\0');
    base64_into(data, n, &out[0]);
    return out;
}

}  // namespace

void write_vti(const std::string& path, const float* image, int res_x, int res_y, bool compressed) {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot write '" + path + "'");
    const uint64_t n_values = static_cast<uint64_t>(res_x) * static_cast<uint64_t>(res_y) * 2u;
    const uint64_t n_bytes = n_values * sizeof(double);
    f << "<?xml version=\"1.0\"?>\n"
      << "<VTKFile type=\"ImageData\" version=\"1.0\" byte_order=\"LittleEndian\" header_type=\"UInt64\""
      << (compressed ? " compressor=\"vtkZLibDataCompressor\"" : "") << ">\n"
      << "  <ImageData WholeExtent=\"0 " << res_x - 1 << " 0 " << res_y - 1 << " 0 0\" Origin=\"0 0 0\" Spacing=\"1 1 1\">\n"
      << "    <Piece Extent=\"0 " << res_x - 1 << " 0 " << res_y - 1 << " 0 0\">\n"
      << "      <PointData Scalars=\"ImageScalars\">\n"
      << "        <DataArray type=\"Float64\" Name=\"ImageScalars\" NumberOfComponents=\"2\" format=\"appended\" offset=\"0\"/>\n"
      << "      </PointData>\n"
      << "      <CellData/>\n"
      << "    </Piece>\n"
      << "  </ImageData>\n"
      << "  <AppendedData encoding=\"" << (compressed ? "base64" : "raw") << "\">\n   _";
    // fp32 results widened to the VTK_DOUBLE the reference allocates (object2d.cpp:13,17-21)
    if (!compressed) {
        f.write(reinterpret_cast<const char*>(&n_bytes), sizeof n_bytes);
        std::vector<double> row(static_cast<size_t>(res_x) * 2);
        for (int y = 0; y < res_y; ++y) {
            const float* src = image + static_cast<size_t>(y) * res_x * 2;
            for (size_t k = 0; k < row.size(); ++k) row[k] = static_cast<double>(src[k]);
            f.write(reinterpret_cast<const char*>(row.data()), static_cast<std::streamsize>(row.size() * sizeof(double)));
        }
    } else {
        // vtkZLibDataCompressor layout: header {#blocks, block size, size of the last block (0 = full),
        // compressed size of each block} as header_type words, base64-encoded on its own, followed by
        // the concatenated deflated blocks, base64-encoded as one stream.
        constexpr uint64_t kBlock = 32768;
        const uint64_t n_blocks = (n_bytes + kBlock - 1) / kBlock;
        std::vector<uint64_t> header(3 + n_blocks);
        header[0] = n_blocks;
        header[1] = kBlock;
        header[2] = n_bytes % kBlock;
        // every block deflates into its own stretch of one scratch buffer (no allocation per block), then
        // the stretches are packed together at their prefix offsets; both loops run on all threads.
        // Level 1: a frame of a sweep is written in tens of milliseconds instead of hundreds, the file is
        // 10-15 % larger (readers decode any level)
        const uLong bound = compressBound(static_cast<uLong>(kBlock));
        // (kept between calls: faulting 70 MB of fresh pages in from many threads costs more than the deflate)
        static thread_local std::vector<unsigned char> scratch_buf, body_buf;
        static thread_local std::string text_buf;
        if (scratch_buf.size() < static_cast<size_t>(n_blocks) * bound) scratch_buf.resize(static_cast<size_t>(n_blocks) * bound);
        unsigned char* const scratch = scratch_buf.data();
        std::vector<uint64_t> packed_size(n_blocks, 0);
        const int64_t nb = static_cast<int64_t>(n_blocks);
        const bool tm_ = std::getenv("C5_VTI_TIMING") != nullptr;
        auto now_ = [] { return std::chrono::steady_clock::now(); };
        auto t_a = now_();
        // A full block of zeros (rays that meet nothing: most of a frame) deflates to the same bytes every time:
        // deflated once, copied thereafter — deflate at level 1 runs at ~130 MB/s per core and is all a frame's
        // write costs (measured: 570 ms on one core for a 2400x1800 frame, 135 ms on eight).
        static const std::vector<unsigned char> zero_block = [&] {
            std::vector<unsigned char> zeros(kBlock, 0), out(bound);
            uLongf cap = bound;
            if (compress2(out.data(), &cap, zeros.data(), static_cast<uLong>(kBlock), Z_BEST_SPEED) != Z_OK) cap = 0;
            out.resize(cap);
            return out;
        }();
#pragma omp parallel for schedule(dynamic, 16) num_threads(writer_threads())
        for (int64_t b = 0; b < nb; ++b) {
            const uint64_t first = static_cast<uint64_t>(b) * kBlock / sizeof(double);
            const uint64_t count = std::min<uint64_t>(kBlock / sizeof(double), n_values - first);
            uint32_t any = 0;  // +0.0f only (a -0.0f or a NaN has bits set)
            for (uint64_t k = 0; k < count; ++k) {
                uint32_t bits;
                std::memcpy(&bits, image + first + k, sizeof bits);
                any |= bits;
            }
            if (any == 0 && count == kBlock / sizeof(double) && !zero_block.empty()) {
                std::memcpy(scratch + static_cast<size_t>(b) * bound, zero_block.data(), zero_block.size());
                packed_size[static_cast<size_t>(b)] = zero_block.size();
                continue;
            }
            // the known parse of widened floats, straight from the floats (fast_deflate.cpp) - zlib's own level 1 for
            // whatever it declines (a last block too short to carry its code tables)
            uLongf cap = static_cast<uLongf>(c5::deflate_floats_as_doubles(image + first, static_cast<size_t>(count), scratch + static_cast<size_t>(b) * bound, bound));
            if (cap == 0) {
                double vals[kBlock / sizeof(double)];
                for (uint64_t k = 0; k < count; ++k) vals[k] = static_cast<double>(image[first + k]);
                cap = bound;
                if (compress2(scratch + static_cast<size_t>(b) * bound, &cap, reinterpret_cast<const Bytef*>(vals),
                              static_cast<uLong>(count * sizeof(double)), Z_BEST_SPEED) != Z_OK)
                    cap = 0;
            }
            packed_size[static_cast<size_t>(b)] = cap;
        }
        auto t_b = now_();
        std::vector<uint64_t> offset(n_blocks + 1, 0);
        for (uint64_t b = 0; b < n_blocks; ++b) {
            if (packed_size[b] == 0) throw std::runtime_error("zlib failed while writing '" + path + "'");
            header[3 + b] = packed_size[b];
            offset[b + 1] = offset[b] + packed_size[b];
        }
        if (body_buf.size() < offset[n_blocks] + 1) body_buf.resize(offset[n_blocks] + 1);
        unsigned char* const body = body_buf.data();
#pragma omp parallel for schedule(static) num_threads(writer_threads())
        for (int64_t b = 0; b < nb; ++b)
            std::memcpy(body + offset[static_cast<size_t>(b)], scratch + static_cast<size_t>(b) * bound,
                        packed_size[static_cast<size_t>(b)]);
        auto t_c = now_();
        f << base64(reinterpret_cast<const unsigned char*>(header.data()), header.size() * sizeof(uint64_t));
        const size_t text_len = (offset[n_blocks] + 2) / 3 * 4;
        if (text_buf.size() < text_len) text_buf.resize(text_len);
        base64_into(body, offset[n_blocks], &text_buf[0]);
        auto t_d = now_();
        f.write(text_buf.data(), static_cast<std::streamsize>(text_len));
        auto t_e = now_();
        if (tm_) {
            auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
            std::fprintf(stderr, "write_vti: deflate %.2f pack %.2f base64 %.2f write %.2f ms (threads %d)\n", ms(t_a, t_b), ms(t_b, t_c), ms(t_c, t_d), ms(t_d, t_e), writer_threads());
        }
    }
    f << "\n  </AppendedData>\n</VTKFile>\n";
    if (!f) throw std::runtime_error("error while writing '" + path + "'");
}

// ------------------------------------------------------------------------------------------
// Colour-mapped PNG of one channel: what utility/screen.py asks ParaView for, frame by frame
// (ColorBy(('POINTS', 'ImageScalars', 'Y')) + SaveScreenshot, screen.py:11-14), without ParaView.
// ------------------------------------------------------------------------------------------
namespace {

// "Cool to Warm": ParaView's default colour map, three control points joined in Moreland's diverging
// (Msh) colour space.  Restated from the published construction (K. Moreland, "Diverging Color Maps for
// Scientific Visualization", 2009, section 5); there is no ParaView here to compare rendered pixels with.
struct msh_t { double M, s, h; };
constexpr double PI_3 = 3.14159265358979323846 / 3.0;

double srgb_to_linear(double c) { return c > 0.04045 ? std::pow((c + 0.055) / 1.055, 2.4) : c / 12.92; }
double linear_to_srgb(double c) { return c > 0.0031308 ? 1.055 * std::pow(c, 1.0 / 2.4) - 0.055 : 12.92 * c; }
constexpr double kWhite[3] = {0.9505, 1.0, 1.0890};  // D65
double lab_f(double t) { return t > 0.008856 ? std::cbrt(t) : 7.787 * t + 16.0 / 116.0; }
double lab_f_inv(double t) { return t > 0.20689 ? t * t * t : (t - 16.0 / 116.0) / 7.787; }

msh_t rgb_to_msh(const double rgb[3]) {
    const double r = srgb_to_linear(rgb[0]), g = srgb_to_linear(rgb[1]), b = srgb_to_linear(rgb[2]);
    const double X = 0.4124 * r + 0.3576 * g + 0.1805 * b;
    const double Y = 0.2126 * r + 0.7152 * g + 0.0722 * b;
    const double Z = 0.0193 * r + 0.1192 * g + 0.9505 * b;
    const double fx = lab_f(X / kWhite[0]), fy = lab_f(Y / kWhite[1]), fz = lab_f(Z / kWhite[2]);
    const double L = 116.0 * fy - 16.0, a = 500.0 * (fx - fy), bb = 200.0 * (fy - fz);
    msh_t m;
    m.M = std::sqrt(L * L + a * a + bb * bb);
    m.s = m.M > 0.001 ? std::acos(L / m.M) : 0.0;
    m.h = m.s > 0.001 ? std::atan2(bb, a) : 0.0;
    return m;
}
void msh_to_rgb(const msh_t& m, double rgb[3]) {
    const double L = m.M * std::cos(m.s), a = m.M * std::sin(m.s) * std::cos(m.h), b = m.M * std::sin(m.s) * std::sin(m.h);
    const double fy = (L + 16.0) / 116.0, fx = a / 500.0 + fy, fz = fy - b / 200.0;
    const double X = kWhite[0] * lab_f_inv(fx), Y = kWhite[1] * lab_f_inv(fy), Z = kWhite[2] * lab_f_inv(fz);
    const double r = 3.2406 * X - 1.5372 * Y - 0.4986 * Z;
    const double g = -0.9689 * X + 1.8758 * Y + 0.0415 * Z;
    const double bl = 0.0557 * X - 0.2040 * Y + 1.0570 * Z;
    const double lin[3] = {r, g, bl};
    for (int k = 0; k < 3; ++k) rgb[k] = std::min(1.0, std::max(0.0, linear_to_srgb(std::max(0.0, lin[k]))));
}
// hue an unsaturated end takes so that the path towards the saturated end does not swing through other hues
double adjust_hue(const msh_t& saturated, double M_unsaturated) {
    if (saturated.M >= M_unsaturated - 0.1) return saturated.h;
    const double spin = saturated.s * std::sqrt(M_unsaturated * M_unsaturated - saturated.M * saturated.M) /
                        (saturated.M * std::sin(saturated.s));
    return saturated.h > -PI_3 ? saturated.h + spin : saturated.h - spin;
}
void diverging_between(const double rgb1[3], const double rgb2[3], double t, double out[3]) {
    msh_t a = rgb_to_msh(rgb1), b = rgb_to_msh(rgb2);
    double dh = std::fabs(a.h - b.h);
    if (dh > 3.14159265358979323846) dh = 2 * 3.14159265358979323846 - dh;
    if (a.s > 0.05 && b.s > 0.05 && dh > PI_3) {  // two saturated ends: pass through white
        const double mid = std::max(std::max(a.M, b.M), 88.0);
        if (t < 0.5) { b = {mid, 0.0, 0.0}; t *= 2.0; }
        else { a = {mid, 0.0, 0.0}; t = 2.0 * t - 1.0; }
    }
    if (a.s < 0.05 && b.s > 0.05) a.h = adjust_hue(b, a.M);
    else if (b.s < 0.05 && a.s > 0.05) b.h = adjust_hue(a, b.M);
    const msh_t m{(1 - t) * a.M + t * b.M, (1 - t) * a.s + t * b.s, (1 - t) * a.h + t * b.h};
    msh_to_rgb(m, out);
}

struct colour_table {
    unsigned char rgb[256][3];
    colour_table() {
        const double cool[3] = {0.23137254902, 0.298039215686, 0.752941176471};
        const double mid[3] = {0.865, 0.865, 0.865};
        const double warm[3] = {0.705882352941, 0.0156862745098, 0.149019607843};
        for (int k = 0; k < 256; ++k) {
            const double t = k / 255.0;
            double c[3];
            if (t < 0.5) diverging_between(cool, mid, 2.0 * t, c);
            else diverging_between(mid, warm, 2.0 * t - 1.0, c);
            for (int j = 0; j < 3; ++j) rgb[k][j] = static_cast<unsigned char>(std::lround(c[j] * 255.0));
        }
    }
};

void put_be32(unsigned char* p, uint32_t v) {
    p[0] = static_cast<unsigned char>(v >> 24);
    p[1] = static_cast<unsigned char>(v >> 16);
    p[2] = static_cast<unsigned char>(v >> 8);
    p[3] = static_cast<unsigned char>(v);
}
void png_chunk(std::ofstream& f, const char type[4], const unsigned char* data, size_t n) {
    unsigned char head[8];
    put_be32(head, static_cast<uint32_t>(n));
    std::memcpy(head + 4, type, 4);
    f.write(reinterpret_cast<const char*>(head), 8);
    if (n) f.write(reinterpret_cast<const char*>(data), static_cast<std::streamsize>(n));
    uLong crc = crc32(0L, reinterpret_cast<const Bytef*>(type), 4);
    if (n) crc = crc32(crc, data, static_cast<uInt>(n));
    unsigned char tail[4];
    put_be32(tail, static_cast<uint32_t>(crc));
    f.write(reinterpret_cast<const char*>(tail), 4);
}

}  // namespace

void colour_range(const float* image, int res_x, int res_y, int channel, double* lo, double* hi) {
    const int64_t n = static_cast<int64_t>(res_x) * res_y;
    float mn = std::numeric_limits<float>::infinity(), mx = -std::numeric_limits<float>::infinity();
#pragma omp parallel for schedule(static) reduction(min : mn) reduction(max : mx) num_threads(writer_threads())
    for (int64_t k = 0; k < n; ++k) {
        const float v = image[2 * k + channel];
        if (std::isfinite(v)) {
            mn = std::min(mn, v);
            mx = std::max(mx, v);
        }
    }
    if (!(mn <= mx)) mn = mx = 0.f;  // no finite value at all
    *lo = mn;
    *hi = mx;
}

void write_png(const std::string& path, const float* image, int res_x, int res_y, int channel, double lo, double hi) {
    static const colour_table table;
    constexpr unsigned char kNan[3] = {255, 255, 0};  // ParaView's default NaN colour
    if (res_x <= 0 || res_y <= 0 || channel < 0 || channel > 1) throw std::runtime_error("write_png: bad image shape");
    const double scale = hi > lo ? 255.0 / (hi - lo) : 0.0;
    // scanlines top to bottom = image rows res_y - 1 ... 0 (y points up in the reference's ParaView view);
    // every scanline: filter byte 1 (Sub: each byte minus the one three to its left), then RGB
    const size_t line = 1 + static_cast<size_t>(res_x) * 3;
    static thread_local std::vector<unsigned char> raw_buf, packed_buf;
    if (raw_buf.size() < line * static_cast<size_t>(res_y)) raw_buf.resize(line * static_cast<size_t>(res_y));
    unsigned char* const raw = raw_buf.data();
#pragma omp parallel for schedule(static) num_threads(writer_threads())
    for (int s = 0; s < res_y; ++s) {
        const float* src = image + static_cast<size_t>(res_y - 1 - s) * res_x * 2 + channel;
        unsigned char* dst = raw + static_cast<size_t>(s) * line;
        *dst++ = 1;
        unsigned char left[3] = {0, 0, 0};
        for (int x = 0; x < res_x; ++x) {
            const float v = src[2 * static_cast<size_t>(x)];
            const unsigned char* c = kNan;
            if (v == v) {  // (+-inf clamp to the ends of the map)
                const double t = (static_cast<double>(v) - lo) * scale;
                c = table.rgb[static_cast<int>(std::lround(std::min(255.0, std::max(0.0, t))))];
            }
            for (int j = 0; j < 3; ++j) {
                dst[j] = static_cast<unsigned char>(c[j] - left[j]);
                left[j] = c[j];
            }
            dst += 3;
        }
    }
    // one zlib stream out of independently deflated bands (each ends on a byte boundary with a full flush,
    // the last with the final block), so that the bands are compressed side by side
    const int bands = std::max(1, std::min(res_y / 16, 4 * writer_threads()));
    std::vector<size_t> band_first(bands + 1);
    for (int b = 0; b <= bands; ++b) band_first[b] = static_cast<size_t>(static_cast<int64_t>(res_y) * b / bands);
    std::vector<uLong> bound(bands), used(bands, 0), adler(bands);
    std::vector<size_t> at(bands + 1, 0);
    for (int b = 0; b < bands; ++b) {
        bound[b] = compressBound(static_cast<uLong>((band_first[b + 1] - band_first[b]) * line)) + 16;
        at[b + 1] = at[b] + bound[b];
    }
    if (packed_buf.size() < at[bands]) packed_buf.resize(at[bands]);
    unsigned char* const packed = packed_buf.data();
    bool failed = false;
#pragma omp parallel for schedule(dynamic, 1) num_threads(writer_threads())
    for (int b = 0; b < bands; ++b) {
        const unsigned char* in = raw + band_first[b] * line;
        const size_t n_in = (band_first[b + 1] - band_first[b]) * line;
        z_stream z;
        std::memset(&z, 0, sizeof z);
        if (deflateInit2(&z, Z_BEST_SPEED, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) {
#pragma omp atomic write
            failed = true;
            continue;
        }
        z.next_in = const_cast<Bytef*>(in);
        z.avail_in = static_cast<uInt>(n_in);
        z.next_out = packed + at[b];
        z.avail_out = static_cast<uInt>(bound[b]);
        const int rc = deflate(&z, b + 1 == bands ? Z_FINISH : Z_FULL_FLUSH);
        if ((b + 1 == bands && rc != Z_STREAM_END) || (b + 1 != bands && (rc != Z_OK || z.avail_in != 0))) {
#pragma omp atomic write
            failed = true;
        }
        used[b] = z.total_out;
        deflateEnd(&z);
        adler[b] = adler32(adler32(0L, Z_NULL, 0), in, static_cast<uInt>(n_in));
    }
    if (failed) throw std::runtime_error("zlib failed while writing '" + path + "'");
    uLong check = adler[0];
    for (int b = 1; b < bands; ++b) check = adler32_combine(check, adler[b], static_cast<z_off_t>((band_first[b + 1] - band_first[b]) * line));
    size_t total = 2 + 4;
    for (int b = 0; b < bands; ++b) total += used[b];
    std::vector<unsigned char> idat(total);
    idat[0] = 0x78;
    idat[1] = 0x01;
    size_t w = 2;
    for (int b = 0; b < bands; ++b) {
        std::memcpy(idat.data() + w, packed + at[b], used[b]);
        w += used[b];
    }
    put_be32(idat.data() + w, static_cast<uint32_t>(check));

    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot write '" + path + "'");
    static const unsigned char magic[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    f.write(reinterpret_cast<const char*>(magic), 8);
    unsigned char ihdr[13];
    put_be32(ihdr, static_cast<uint32_t>(res_x));
    put_be32(ihdr + 4, static_cast<uint32_t>(res_y));
    ihdr[8] = 8;   // bits per sample
    ihdr[9] = 2;   // truecolour
    ihdr[10] = ihdr[11] = ihdr[12] = 0;
    png_chunk(f, "IHDR", ihdr, sizeof ihdr);
    png_chunk(f, "IDAT", idat.data(), idat.size());
    png_chunk(f, "IEND", nullptr, 0);
    if (!f) throw std::runtime_error("error while writing '" + path + "'");
}
