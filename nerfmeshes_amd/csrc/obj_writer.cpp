// OBJ text writer (host code): the byte-for-byte output of the reference's export_obj
// (/root/reference/src/nerf/nerf_helpers.py:86-111) --
//     v x y z [r g b]      one line per vertex; colours only while the diffuse array lasts
//     vn x y z
//     f a//a b//b c//c     1-based
// with every number printed the way Python's "{}".format(tensor_element) prints it: repr() of the fp32 value widened
// to a double (shortest round-trip digits; fixed notation for 1e-4 <= |x| < 1e16 with a trailing ".0" on integral
// values, otherwise d[.ddd]e+XX).  Once marching cubes runs on the GPU (0.8 ms) the per-element Python writer is the
// bottleneck of mesh_nerf (10.6 s for the 480^3 mesh, 209 MB of text); here the lines are formatted by all host
// threads into per-chunk buffers and written in order.
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "nm_internal.h"

namespace nm {

// repr(float(x)) appended to out
static void append_repr(std::string& out, float x32) {
    const double x = static_cast<double>(x32);
    if (std::isnan(x)) { out += "nan"; return; }
    if (std::isinf(x)) { out += x < 0 ? "-inf" : "inf"; return; }
    char buf[64];
    const auto res = std::to_chars(buf, buf + sizeof(buf) - 1, x, std::chars_format::scientific);   // shortest digits
    *res.ptr = '\0';
    const char* p = buf;
    const char* end = res.ptr;
    if (*p == '-') { out += '-'; ++p; }
    char digits[32];
    int nd = 0;
    const char* e = p;
    for (; e < end && *e != 'e'; ++e)
        if (*e != '.') digits[nd++] = *e;
    const int exp10 = std::atoi(e + 1);
    const int decpt = exp10 + 1;                      // value = 0.d1d2... * 10^decpt
    if (x == 0.0) { out += "0.0"; return; }
    if (decpt <= -4 || decpt > 16) {                  // float_repr_style 'short', format code 'r'
        out += digits[0];
        if (nd > 1) { out += '.'; out.append(digits + 1, nd - 1); }
        const int ex = decpt - 1;
        out += 'e';
        out += ex < 0 ? '-' : '+';
        const int a = ex < 0 ? -ex : ex;
        if (a < 10) out += '0';
        out += std::to_string(a);
    } else if (decpt <= 0) {
        out += "0.";
        out.append(static_cast<size_t>(-decpt), '0');
        out.append(digits, nd);
    } else if (decpt >= nd) {
        out.append(digits, nd);
        out.append(static_cast<size_t>(decpt - nd), '0');
        out += ".0";
    } else {
        out.append(digits, decpt);
        out += '.';
        out.append(digits + decpt, nd - decpt);
    }
}

static void append_triple(std::string& out, const float* v) {
    append_repr(out, v[0]); out += ' ';
    append_repr(out, v[1]); out += ' ';
    append_repr(out, v[2]);
}

template <typename F>
static void parallel_chunks(int64_t n, int threads, std::vector<std::string>& bufs, F&& format_range) {
    const int64_t per = (n + threads - 1) / threads;
    bufs.assign(threads, std::string());
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) {
        const int64_t lo = t * per, hi = lo + per < n ? lo + per : n;
        if (lo >= hi) break;
        pool.emplace_back([&, t, lo, hi] { bufs[t].reserve(static_cast<size_t>(hi - lo) * 64); format_range(bufs[t], lo, hi); });
    }
    for (auto& th : pool) th.join();
}

}  // namespace nm

using namespace nm;

extern "C" int nm_export_obj(const float* h_vertices, int64_t num_vertices, const float* h_diffuse, int64_t num_diffuse,
                             const float* h_normals, int64_t num_normals, const int32_t* h_triangles,
                             int64_t num_triangles, const char* path) {
    NM_REQUIRE(path && num_vertices >= 0 && num_diffuse >= 0 && num_normals >= 0 && num_triangles >= 0, "bad argument");
    NM_REQUIRE((h_vertices || !num_vertices) && (h_diffuse || !num_diffuse) && (h_normals || !num_normals) &&
               (h_triangles || !num_triangles), "null array");
    FILE* f = std::fopen(path, "wb");
    if (!f) { set_error(std::string("cannot open ") + path); return 6; }
    unsigned hw = std::thread::hardware_concurrency();
    const int threads = hw == 0 ? 4 : (hw > 32 ? 32 : static_cast<int>(hw));
    std::vector<std::string> bufs;
    bool ok = true;
    auto flush = [&] { for (const std::string& b : bufs) ok = ok && std::fwrite(b.data(), 1, b.size(), f) == b.size(); };

    parallel_chunks(num_vertices, threads, bufs, [&](std::string& out, int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            out += "v ";
            append_triple(out, h_vertices + 3 * i);
            if (i < num_diffuse) { out += ' '; append_triple(out, h_diffuse + 3 * i); }
            out += '\n';
        }
    });
    flush();
    parallel_chunks(num_normals, threads, bufs, [&](std::string& out, int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) { out += "vn "; append_triple(out, h_normals + 3 * i); out += '\n'; }
    });
    flush();
    parallel_chunks(num_triangles, threads, bufs, [&](std::string& out, int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            out += 'f';
            for (int k = 0; k < 3; ++k) {
                const std::string id = std::to_string(static_cast<int64_t>(h_triangles[3 * i + k]) + 1);
                out += ' '; out += id; out += "//"; out += id;
            }
            out += '\n';
        }
    });
    flush();
    ok = std::fclose(f) == 0 && ok;
    if (!ok) { set_error(std::string("short write to ") + path); return 6; }
    return 0;
}
