// Matrix file readers at the C boundary: what the reference keeps beside
// the path in src/spllt_mod.F90:426-620 (mm_double_read, coo_to_csc_double) and takes from
// SPRAL for the drivers' "csc" input (rb_read with rb_options%values = 3,
// drivers/spllt_omp.F90:78-85).  Both return the LOWER triangle as 1-based CSC -- the input of
// spllt_analyse / spllt_factor -- in malloc'ed arrays (spllt_hip_free_matrix).
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <sstream>
#include <string>
#include <vector>

#include "spllt_hip.h"
#include "spllt_iface.h"

namespace {

// values for pattern-only files: the reference invents them with SPRAL's random_real
// (spllt_mod.F90:480-485); here splitmix64 -> uniform in (-1, 1), the same stream as
// spllt_amd/matgen.py invent_values (the two readers are tested against each other)
struct SplitMix {
  uint64_t s;
  explicit SplitMix(uint64_t seed) : s(seed) {}
  double next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return ((double)(z >> 11) + 0.5) * (2.0 / 9007199254740992.0) - 1.0;   // (k + 1/2) / 2^52 - 1
  }
};

struct Coo {
  int i, j;      // 0-based, i >= j (lower)
  double v;
  char up = 0;   // the file listed the entry above the diagonal (MatrixMarket symmetric files)
};

// lower-triangular entries (duplicates summed) -> 1-based CSC with sorted rows; values = 3: every
// diagonal entry becomes 1 + sum |off-diagonal entries of its row of the full symmetric matrix|
// reject_mirrored: an off-diagonal entry listed once below and once above the diagonal (a "symmetric"
// file that stores both triangles) is an error instead of a sum of the two
int finish(int n, std::vector<Coo>& e, int values, int* n_out, int* nnz_out, int** ptr_out, int** row_out,
           double** val_out, bool reject_mirrored = false) {
  std::stable_sort(e.begin(), e.end(), [](const Coo& a, const Coo& b) { return a.j != b.j ? a.j < b.j : a.i < b.i; });
  std::vector<Coo> u;
  u.reserve(e.size() + (size_t)n);
  for (const Coo& c : e) {
    if (!u.empty() && u.back().i == c.i && u.back().j == c.j) {
      if (reject_mirrored && c.i != c.j && u.back().up != c.up) {
        std::fprintf(stderr, "spllt-hip: symmetric file lists entry (%d, %d) in both triangles\n", c.i + 1, c.j + 1);
        return SPLLT_ERROR_PARAMETER;
      }
      u.back().v += c.v;
    } else {
      u.push_back(c);
    }
  }
  if (values == 3) {
    std::vector<double> rs((size_t)n, 0.0);
    std::vector<char> has((size_t)n, 0);
    for (const Coo& c : u) {
      if (c.i == c.j) { has[(size_t)c.i] = 1; continue; }
      rs[(size_t)c.i] += std::fabs(c.v);
      rs[(size_t)c.j] += std::fabs(c.v);
    }
    for (Coo& c : u)
      if (c.i == c.j) c.v = 1.0 + rs[(size_t)c.i];
    bool missing = false;
    for (int k = 0; k < n; ++k)
      if (!has[(size_t)k]) { u.push_back(Coo{k, k, 1.0 + rs[(size_t)k], 0}); missing = true; }
    if (missing)
      std::sort(u.begin(), u.end(), [](const Coo& a, const Coo& b) { return a.j != b.j ? a.j < b.j : a.i < b.i; });
  }
  if (u.size() > (size_t)INT32_MAX) return SPLLT_ERROR_PARAMETER;   // (the reference's C-ABI: int nnz)
  const size_t nnz = u.size();
  int* ptr = (int*)std::malloc(sizeof(int) * ((size_t)n + 1));
  int* row = (int*)std::malloc(sizeof(int) * std::max<size_t>(nnz, 1));
  double* val = (double*)std::malloc(sizeof(double) * std::max<size_t>(nnz, 1));
  if (!ptr || !row || !val) { std::free(ptr); std::free(row); std::free(val); return SPLLT_ERROR_ALLOCATION; }
  std::fill(ptr, ptr + n + 1, 0);
  for (const Coo& c : u) ptr[c.j + 1]++;
  ptr[0] = 1;
  for (int j = 0; j < n; ++j) ptr[j + 1] += ptr[j];
  for (size_t k = 0; k < nnz; ++k) { row[k] = u[k].i + 1; val[k] = u[k].v; }
  *n_out = n; *nnz_out = (int)nnz; *ptr_out = ptr; *row_out = row; *val_out = val;
  return 0;
}

// (count per line, field width) of a Fortran format like (10I8), (1P,3E25.16), (4D20.12), (8F10.3)
bool fortran_fields(const std::string& fmt, int& per, int& width) {
  std::string f;
  for (char c : fmt) if (!std::isspace((unsigned char)c)) f.push_back(c);
  for (size_t k = 0; k < f.size(); ++k) {
    if (!std::isdigit((unsigned char)f[k])) continue;
    size_t a = k;
    while (k < f.size() && std::isdigit((unsigned char)f[k])) ++k;
    if (k < f.size() && std::strchr("IiEeDdFfGg", f[k])) {
      size_t b = k + 1, c = b;
      while (c < f.size() && std::isdigit((unsigned char)f[c])) ++c;
      if (c > b) {
        per = std::atoi(f.substr(a, k - a).c_str());
        width = std::atoi(f.substr(b, c - b).c_str());
        return per > 0 && width > 0;
      }
    }
  }
  return false;
}

template <class Tp, class Conv>
bool read_fixed(const std::vector<std::string>& lines, size_t pos, size_t nlines, const std::string& fmt,
                size_t count, std::vector<Tp>& out, Conv conv) {
  int per = 0, width = 0;
  if (!fortran_fields(fmt, per, width)) return false;
  out.clear();
  // (the header's counts are claims: never more than the lines that are really there can hold)
  if (pos > lines.size() || nlines > lines.size() - pos || count > nlines * (size_t)per) return false;
  out.reserve(count);
  for (size_t l = pos; l < pos + nlines && l < lines.size(); ++l) {
    const std::string& ln = lines[l];
    for (int k = 0; k < per && out.size() < count; ++k) {
      if ((size_t)k * width >= ln.size()) break;
      std::string fld = ln.substr((size_t)k * width, (size_t)width);
      if (fld.find_first_not_of(" \t\r") == std::string::npos) continue;
      out.push_back(conv(fld));
    }
  }
  return out.size() == count;
}

std::vector<std::string> read_lines(const char* path, bool& ok) {
  std::vector<std::string> lines;
  std::ifstream in(path);
  ok = (bool)in;
  std::string ln;
  while (ok && std::getline(in, ln)) {
    while (!ln.empty() && (ln.back() == '\r' || ln.back() == '\n')) ln.pop_back();
    lines.push_back(ln);
  }
  return lines;
}

std::string lower(std::string s) {
  for (char& c : s) c = (char)std::tolower((unsigned char)c);
  return s;
}

// (the C boundary lets no C++ exception through: a header that promises more than memory holds, or
// anything else that throws, is an error flag, not std::terminate)
template <class F>
int guarded(const char* path, F&& f) {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    std::fprintf(stderr, "spllt-hip: %s: out of memory while reading\n", path ? path : "(null)");
    return SPLLT_ERROR_ALLOCATION;
  } catch (...) {
    std::fprintf(stderr, "spllt-hip: %s: malformed file\n", path ? path : "(null)");
    return SPLLT_ERROR_PARAMETER;
  }
}

}  // namespace

extern "C" {

static int read_rb_impl(const char* path, int values, int seed, int* n, int* nnz, int** ptr, int** row,
                        double** val) {
  if (!path || !n || !nnz || !ptr || !row || !val || (values != 0 && values != 3)) return SPLLT_ERROR_PARAMETER;
  bool ok = false;
  std::vector<std::string> lines = read_lines(path, ok);
  if (!ok || lines.size() < 4) {
    std::fprintf(stderr, "spllt-hip: %s: not a Rutherford-Boeing file\n", path);
    return SPLLT_ERROR_PARAMETER;
  }
  long totcrd = 0, ptrcrd = 0, indcrd = 0, valcrd = -1;
  {
    std::istringstream is(lines[1]);
    is >> totcrd >> ptrcrd >> indcrd;
    if (!(is >> valcrd)) valcrd = totcrd - ptrcrd - indcrd;
  }
  std::string mxtype;
  long nrow = 0, ncol = 0, ne = 0;
  {
    std::istringstream is(lines[2]);
    is >> mxtype >> nrow >> ncol >> ne;
    mxtype = lower(mxtype);
  }
  if (nrow > INT32_MAX || ne > INT32_MAX || totcrd < 0 || ptrcrd < 0 || indcrd < 0 || valcrd < -1 ||
      (size_t)ptrcrd > lines.size() || (size_t)indcrd > lines.size() || (valcrd > 0 && (size_t)valcrd > lines.size())) {
    std::fprintf(stderr, "spllt-hip: %s: header counts out of range\n", path);
    return SPLLT_ERROR_PARAMETER;
  }
  if (mxtype.size() != 3 || mxtype[1] != 's' || mxtype[2] != 'a' || nrow != ncol || nrow <= 0 || ne < 0) {
    std::fprintf(stderr, "spllt-hip: %s: only assembled symmetric matrices (?sa) are supported, got '%s'\n", path,
                 mxtype.c_str());
    return SPLLT_ERROR_PARAMETER;
  }
  const std::string& fm = lines[3];
  const std::string ptrfmt = fm.substr(0, 16), indfmt = fm.size() > 16 ? fm.substr(16, 16) : "",
                    valfmt = fm.size() > 32 ? fm.substr(32, 20) : "";
  size_t pos = 4;
  std::vector<long> cp, ri;
  auto to_long = [](const std::string& t) { return std::atol(t.c_str()); };
  if (!read_fixed(lines, pos, (size_t)ptrcrd, ptrfmt, (size_t)ncol + 1, cp, to_long)) return SPLLT_ERROR_PARAMETER;
  pos += (size_t)ptrcrd;
  if (!read_fixed(lines, pos, (size_t)indcrd, indfmt, (size_t)ne, ri, to_long)) return SPLLT_ERROR_PARAMETER;
  pos += (size_t)indcrd;
  const bool has_values = (mxtype[0] == 'r' || mxtype[0] == 'i') && valcrd > 0;
  std::vector<double> v;
  if (has_values) {
    auto to_double = [](std::string t) {
      for (char& c : t) if (c == 'D' || c == 'd') c = 'E';
      return std::atof(t.c_str());
    };
    if (!read_fixed(lines, pos, (size_t)valcrd, valfmt, (size_t)ne, v, to_double)) return SPLLT_ERROR_PARAMETER;
  } else if (values == 0) {
    std::fprintf(stderr, "spllt-hip: %s: pattern-only file and values = 0\n", path);
    return SPLLT_ERROR_PARAMETER;
  } else {
    SplitMix g((uint64_t)seed);
    v.resize((size_t)ne);
    for (double& x : v) x = g.next();
  }
  if (cp[0] != 1 || cp[(size_t)ncol] != ne + 1) return SPLLT_ERROR_PARAMETER;
  std::vector<Coo> e;
  e.reserve((size_t)ne);
  for (long j = 0; j < ncol; ++j) {
    if (cp[(size_t)j + 1] < cp[(size_t)j]) return SPLLT_ERROR_PARAMETER;
    for (long k = cp[(size_t)j] - 1; k < cp[(size_t)j + 1] - 1; ++k) {
      const long i = ri[(size_t)k] - 1;
      if (i < 0 || i >= nrow) return SPLLT_ERROR_PARAMETER;
      if (i < j) continue;                      // the stored triangle is the lower one; anything above is dropped
      e.push_back(Coo{(int)i, (int)j, v[(size_t)k], 0});
    }
  }
  return finish((int)nrow, e, values, n, nnz, ptr, row, val);
}

int spllt_hip_read_rb(const char* path, int values, int seed, int* n, int* nnz, int** ptr, int** row,
                      double** val) {
  return guarded(path, [&] { return read_rb_impl(path, values, seed, n, nnz, ptr, row, val); });
}

static int read_mm_impl(const char* path, int values, int seed, int* n, int* nnz, int** ptr, int** row,
                        double** val) {
  if (!path || !n || !nnz || !ptr || !row || !val || (values != 0 && values != 3)) return SPLLT_ERROR_PARAMETER;
  std::ifstream in(path);
  if (!in) {
    std::fprintf(stderr, "spllt-hip: %s: cannot open\n", path);
    return SPLLT_ERROR_PARAMETER;
  }
  std::string ln;
  if (!std::getline(in, ln)) return SPLLT_ERROR_PARAMETER;
  std::string banner, obj, rep, field, symm;
  {
    std::istringstream is(ln);
    is >> banner >> obj >> rep >> field >> symm;
  }
  rep = lower(rep); field = lower(field); symm = lower(symm);
  if (lower(banner) != "%%matrixmarket" || rep != "coordinate" ||
      (field != "real" && field != "integer" && field != "pattern") || (symm != "symmetric" && symm != "general")) {
    std::fprintf(stderr, "spllt-hip: %s: unsupported MatrixMarket header '%s'\n", path, ln.c_str());
    return SPLLT_ERROR_PARAMETER;
  }
  while (std::getline(in, ln))
    if (!ln.empty() && ln[0] != '%') break;
  long m = 0, nc = 0, ne = 0;
  {
    std::istringstream is(ln);
    if (!(is >> m >> nc >> ne) || m != nc || m <= 0 || ne < 0) return SPLLT_ERROR_PARAMETER;
    if (m > INT32_MAX || ne > INT32_MAX) {      // (the reference's C-ABI: int n, int nnz)
      std::fprintf(stderr, "spllt-hip: %s: size line %ld %ld %ld beyond the 32-bit interface\n", path, m, nc, ne);
      return SPLLT_ERROR_PARAMETER;
    }
  }
  // an entry takes at least four bytes ("1 1\n"): the header cannot promise more than the file holds
  {
    const std::streampos here = in.tellg();
    in.seekg(0, std::ios::end);
    const std::streampos end = in.tellg();
    in.seekg(here);
    if (here >= 0 && end >= here && (long long)ne > ((long long)(end - here) + 3) / 4) {
      std::fprintf(stderr, "spllt-hip: %s: the size line promises %ld entries, the file is too short for them\n", path, ne);
      return SPLLT_ERROR_PARAMETER;
    }
  }
  const bool pattern = field == "pattern";
  if (pattern && values == 0) {
    std::fprintf(stderr, "spllt-hip: %s: pattern-only file and values = 0\n", path);
    return SPLLT_ERROR_PARAMETER;
  }
  SplitMix g((uint64_t)seed);
  std::vector<Coo> e;
  e.reserve((size_t)ne);
  for (long k = 0; k < ne; ++k) {
    long i = 0, j = 0;
    double x = 0.0;
    if (!(in >> i >> j)) return SPLLT_ERROR_PARAMETER;
    if (pattern) x = g.next();                      // (reference spllt_mod.F90:480-485: values made up)
    else if (!(in >> x)) return SPLLT_ERROR_PARAMETER;
    if (i < 1 || j < 1 || i > m || j > m) return SPLLT_ERROR_PARAMETER;
    --i; --j;
    // symmetric: the entry stands for both triangles; general: A := (A + A^T) / 2, i.e. every
    // off-diagonal entry contributes half to the lower-triangular position of its pair
    const double w = (symm == "general" && i != j) ? 0.5 * x : x;
    e.push_back(Coo{(int)std::max(i, j), (int)std::min(i, j), w, (char)(i < j)});
  }
  return finish((int)m, e, values, n, nnz, ptr, row, val, symm == "symmetric");
}

int spllt_hip_read_mm(const char* path, int values, int seed, int* n, int* nnz, int** ptr, int** row,
                      double** val) {
  return guarded(path, [&] { return read_mm_impl(path, values, seed, n, nnz, ptr, row, val); });
}

void spllt_hip_free_matrix(int* ptr, int* row, double* val) {
  std::free(ptr);
  std::free(row);
  std::free(val);
}

}  // extern "C"
