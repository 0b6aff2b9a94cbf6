// Armadillo file formats of the reference's on-disk chain batches, written and read without Armadillo.
//
// BFMMM_MTT_warm_start saves every r_stored_iters draws as `<dir>/<Name><q>.txt` (BFMMM.h:1680-1746):
//   cubes / matrices / vectors with `.save(file, arma::arma_ascii)`  ->  "ARMA_CUB_TXT_FN008" / "ARMA_MAT_TXT_FN008"
//   fields of cubes with `.save(file)` (arma_binary)                 ->  "ARMA_FLD_BIN" of "ARMA_CUB_BIN_FN008" objects
// and the package reads them back with ReadVec / ReadMat / ReadCube / ReadFieldCube / ReadFieldMat / ReadFieldVec
// (src/UserFunctions.cpp:2157-2357, plain `.load(file)` with format auto-detection).  Pinned by the files the
// reference ships (inst/test-data/Functional_trace/*, fieldmat.txt, fieldvec.txt): tests/test_arma_io.py reads each of
// them and requires the writer to reproduce it byte for byte.
//
// Text layout (Armadillo diskio::save_arma_ascii): header line, "n_rows n_cols[ n_slices]" line, then per slice one
// line per row, every element as ' ' + width-24 scientific with 16 digits.  Binary objects: header line, dims line,
// raw little-endian column-major doubles.  A field: "ARMA_FLD_BIN", n_rows, n_cols on separate lines, then the
// objects in column-major order of the field, each with its own header.
#include "../../include/bfmmm_entry.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

int bfmmm_io_fail(const std::string& m);      // entry_points.cpp: sets bfmmm_entry_last_error

namespace {

struct Obj {                 // one matrix / cube
  int64_t r = 0, c = 0, s = 1;
  bool cube = false;
  std::vector<double> v;
};

void put_elem(std::string& out, double x) {
  char buf[64];
  if (std::isfinite(x)) snprintf(buf, sizeof buf, " %24.16e", x);
  else snprintf(buf, sizeof buf, " %24s", std::isnan(x) ? "nan" : (x > 0 ? "inf" : "-inf"));     // Armadillo's spellings
  out += buf;
}

std::string ascii_of(const double* d, int64_t r, int64_t c, int64_t s, bool cube) {
  std::string out = cube ? "ARMA_CUB_TXT_FN008\n" : "ARMA_MAT_TXT_FN008\n";
  out += std::to_string(r) + " " + std::to_string(c) + (cube ? " " + std::to_string(s) : "") + "\n";
  out.reserve(out.size() + (size_t)(r * c * s) * 25 + (size_t)(r * s) + 16);
  for (int64_t k = 0; k < s; ++k)
    for (int64_t i = 0; i < r; ++i) {
      for (int64_t j = 0; j < c; ++j) put_elem(out, d[i + r * (j + c * k)]);
      out += '\n';
    }
  return out;
}

std::string binary_of(const double* d, int64_t r, int64_t c, int64_t s, bool cube) {
  std::string out = cube ? "ARMA_CUB_BIN_FN008\n" : "ARMA_MAT_BIN_FN008\n";
  out += std::to_string(r) + " " + std::to_string(c) + (cube ? " " + std::to_string(s) : "") + "\n";
  if (r * c * s > 0) out.append((const char*)d, (size_t)(r * c * s) * sizeof(double));
  return out;
}

int write_file(const char* path, const std::string& bytes) {
  FILE* f = fopen(path, "wb");
  if (!f) return bfmmm_io_fail(std::string("cannot open '") + path + "' for writing");
  const size_t w = fwrite(bytes.data(), 1, bytes.size(), f);
  if (fclose(f) != 0 || w != bytes.size()) return bfmmm_io_fail(std::string("short write to '") + path + "'");
  return 0;
}

struct Reader {
  std::vector<char> b;       // file bytes + one NUL terminator (strtod)
  size_t len = 0, pos = 0;
  bool line(std::string& out) {
    if (pos >= len) return false;
    size_t e = pos;
    while (e < len && b[e] != '\n') ++e;
    out.assign(b.data() + pos, e - pos);
    pos = (e < len) ? e + 1 : e;
    return true;
  }
};

// one matrix / cube object at the reader's position (text or binary)
int read_obj(Reader& rd, Obj& o, const char* path) {
  std::string h, dims;
  if (!rd.line(h) || !rd.line(dims)) return bfmmm_io_fail(std::string("'") + path + "': truncated object header");
  const bool txt = h == "ARMA_MAT_TXT_FN008" || h == "ARMA_CUB_TXT_FN008";
  const bool bin = h == "ARMA_MAT_BIN_FN008" || h == "ARMA_CUB_BIN_FN008";
  if (!txt && !bin) return bfmmm_io_fail(std::string("'") + path + "': unsupported Armadillo header '" + h + "'");
  o.cube = h.compare(5, 3, "CUB") == 0;
  long long r = 0, c = 0, s = 1;
  const int got = sscanf(dims.c_str(), "%lld %lld %lld", &r, &c, &s);
  if (got < (o.cube ? 3 : 2) || r < 0 || c < 0 || s < 0) return bfmmm_io_fail(std::string("'") + path + "': bad dimension line");
  if (!o.cube) s = 1;
  o.r = r; o.c = c; o.s = s;
  const size_t cnt = (size_t)(r * c * s);
  o.v.assign(cnt, 0.0);
  if (bin) {
    if (rd.pos + cnt * sizeof(double) > rd.len) return bfmmm_io_fail(std::string("'") + path + "': truncated binary data");
    memcpy(o.v.data(), rd.b.data() + rd.pos, cnt * sizeof(double));
    rd.pos += cnt * sizeof(double);
    return 0;
  }
  const char* p = rd.b.data() + rd.pos;
  const char* end = rd.b.data() + rd.len;
  for (int64_t k = 0; k < s; ++k)
    for (int64_t i = 0; i < r; ++i)
      for (int64_t j = 0; j < c; ++j) {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
        if (p >= end) return bfmmm_io_fail(std::string("'") + path + "': truncated text data");
        char* q = nullptr;
        o.v[(size_t)(i + r * (j + c * k))] = strtod(p, &q);
        if (q == p) return bfmmm_io_fail(std::string("'") + path + "': malformed number");
        p = q;
      }
  while (p < end && (*p == ' ' || *p == '\r')) ++p;
  if (p < end && *p == '\n') ++p;
  rd.pos = (size_t)(p - rd.b.data());
  return 0;
}

int load(const char* path, Reader& rd) {
  FILE* f = fopen(path, "rb");
  if (!f) return bfmmm_io_fail(std::string("cannot open '") + path + "'");
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  rd.b.assign((size_t)n + 1, '\0');
  const size_t got = fread(rd.b.data(), 1, (size_t)n, f);
  fclose(f);
  if (got != (size_t)n) return bfmmm_io_fail(std::string("short read from '") + path + "'");
  rd.len = (size_t)n;
  return 0;
}

void set_obj(bfmmm_result* r, const std::string& name, const Obj& o) {
  const int64_t d3[3] = {o.r, o.c, o.s};
  bfmmm_result_set(r, name.c_str(), o.v.data(), (int64_t)o.v.size(), d3, o.cube ? 3 : 2);
}

}  // namespace

// ---- internal writers used by the entry points (entry_points.cpp) -------------------------------------------
int arma_save_ascii(const std::string& path, const double* d, int64_t r, int64_t c, int64_t s, bool cube) {
  return write_file(path.c_str(), ascii_of(d, r, c, s, cube));
}

// field of n_rows x n_cols cubes; object (p, k) is cubes[p + n_rows * k]; an empty vector is a 0 x 0 x 0 cube
int arma_save_field_cubes(const std::string& path, const std::vector<std::vector<double>>& cubes, int64_t n_rows, int64_t n_cols,
                          int64_t r, int64_t c, int64_t s) {
  std::string out = "ARMA_FLD_BIN\n" + std::to_string(n_rows) + "\n" + std::to_string(n_cols) + "\n";
  for (const auto& q : cubes) {
    if (q.empty()) out += binary_of(nullptr, 0, 0, 0, true);
    else out += binary_of(q.data(), r, c, s, true);
  }
  return write_file(path.c_str(), out);
}

// ---- internal readers used by the post-processing entry points (post_entry.cpp) --------------------------------
int arma_load_obj(const std::string& path, std::vector<double>& v, int64_t dims[3]) {
  Reader rd;
  if (load(path.c_str(), rd)) return 1;
  Obj o;
  if (read_obj(rd, o, path.c_str())) return 1;
  dims[0] = o.r; dims[1] = o.c; dims[2] = o.s;
  v = std::move(o.v);
  return 0;
}

// objects in the field's column-major order; dims of the first object
int arma_load_field(const std::string& path, std::vector<std::vector<double>>& objs, int64_t* n_rows, int64_t* n_cols, int64_t dims[3]) {
  Reader rd;
  if (load(path.c_str(), rd)) return 1;
  std::string h, l1, l2;
  if (!rd.line(h) || h != "ARMA_FLD_BIN" || !rd.line(l1) || !rd.line(l2)) return bfmmm_io_fail("'" + path + "': not an ARMA_FLD_BIN file");
  const long long nr = atoll(l1.c_str()), nc = atoll(l2.c_str());
  if (nr < 0 || nc < 0) return bfmmm_io_fail("'" + path + "': bad field dimensions");
  objs.clear();
  for (long long e = 0; e < nr * nc; ++e) {
    Obj o;
    if (read_obj(rd, o, path.c_str())) return 1;
    if (e == 0) { dims[0] = o.r; dims[1] = o.c; dims[2] = o.s; }
    objs.push_back(std::move(o.v));
  }
  *n_rows = nr; *n_cols = nc;
  return 0;
}

// ---- C ABI --------------------------------------------------------------------------------------------------
extern "C" int bfmmm_arma_write_ascii(const char* file, const double* data, const int64_t* dims, int n_dims) {
  if (!file || !dims || n_dims < 1 || n_dims > 3) return bfmmm_io_fail("bfmmm_arma_write_ascii: bad arguments");
  const int64_t r = dims[0], c = n_dims > 1 ? dims[1] : 1, s = n_dims > 2 ? dims[2] : 1;
  if ((!data && r * c * s > 0) || r < 0 || c < 0 || s < 0) return bfmmm_io_fail("bfmmm_arma_write_ascii: bad arguments");
  return arma_save_ascii(file, data, r, c, s, n_dims == 3);
}

extern "C" int bfmmm_arma_write_field(const char* file, const bfmmm_result* items, int64_t n_rows, int64_t n_cols) {
  if (!file || !items || n_rows < 0 || n_cols < 0) return bfmmm_io_fail("bfmmm_arma_write_field: bad arguments");
  std::string out = "ARMA_FLD_BIN\n" + std::to_string(n_rows) + "\n" + std::to_string(n_cols) + "\n";
  for (int64_t e = 0; e < n_rows * n_cols; ++e) {
    const double* d; int64_t cnt; const int64_t* dims; int nd;
    if (bfmmm_result_get(items, std::to_string(e).c_str(), &d, &cnt, &dims, &nd)) return 1;
    if (nd < 2 || nd > 3) return bfmmm_io_fail("bfmmm_arma_write_field: items must be matrices or cubes");
    out += binary_of(d, dims[0], dims[1], nd == 3 ? dims[2] : 1, nd == 3);
  }
  return write_file(file, out);
}

// ReadVec / ReadMat / ReadCube (UserFunctions.cpp:2158, :2205, :2253): result element "value"
extern "C" int bfmmm_arma_read(const char* file, bfmmm_result** out) {
  if (!file || !out) return bfmmm_io_fail("bfmmm_arma_read: null argument");
  Reader rd;
  if (load(file, rd)) return 1;
  Obj o;
  if (read_obj(rd, o, file)) return 1;
  bfmmm_result* r = bfmmm_result_create();
  set_obj(r, "value", o);
  *out = r;
  return 0;
}

// ReadFieldCube / ReadFieldMat / ReadFieldVec (UserFunctions.cpp:2303, :2351, :2399): result elements "field_dims"
// (n_rows, n_cols) and "0" .. "n-1" in the field's column-major order
extern "C" int bfmmm_arma_read_field(const char* file, bfmmm_result** out) {
  if (!file || !out) return bfmmm_io_fail("bfmmm_arma_read_field: null argument");
  Reader rd;
  if (load(file, rd)) return 1;
  std::string h, l1, l2;
  if (!rd.line(h) || h != "ARMA_FLD_BIN" || !rd.line(l1) || !rd.line(l2))
    return bfmmm_io_fail(std::string("'") + file + "': not an ARMA_FLD_BIN file");
  const long long nr = atoll(l1.c_str()), nc = atoll(l2.c_str());
  if (nr < 0 || nc < 0) return bfmmm_io_fail(std::string("'") + file + "': bad field dimensions");
  bfmmm_result* r = bfmmm_result_create();
  const double fd[2] = {(double)nr, (double)nc};
  const int64_t two = 2;
  bfmmm_result_set(r, "field_dims", fd, 2, &two, 1);
  for (long long e = 0; e < nr * nc; ++e) {
    Obj o;
    if (read_obj(rd, o, file)) { bfmmm_result_free(r); return 1; }
    set_obj(r, std::to_string(e), o);
  }
  *out = r;
  return 0;
}
