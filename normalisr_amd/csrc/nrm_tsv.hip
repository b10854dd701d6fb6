// Text matrices of the command line (host only, no device code): the reference reads its inputs with numpy.loadtxt and writes its
// results with numpy.savetxt('%.8G') (run.py:20-35) -- tab-delimited, no header, one row per line.  Both go through Python objects
// number by number: BASELINE configs[1] as files (5000 x 10 000 in, two 5000 x 5000 out) spends over a minute there and 2.5 ms in the
// association itself.  Here: the file's bytes are cut at line ends into one piece per thread, std::from_chars (correctly rounded, as
// Python's float()) fills the matrix in place; results are printed with snprintf("%.8G"), which is the C formatting Python's '%' operator
// follows digit for digit, a block of rows per thread.  The contract stays the reference's: same text in, same text out.
#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "nrm_common.h"

namespace {

struct Piece {
	const char* a;
	const char* b;
	int64_t rows;  // data lines inside
};

inline bool blank(char ch) { return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\v' || ch == '\f'; }

// [a, e): one line without its '\n'; the part from '#' on is a comment (numpy.loadtxt's default); returns the end of the data part
// with trailing blanks (and '\r') cut off, a == end for a line without data
inline const char* data_end(const char* a, const char* e) {
	const char* h = (const char*)memchr(a, '#', (size_t)(e - a));
	if (h) e = h;
	while (e > a && blank(e[-1])) e--;
	return e;
}

inline bool has_data(const char* a, const char* e, char delim) {
	e = data_end(a, e);
	for (const char* p = a; p < e; p++)
		if (!blank(*p) || *p == delim) return true;  // (a delimiter is data: a line of tabs is a row of empty fields, which is an error later)
	return false;
}

int64_t count_rows(const char* a, const char* b, char delim) {
	int64_t rows = 0;
	while (a < b) {
		const char* e = (const char*)memchr(a, '\n', (size_t)(b - a));
		if (!e) e = b;
		if (has_data(a, e, delim)) rows++;
		a = e + 1;
	}
	return rows;
}

std::vector<Piece> cut(const char* buf, int64_t len, int threads, char delim) {
	const char* end = buf + len;
	int t = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
	if (t > 64) t = 64;
	const int64_t by_size = len / (1 << 20) + 1;  // a thread per MB at least
	if (t > by_size) t = (int)by_size;
	if (t < 1) t = 1;
	std::vector<Piece> ps;
	const char* a = buf;
	for (int i = 0; i < t && a < end; i++) {
		const char* b = i == t - 1 ? end : buf + (len * (i + 1)) / t;
		if (b < a) b = a;
		if (b < end) {
			const char* nl = (const char*)memchr(b, '\n', (size_t)(end - b));
			b = nl ? nl + 1 : end;
		}
		ps.push_back({a, b, 0});
		a = b;
	}
	std::vector<std::thread> th;
	for (size_t i = 1; i < ps.size(); i++) th.emplace_back([&ps, i, delim] { ps[i].rows = count_rows(ps[i].a, ps[i].b, delim); });
	if (!ps.empty()) ps[0].rows = count_rows(ps[0].a, ps[0].b, delim);
	for (auto& x : th) x.join();
	return ps;
}

// one field -> double, as Python's float(): blanks around it, an optional '+', inf / nan in any case, out-of-range -> +-inf / 0
// (Only what float() takes: a NaN spelled with a payload, "nan(abc)", parses for from_chars / strtod and raises for numpy.loadtxt; a '\r' inside a
// line is a line break for numpy's text reader -- neither is a blank here, so such text fails to parse and the caller hands the file to
// numpy.loadtxt itself, whose answer or exception then is the reference's: run.py, _read_text.)
inline bool inner_blank(char ch) { return ch == ' ' || ch == '\t' || ch == '\v' || ch == '\f'; }
inline bool plain_nan(const char* a, const char* e) {
	if (a < e && (*a == '+' || *a == '-')) a++;
	return e - a == 3 && (a[0] | 32) == 'n' && (a[1] | 32) == 'a' && (a[2] | 32) == 'n';
}
inline bool field(const char* a, const char* e, double& v) {
	while (a < e && inner_blank(*a)) a++;
	while (e > a && inner_blank(e[-1])) e--;
	if (a == e) return false;
	auto r = std::from_chars(a, e, v);
	if (r.ec == std::errc() && r.ptr == e) return v == v || plain_nan(a, e);
	if (e - a > 400) return false;
	char tmp[408];
	memcpy(tmp, a, (size_t)(e - a));
	tmp[e - a] = 0;
	const char* s = tmp;
	if (*s == '+' && s[1] != '+' && s[1] != '-') s++;
	if (*s == 0 || blank(*s)) return false;
	if ((s[0] == '0' && (s[1] == 'x' || s[1] == 'X')) || (s[0] == '-' && s[1] == '0' && (s[2] == 'x' || s[2] == 'X'))) return false;  // float() takes no hex
	char* stop = nullptr;
	v = strtod(s, &stop);
	return stop && *stop == 0 && stop != s && (v == v || plain_nan(a, e));
}

struct Fail {
	int64_t row = -1, col = -1, got = -1;
	std::string text;
};

template <typename O>
void parse_piece(const Piece& p, char delim, O* out, int64_t ld, int64_t cols, int64_t row0, Fail& fail) {
	const char* a = p.a;
	int64_t row = row0;
	while (a < p.b) {
		const char* e = (const char*)memchr(a, '\n', (size_t)(p.b - a));
		if (!e) e = p.b;
		if (has_data(a, e, delim)) {
			const char* de = data_end(a, e);
			O* o = out + row * ld;
			int64_t c = 0;
			const char* f = a;
			while (true) {
				const char* g = (const char*)memchr(f, delim, (size_t)(de - f));
				const char* fe = g ? g : de;
				double v;
				if (c >= cols || !field(f, fe, v)) {
					if (fail.row < 0) {
						fail.row = row;
						fail.col = c;
						if (c >= cols) {  // more fields than the first row has
							int64_t n = c + 1;
							for (const char* q = fe; q < de; q++) n += *q == delim;
							fail.got = n;
						} else
							fail.text.assign(f, (size_t)((fe - f) < 40 ? (fe - f) : 40));
					}
					return;
				}
				o[c++] = (O)v;
				if (!g) break;
				f = g + 1;
			}
			if (c != cols) {
				if (fail.row < 0) {
					fail.row = row;
					fail.col = c;
					fail.got = c;
				}
				return;
			}
			row++;
		}
		a = e + 1;
	}
}

}  // namespace

// Shape of the matrix in a text buffer: rows = lines with data (blank lines and '#' comments skipped, as numpy.loadtxt), cols = fields
// of the first such line.
extern "C" int nrm_tsv_shape(const char* buf, int64_t len, int delim, int threads, int64_t* rows, int64_t* cols) {
	NRM_REQUIRE(len >= 0 && (buf || len == 0) && rows && cols, "nrm_tsv_shape: bad arguments");
	const char dl = (char)delim;
	auto ps = cut(buf, len, threads, dl);
	int64_t r = 0;
	for (auto& p : ps) r += p.rows;
	*rows = r;
	*cols = 0;
	const char* a = buf;
	const char* end = buf + len;
	while (a < end) {
		const char* e = (const char*)memchr(a, '\n', (size_t)(end - a));
		if (!e) e = end;
		if (has_data(a, e, dl)) {
			const char* de = data_end(a, e);
			int64_t c = 1;
			for (const char* q = a; q < de; q++) c += *q == dl;
			*cols = c;
			break;
		}
		a = e + 1;
	}
	return NRM_OK;
}

// The matrix itself into out (rows, ld) of dtype NRM_F32 / NRM_F64; rows and cols as nrm_tsv_shape reported them.  A field that is
// not a number or a row with another number of fields: NRM_E_ARG with the place in the message (ValueError on the Python side, as
// numpy.loadtxt raises).
extern "C" int nrm_tsv_parse(const char* buf, int64_t len, int delim, int threads, void* out, int out_dtype, int64_t rows, int64_t cols, int64_t ld) {
	NRM_REQUIRE(len >= 0 && (buf || len == 0) && rows >= 0 && cols >= 0 && ld >= cols && (out || rows * cols == 0), "nrm_tsv_parse: bad arguments");
	NRM_REQUIRE(out_dtype == NRM_F32 || out_dtype == NRM_F64, "nrm_tsv_parse: bad dtype");
	const char dl = (char)delim;
	auto ps = cut(buf, len, threads, dl);
	int64_t total = 0;
	std::vector<int64_t> row0(ps.size());
	for (size_t i = 0; i < ps.size(); i++) {
		row0[i] = total;
		total += ps[i].rows;
	}
	NRM_REQUIRE(total == rows, "nrm_tsv_parse: %lld rows in the text, %lld expected", (long long)total, (long long)rows);
	std::vector<Fail> fails(ps.size());
	auto work = [&](size_t i) {
		if (out_dtype == NRM_F64)
			parse_piece<double>(ps[i], dl, (double*)out, ld, cols, row0[i], fails[i]);
		else
			parse_piece<float>(ps[i], dl, (float*)out, ld, cols, row0[i], fails[i]);
	};
	std::vector<std::thread> th;
	for (size_t i = 1; i < ps.size(); i++) th.emplace_back(work, i);
	if (!ps.empty()) work(0);
	for (auto& x : th) x.join();
	for (auto& f : fails)
		if (f.row >= 0) {
			if (f.got >= 0)
				nrm_set_error("the number of columns changed from %lld to %lld at row %lld", (long long)cols, (long long)f.got, (long long)f.row + 1);
			else
				nrm_set_error("could not convert string '%s' to float64 at row %lld, column %lld", f.text.c_str(), (long long)f.row, (long long)f.col + 1);
			return NRM_E_ARG;
		}
	return NRM_OK;
}

// Widest text of one value and its delimiter: "-1.2345678E-308" + 1; integers up to 20 digits and a sign + 1
extern "C" int64_t nrm_tsv_width(int kind) { return kind == 0 ? 16 : 22; }

namespace {

template <typename T>
inline int put(char* o, T v, int kind);
// '%.8G' of a double, digit for digit what printf / Python's '%' operator print: the 8 significant digits come from std::to_chars
// (scientific, precision 7: exact, round-half-even, as printf rounds), the layout rules of %G are applied here -- exponent form when the
// decimal exponent is < -4 or >= 8, trailing zeros dropped, at least two exponent digits.  (glibc's snprintf takes 1 - 3 us per number
// here -- multi-precision arithmetic for every value; this takes ~0.1.)
template <>
inline int put<double>(char* o, double v, int) {
	if (v != v) {  // Python prints NAN whatever the sign bit says; printf may print -NAN
		memcpy(o, "NAN", 3);
		return 3;
	}
	char* p = o;
	if (std::signbit(v)) {
		*p++ = '-';
		v = -v;
	}
	if (v == 0.0) {
		*p++ = '0';
		return (int)(p - o);
	}
	if (std::isinf(v)) {
		memcpy(p, "INF", 3);
		return (int)(p - o) + 3;
	}
	// The digits: the shortest decimal that reads back as v (std::to_chars, ~50 ns) rounded to 8 places.  That is the rounding of v
	// itself unless the shortest decimal IS the tie (8 digits and a 5): no other decimal of <= 9 digits lies as close to v as a
	// double's neighbours are, so v and its shortest decimal are on the same side of every other tie.  On the tie the exact
	// expansion decides (to_chars with a precision: exact, but 10 - 50 times slower; it was the whole formatter at first).
	char t[48];
	auto r = std::to_chars(t, t + 48, v, std::chars_format::scientific);  // d[.ddd...]e[+-]XX
	char dg[9];
	int x = 0, have = 0;
	const char* q = t;
	bool more = false, tie = false;
	for (; q < r.ptr && *q != 'e'; q++) {
		if (*q == '.') continue;
		if (have < 9)
			dg[have++] = *q;
		else if (*q != '0')
			more = true;
	}
	{
		q++;  // past 'e'
		const bool neg = *q == '-';
		for (q++; q < r.ptr; q++) x = x * 10 + (*q - '0');
		if (neg) x = -x;
	}
	if (v < 2.2250738585072014e-308)
		tie = true;  // subnormal: its neighbours are far apart (few bits), the argument above does not hold -- the exact expansion
	else if (have == 9) {
		const bool up = dg[8] > '5' || (dg[8] == '5' && more);
		tie = dg[8] == '5' && !more;
		if (up) {
			int i = 7;
			while (i >= 0 && dg[i] == '9') dg[i--] = '0';
			if (i >= 0)
				dg[i]++;
			else {  // 99999999|5.. -> 1.0000000 and the next exponent
				dg[0] = '1';
				x++;
			}
		}
	} else
		while (have < 8) dg[have++] = '0';
	if (tie) {
		auto r2 = std::to_chars(t, t + 48, v, std::chars_format::scientific, 7);  // d.ddddddde[+-]XX
		dg[0] = t[0];
		memcpy(dg + 1, t + 2, 7);
		x = 0;
		q = t + 10;
		const bool neg = *q == '-';
		for (q++; q < r2.ptr; q++) x = x * 10 + (*q - '0');
		if (neg) x = -x;
	}
	int nd = 8;
	while (nd > 1 && dg[nd - 1] == '0') nd--;
	if (x < -4 || x >= 8) {
		*p++ = dg[0];
		if (nd > 1) {
			*p++ = '.';
			memcpy(p, dg + 1, (size_t)(nd - 1));
			p += nd - 1;
		}
		*p++ = 'E';
		*p++ = x < 0 ? '-' : '+';
		int ax = x < 0 ? -x : x;
		if (ax >= 100) {
			*p++ = (char)('0' + ax / 100);
			ax %= 100;
			*p++ = (char)('0' + ax / 10);
		} else
			*p++ = (char)('0' + ax / 10);
		*p++ = (char)('0' + ax % 10);
	} else if (x >= 0) {
		for (int i = 0; i <= x; i++) *p++ = i < nd ? dg[i] : '0';
		if (nd > x + 1) {
			*p++ = '.';
			memcpy(p, dg + x + 1, (size_t)(nd - x - 1));
			p += nd - x - 1;
		}
	} else {
		*p++ = '0';
		*p++ = '.';
		for (int i = 0; i < -x - 1; i++) *p++ = '0';
		memcpy(p, dg, (size_t)nd);
		p += nd;
	}
	return (int)(p - o);
}
template <>
inline int put<float>(char* o, float v, int) {
	return put<double>(o, (double)v, 0);  // the value itself, widened: what Python's float(numpy.float32) holds
}
template <>
inline int put<int64_t>(char* o, int64_t v, int) {
	auto r = std::to_chars(o, o + 24, (long long)v);
	return (int)(r.ptr - o);
}
template <>
inline int put<int32_t>(char* o, int32_t v, int) {
	auto r = std::to_chars(o, o + 24, v);
	return (int)(r.ptr - o);
}
template <>
inline int put<uint8_t>(char* o, uint8_t v, int) {
	auto r = std::to_chars(o, o + 24, (unsigned)v);
	return (int)(r.ptr - o);
}

template <typename T>
int64_t format_rows(const T* d, int64_t r0, int64_t r1, int64_t cols, int64_t ld, char delim, int kind, char* o) {
	char* p = o;
	for (int64_t r = r0; r < r1; r++) {
		const T* x = d + r * ld;
		for (int64_t c = 0; c < cols; c++) {
			p += put<T>(p, x[c], kind);
			*p++ = c + 1 < cols ? delim : '\n';
		}
		if (cols == 0) *p++ = '\n';
	}
	return p - o;
}

}  // namespace

// rows x cols of data (dtype: NRM_F32, NRM_F64 with kind 0 = '%.8G'; NRM_TSV_I64 / _I32 / _U8 with kind 1 = '%i') as text, one row per
// line.  The rows are dealt to `parts` threads in order; part t writes at out + t * part_cap and reports its length in lens[t], so the
// caller writes the parts one after the other.  part_cap >= ceil(rows / parts) * max(cols, 1) * nrm_tsv_width(kind).
extern "C" int nrm_tsv_format(const void* data, int dtype, int64_t rows, int64_t cols, int64_t ld, int delim, int kind, char* out, int64_t part_cap,
							  int64_t* lens, int parts) {
	NRM_REQUIRE(rows >= 0 && cols >= 0 && ld >= cols && parts >= 1 && out && lens && (data || rows * cols == 0), "nrm_tsv_format: bad arguments");
	NRM_REQUIRE((kind == 0 && (dtype == NRM_F32 || dtype == NRM_F64)) || (kind == 1 && (dtype == NRM_TSV_I64 || dtype == NRM_TSV_I32 || dtype == NRM_TSV_U8)),
				"nrm_tsv_format: dtype %d does not go with format kind %d", dtype, kind);
	const int64_t per = (rows + parts - 1) / parts;
	NRM_REQUIRE(part_cap >= per * (cols > 0 ? cols : 1) * nrm_tsv_width(kind), "nrm_tsv_format: part_cap too small");
	auto work = [&](int t) {
		const int64_t r0 = per * t < rows ? per * t : rows, r1 = r0 + per < rows ? r0 + per : rows;
		char* o = out + (int64_t)t * part_cap;
		const char dl = (char)delim;
		switch (dtype) {
			case NRM_F64: lens[t] = format_rows<double>((const double*)data, r0, r1, cols, ld, dl, kind, o); break;
			case NRM_F32: lens[t] = format_rows<float>((const float*)data, r0, r1, cols, ld, dl, kind, o); break;
			case NRM_TSV_I64: lens[t] = format_rows<int64_t>((const int64_t*)data, r0, r1, cols, ld, dl, kind, o); break;
			case NRM_TSV_I32: lens[t] = format_rows<int32_t>((const int32_t*)data, r0, r1, cols, ld, dl, kind, o); break;
			default: lens[t] = format_rows<uint8_t>((const uint8_t*)data, r0, r1, cols, ld, dl, kind, o); break;
		}
	};
	std::vector<std::thread> th;
	for (int t = 1; t < parts; t++) th.emplace_back(work, t);
	work(0);
	for (auto& x : th) x.join();
	return NRM_OK;
}
