// The command line's text parser / printer (csrc/nrm_tsv.hip: host code only) under g++ -fsanitize=address,undefined (tests/test_cabi_cpu.py builds it beside
// this harness; GPU sanitizers are not available on the pool).  A parser reads what users hand it:
//  * round trips: random matrices of every magnitude (subnormals, +-0, INF, NAN, integers) printed with '%.8G' / '%i' and parsed back, in 1 .. 7 threads, buffers
//    with and without a final newline, comments and blank lines between the rows;
//  * hostile text: a million bytes drawn from the alphabet of numbers, delimiters and line ends, and truncations of valid text at every length: any return code, no
//    read or write outside the buffers (the buffers are exact-size heap blocks: ASan sees one byte too many).
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/normalisr_hip.h"

static char g_err[1024];
void nrm_set_error(const char* fmt, ...) {
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

#define CHECK(c)                                                                  \
	do {                                                                          \
		if (!(c)) {                                                               \
			fprintf(stderr, "%s:%d: check failed: %s (%s)\n", __FILE__, __LINE__, #c, g_err); \
			exit(1);                                                              \
		}                                                                         \
	} while (0)

static unsigned long long g_st = 0x9E3779B97F4A7C15ull;
static unsigned long long rnd() {
	g_st ^= g_st << 13;
	g_st ^= g_st >> 7;
	g_st ^= g_st << 17;
	return g_st;
}
static double rnd_value() {
	switch (rnd() % 12) {
		case 0: return 0.0;
		case 1: return -0.0;
		case 2: return INFINITY;
		case 3: return -INFINITY;
		case 4: return NAN;
		case 5: return (double)((long long)(rnd() % 2000001) - 1000000);
		case 6: return 4.9e-324 * (double)(rnd() % 1000);  // subnormals
		default: {
			const double m = (double)(rnd() >> 11) / 9007199254740992.0 + 0.5;
			return ((rnd() & 1) ? -m : m) * std::pow(10.0, (double)((int)(rnd() % 600) - 300));
		}
	}
}

// parse `text` (exact-size heap copy) with `threads`; returns the status and, on success, the matrix
static int parse(const std::string& text, int threads, std::vector<double>& out, int64_t& rows, int64_t& cols) {
	char* buf = (char*)malloc(text.size() ? text.size() : 1);
	memcpy(buf, text.data(), text.size());
	int rc = nrm_tsv_shape(buf, (int64_t)text.size(), '\t', threads, &rows, &cols);
	if (rc == NRM_OK) {
		out.assign((size_t)(rows * cols), -7.0);
		rc = nrm_tsv_parse(buf, (int64_t)text.size(), '\t', threads, out.data(), NRM_F64, rows, cols, cols);
	}
	free(buf);
	return rc;
}

int main() {
	// round trips
	for (int it = 0; it < 300; it++) {
		const int64_t rows = (int64_t)(rnd() % 40), cols = (int64_t)(rnd() % 9) + (rows ? 1 : 0);
		std::vector<double> m((size_t)(rows * cols));
		for (auto& x : m) x = rnd_value();
		const int parts = 1 + (int)(rnd() % 5);
		const int64_t per = (rows + parts - 1) / parts, cap = per * (cols > 0 ? cols : 1) * nrm_tsv_width(0) + 1;
		std::vector<char> text((size_t)(cap * parts));
		std::vector<int64_t> lens((size_t)parts, 0);
		CHECK(nrm_tsv_format(m.data(), NRM_F64, rows, cols, cols, '\t', 0, text.data(), cap, lens.data(), parts) == NRM_OK);
		std::string s;
		for (int t = 0; t < parts; t++) {
			CHECK(lens[(size_t)t] >= 0 && lens[(size_t)t] <= cap);
			if (rnd() % 3 == 0) s += "# a comment\n\n";
			s.append(text.data() + (size_t)t * cap, (size_t)lens[(size_t)t]);
		}
		if (!s.empty() && s.back() == '\n' && rnd() % 2) s.pop_back();  // loadtxt does not ask for a final newline
		std::vector<double> back;
		int64_t r2 = -1, c2 = -1;
		CHECK(parse(s, 1 + (int)(rnd() % 7), back, r2, c2) == NRM_OK);
		CHECK(r2 == rows && (rows == 0 || c2 == cols));
		for (size_t i = 0; i < back.size(); i++) {
			const double a = m[i], b = back[i];
			if (a != a) CHECK(b != b);
			else if (std::isinf(a) || a == 0.0) CHECK(a == b);
			else CHECK(std::fabs(b - a) <= 5.1e-8 * std::fabs(a) + 5e-324);  // '%.8G': eight significant digits
		}
	}
	// integers
	for (int it = 0; it < 50; it++) {
		const int64_t rows = 1 + (int64_t)(rnd() % 20), cols = 1 + (int64_t)(rnd() % 6);
		std::vector<int64_t> m((size_t)(rows * cols));
		for (auto& x : m) x = (int64_t)rnd() >> (rnd() % 64);
		const int64_t cap = rows * cols * nrm_tsv_width(1) + 1;
		std::vector<char> text((size_t)cap);
		int64_t len = 0;
		CHECK(nrm_tsv_format(m.data(), NRM_TSV_I64, rows, cols, cols, '\t', 1, text.data(), cap, &len, 1) == NRM_OK);
		std::vector<double> back;
		int64_t r2, c2;
		CHECK(parse(std::string(text.data(), (size_t)len), 3, back, r2, c2) == NRM_OK && r2 == rows && c2 == cols);
		for (size_t i = 0; i < back.size(); i++) CHECK(back[i] == (double)m[i]);
	}
	// hostile text
	const char alphabet[] = "0123456789.eE+-\t\n\r #naNiIfFxX,()";
	for (int it = 0; it < 400; it++) {
		std::string s((size_t)(rnd() % 3000), ' ');
		for (auto& ch : s) ch = alphabet[rnd() % (sizeof(alphabet) - 1)];
		std::vector<double> back;
		int64_t r2, c2;
		(void)parse(s, 1 + (int)(rnd() % 7), back, r2, c2);
	}
	{
		const std::string good = "1.5\t-2E-3\tNAN\n# c\n4\t5\t6.25\n\n7\t8\t9\n";
		for (size_t cut = 0; cut <= good.size(); cut++) {
			std::vector<double> back;
			int64_t r2, c2;
			(void)parse(good.substr(0, cut), 1 + (int)(cut % 4), back, r2, c2);
		}
		std::vector<double> back;
		int64_t r2, c2;
		CHECK(parse(good, 2, back, r2, c2) == NRM_OK && r2 == 3 && c2 == 3 && back[1] == -2e-3 && back[8] == 9.0);
		CHECK(parse("1\t2\n3\n", 1, back, r2, c2) == NRM_E_ARG);        // a row of another length
		CHECK(parse("1\tabc\n", 1, back, r2, c2) == NRM_E_ARG);        // not a number
		CHECK(parse("nan(abc)\n", 1, back, r2, c2) == NRM_E_ARG);       // strtod's payload form: loadtxt refuses it
		CHECK(parse("", 1, back, r2, c2) == NRM_OK && r2 == 0);
	}
	printf("text io ok\n");
	return 0;
}
