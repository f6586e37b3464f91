// gsalign_amd/csrc/host/synth.cpp -- synthetic genomes for the benchmark and the size tests
// (SURVEY.md section 8(d)): workload tooling, not part of the aligner.
//
//   reference : i.i.d. uniform ACGT, optionally with the repeat-stress injection -- a 300-bp family whose
//               copies (10 % divergent from the family sequence) cover a given fraction of the genome, plus
//               one tandem array of a 40-bp unit with more than MaxSeedFreq (100, bwt_search.cpp:3) copies;
//   query     : the reference with per-base events at total rate d -- 80 % substitutions (uniform over the
//               three other bases), 10 % insertions of U[1,10] random bases, 10 % deletions of U[1,10] bases.
//
// Counter-based RNG (splitmix64 of seed and position): the output depends on (seed, position) only, so a
// 250 Mb genome takes about a second and any slice can be regenerated independently.
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

inline uint64_t mix64(uint64_t x)
{
	x += 0x9E3779B97F4A7C15ull;
	x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
	x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
	return x ^ (x >> 31);
}
inline uint64_t key_of(uint64_t seed, uint64_t stream) { return mix64(seed * 0x2545F4914F6CDD1Dull + stream); }
inline uint64_t rnd(uint64_t seed, uint64_t stream, uint64_t i) { return mix64(key_of(seed, stream) ^ i); }
const char ACGT[5] = "ACGT";
inline int code_of(char c) { switch (c) { case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; default: return 3; } }

// the event mix on src[0..n) written to o[0..cap); returns the length, or -1 when cap is too small.
// `stream` separates independent uses of one seed.
int64_t mutate_into(const char *src, int64_t n, double d, uint64_t seed, uint64_t stream, char *o, int64_t cap)
{
	const uint64_t key = key_of(seed, stream), key2 = key_of(seed, stream + 1);
	// the event is decided on the top 32 bits: thresholds 0.8 d, 0.9 d, d of 2^32
	const uint32_t t_sub = (uint32_t)(0.8 * d * 4294967296.0), t_ins = (uint32_t)(0.9 * d * 4294967296.0), t_del = (uint32_t)(d * 4294967296.0);
	int64_t w = 0;
	for (int64_t i = 0; i < n;) {
		if (w + 12 > cap) return -1;
		const uint64_t r = mix64(key ^ (uint64_t)i);
		const uint32_t ev = (uint32_t)(r >> 32);
		if (ev >= t_del) { o[w++] = src[i++]; continue; }
		if (ev < t_sub) {
			// (an N stays an N, a soft-masked base stays lower case)
			const char c = src[i], sub = ACGT[(code_of(c) + 1 + (int)((r & 0xffff) % 3)) & 3];
			o[w++] = (c == 'N' || c == 'n') ? c : ((c >= 'a' && c <= 'z') ? (char)(sub | 0x20) : sub); i++;
		}
		else if (ev < t_ins) {
			const int len = 1 + (int)((r & 0xffff) % 10);
			for (int k = 0; k < len; k++) o[w++] = ACGT[mix64(key2 ^ ((uint64_t)i * 16 + k)) & 3];
			o[w++] = src[i++];
		} else i += 1 + (int)((r & 0xffff) % 10);
	}
	return w;
}

} // namespace

extern "C" {

void gsah_c_synth_genome(int64_t n, uint64_t seed, char *out)
{
	const uint64_t key = key_of(seed, 0);
	for (int64_t i = 0; i < n; i += 32) {
		uint64_t r = mix64(key ^ (uint64_t)(i >> 5));
		const int64_t e = i + 32 < n ? i + 32 : n;
		for (int64_t p = i; p < e; p++, r >>= 2) out[p] = ACGT[r & 3];
	}
}

// Repeat-stress injection, in place.  Copies of one fam_len-bp family (each copy_div divergent from the family
// sequence, same event mix, cut or padded to fam_len) are written at random positions until they cover `frac` of
// the genome; then one tandem array of tandem_copies x tandem_unit bases in the middle.  Returns the copy count.
int64_t gsah_c_synth_repeats(char *seq, int64_t n, uint64_t seed, double frac, int fam_len, double copy_div, int tandem_unit, int tandem_copies)
{
	if (fam_len <= 0 || n < 4 * (int64_t)fam_len) return 0;
	std::vector<char> fam((size_t)fam_len);
	gsah_c_synth_genome(fam_len, seed ^ 0xFA111ull, fam.data());
	const int64_t copies = (int64_t)(frac * (double)n / fam_len);
	std::vector<char> cp((size_t)fam_len * 12 + 64);
	for (int64_t c = 0; c < copies; c++) {
		int64_t got = mutate_into(fam.data(), fam_len, copy_div, seed, 1000 + 2 * (uint64_t)c, cp.data(), (int64_t)cp.size());
		if (got < 0) got = 0;
		for (; got < fam_len; got++) cp[(size_t)got] = ACGT[rnd(seed, 7, (uint64_t)c * 64 + (uint64_t)got) & 3];
		const int64_t pos = (int64_t)(rnd(seed, 3, (uint64_t)c) % (uint64_t)(n - fam_len));
		memcpy(seq + pos, cp.data(), (size_t)fam_len);
	}
	const int64_t tl = (int64_t)tandem_unit * tandem_copies;
	if (tandem_unit > 0 && tl > 0 && tl < n / 2) {
		std::vector<char> u((size_t)tandem_unit);
		gsah_c_synth_genome(tandem_unit, seed ^ 0x7A2DE3ull, u.data());
		const int64_t p0 = n / 2 - tl / 2;
		for (int64_t k = 0; k < tl; k++) seq[p0 + k] = u[(size_t)(k % tandem_unit)];
	}
	return copies;
}

// Adversarial injection, in place: what real genomes do to a seed search and the i.i.d. text does not.
//   * n_fam repeat families with a copy-number spectrum: family f has length L_f (80 .. 6000 bp), divergence d_f of its copies
//     from the family sequence (1 % .. 15 %: a young family with thousands of near-identical copies is the `freq > MaxSeedFreq`
//     reject-and-restart regime of bwt_search.cpp:177-182 on every start inside a copy) and a copy count from max_copies down by a
//     factor ~3 per family; together they cover about `frac` of the sequence;
//   * microsatellites: short tandem arrays (unit 1-6 bp, 20-300 bp long), one per ~15 kb;
//   * two runs of N (a centromere-like gap), each n_run bases, at 1/3 and 2/3 of the sequence (packed as random bases by the index
//     builder, bntseq.c:159-176, and skipped by the seed search on the query side);
//   * soft-masked blocks: lower-case stretches of 200-5000 bases, one per ~60 kb (nst_nt4_table folds case: same alignment).
// Returns the number of family copies written.
int64_t gsah_c_synth_adversarial(char *seq, int64_t n, uint64_t seed, double frac, int n_fam, int64_t max_copies, int64_t n_run)
{
	if (n < 100000 || n_fam <= 0) return 0;
	static const int fam_len[8] = { 300, 1200, 150, 6000, 80, 500, 2500, 300 };
	static const double fam_div[8] = { 0.01, 0.05, 0.03, 0.10, 0.02, 0.15, 0.08, 0.12 };
	int64_t total = 0;
	// copy counts: max_copies, max_copies / 3, ... scaled so that the families cover `frac` of the sequence
	std::vector<int64_t> copies((size_t)n_fam);
	{
		double bases = 0; int64_t c = max_copies;
		for (int f = 0; f < n_fam; f++) { copies[(size_t)f] = c < 2 ? 2 : c; bases += (double)copies[(size_t)f] * fam_len[f & 7]; c /= 3; }
		const double scale = frac * (double)n / bases;
		if (scale < 1.0) for (int f = 0; f < n_fam; f++) { copies[(size_t)f] = (int64_t)((double)copies[(size_t)f] * scale); if (copies[(size_t)f] < 2) copies[(size_t)f] = 2; }
	}
	for (int f = 0; f < n_fam; f++) {
		const int L = fam_len[f & 7];
		if (n < 8 * (int64_t)L) continue;
		std::vector<char> fam((size_t)L), cp((size_t)L * 12 + 64);
		gsah_c_synth_genome(L, seed ^ (0xFA111ull + 977ull * (uint64_t)f), fam.data());
		for (int64_t c = 0; c < copies[(size_t)f]; c++) {
			int64_t got = mutate_into(fam.data(), L, fam_div[f & 7], seed + 31 * (uint64_t)f, 1000 + 2 * (uint64_t)c, cp.data(), (int64_t)cp.size());
			if (got < 0) got = 0;
			for (; got < L; got++) cp[(size_t)got] = ACGT[rnd(seed, 7 + (uint64_t)f, (uint64_t)c * 8192 + (uint64_t)got) & 3];
			const int64_t pos = (int64_t)(rnd(seed, 300 + (uint64_t)f, (uint64_t)c) % (uint64_t)(n - L));
			memcpy(seq + pos, cp.data(), (size_t)L);
			total++;
		}
	}
	for (int64_t k = 0; k < n / 15000; k++) {                              // microsatellites
		const uint64_t r = rnd(seed, 51, (uint64_t)k);
		const int unit = 1 + (int)(r % 6), len = 20 + (int)((r >> 8) % 281);
		const int64_t pos = (int64_t)((r >> 20) % (uint64_t)(n - len));
		char u[6]; for (int t = 0; t < unit; t++) u[t] = ACGT[(r >> (40 + 2 * t)) & 3];
		for (int t = 0; t < len; t++) seq[pos + t] = u[t % unit];
	}
	for (int64_t k = 0; k < n / 60000; k++) {                              // soft-masked blocks
		const uint64_t r = rnd(seed, 52, (uint64_t)k);
		const int len = 200 + (int)(r % 4801);
		const int64_t pos = (int64_t)((r >> 16) % (uint64_t)(n - len));
		for (int t = 0; t < len; t++) { const char c = seq[pos + t]; if (c >= 'A' && c <= 'Z' && c != 'N') seq[pos + t] = (char)(c | 0x20); }
	}
	if (n_run > 0 && 8 * n_run < n) for (int g = 1; g <= 2; g++) memset(seq + g * (n / 3) - n_run / 2, 'N', (size_t)n_run);
	return total;
}

// Human-like injection, in place ("a realistic sibling" of the i.i.d. headline workload): the interspersed-repeat spectrum of a
// primate genome, ~45 % of the sequence, instead of the adversarial spectrum's few very young families:
//   * an Alu-like family (300 bp), 10 % of the sequence, three age classes sharing one consensus: 15 % of the copies 3 % diverged from it, 50 % 8 %, 35 % 14 %;
//   * an L1-like family (6 kb consensus), 17 %: copies are 5'-truncated (length 6 kb x u^2, at least 100 bp, taken from the 3' end), 3 - 20 % diverged;
//   * two LTR/ERV-like families (5 kb and 7 kb; four copies in five are solo LTRs: the first 500 bp), 8 %, 5 - 15 %;
//   * three ancient families (MIR / L2 / DNA-transposon-like, 200-bp pieces, 25 - 30 % diverged: unique at seed level), 5 %;
//   * segmental duplications: 20-kb blocks copied from elsewhere in the sequence at 1 - 3 % divergence, 3 %;
//   * microsatellites (unit 1 - 6 bp, 20 - 300 bp), 2 %; soft-masked blocks and two N runs as in the adversarial injection.
// Copy numbers follow from the fractions and the sequence length (a 250 Mb sequence: 83 000 Alu-like copies, 3 Gbp: a million).  Returns the number of copies written.
int64_t gsah_c_synth_human_like(char *seq, int64_t n, uint64_t seed, double scale, int64_t n_run)
{
	if (n < 200000) return 0;
	int64_t total = 0;
	struct Fam { int len; double frac; int trunc; int solo; double d_lo, d_hi; };      // trunc: 5'-truncated copies; solo: length of the solo-LTR form
	static const Fam fams[7] = {
		{ 300, 0.10, 0, 0, 0.03, 0.14 }, { 6000, 0.17, 1, 0, 0.03, 0.20 }, { 5000, 0.04, 0, 500, 0.05, 0.15 }, { 7000, 0.04, 0, 500, 0.05, 0.15 },
		{ 200, 0.02, 0, 0, 0.25, 0.30 }, { 200, 0.02, 0, 0, 0.25, 0.30 }, { 200, 0.01, 0, 0, 0.25, 0.30 } };
	for (int f = 0; f < 7; f++) {
		const Fam &F = fams[f];
		std::vector<char> fam((size_t)F.len), cp((size_t)F.len * 12 + 64);
		gsah_c_synth_genome(F.len, seed ^ (0x48554D41ull + 7919ull * (uint64_t)f), fam.data());
		const int64_t want = (int64_t)(F.frac * scale * (double)n);
		int64_t have = 0;
		for (int64_t c = 0; have < want; c++) {
			const uint64_t r = rnd(seed, 400 + (uint64_t)f, (uint64_t)c);
			int len = F.len, off = 0;
			if (F.trunc) { const double u = (double)((r >> 8) & 0xffffff) / 16777216.0; len = (int)(F.len * u * u); if (len < 100) len = 100; off = F.len - len; }
			else if (F.solo && (r & 7) < 6 && (r & 7) > 0) len = F.solo;
			double d;
			if (f == 0) { const unsigned a = (unsigned)((r >> 40) % 100); d = a < 15 ? 0.03 : (a < 65 ? 0.08 : 0.14); }
			else d = F.d_lo + (F.d_hi - F.d_lo) * (double)((r >> 40) & 0xffff) / 65536.0;
			int64_t got = mutate_into(fam.data() + off, len, d, seed + 131 * (uint64_t)f, 5000 + 2 * (uint64_t)c, cp.data(), (int64_t)cp.size());
			if (got < 0) got = 0;
			if (got > len) got = len;
			for (; got < len; got++) cp[(size_t)got] = ACGT[rnd(seed, 17 + (uint64_t)f, (uint64_t)c * 8192 + (uint64_t)got) & 3];
			const int64_t pos = (int64_t)(rnd(seed, 500 + (uint64_t)f, (uint64_t)c) % (uint64_t)(n - len));
			memcpy(seq + pos, cp.data(), (size_t)len);
			have += len; total++;
		}
	}
	{                                                                        // segmental duplications: 20-kb blocks, 1 - 3 % diverged from their source
		const int L = 20000; std::vector<char> cp((size_t)L * 2 + 64);
		for (int64_t c = 0; c < (int64_t)(0.03 * scale * (double)n / L); c++) {
			const uint64_t r = rnd(seed, 61, (uint64_t)c);
			const int64_t src = (int64_t)(r % (uint64_t)(n - L)), dst = (int64_t)(rnd(seed, 62, (uint64_t)c) % (uint64_t)(n - L));
			if (src < dst + L && dst < src + L) continue;
			int64_t got = mutate_into(seq + src, L, 0.01 + 0.02 * (double)((r >> 44) & 0xffff) / 65536.0, seed, 9000 + 2 * (uint64_t)c, cp.data(), (int64_t)cp.size());
			if (got < 0) continue;
			if (got > L) got = L;
			memcpy(seq + dst, cp.data(), (size_t)got); total++;
		}
	}
	for (int64_t k = 0; k < (int64_t)(0.02 * scale * (double)n / 160); k++) {    // microsatellites: mean length 160
		const uint64_t r = rnd(seed, 51, (uint64_t)k);
		const int unit = 1 + (int)(r % 6), len = 20 + (int)((r >> 8) % 281);
		const int64_t pos = (int64_t)((r >> 20) % (uint64_t)(n - len));
		char u[6]; for (int t = 0; t < unit; t++) u[t] = ACGT[(r >> (40 + 2 * t)) & 3];
		for (int t = 0; t < len; t++) seq[pos + t] = u[t % unit];
	}
	for (int64_t k = 0; k < n / 60000; k++) {                              // soft-masked blocks
		const uint64_t r = rnd(seed, 52, (uint64_t)k);
		const int len = 200 + (int)(r % 4801);
		const int64_t pos = (int64_t)((r >> 16) % (uint64_t)(n - len));
		for (int t = 0; t < len; t++) { const char c = seq[pos + t]; if (c >= 'A' && c <= 'Z' && c != 'N') seq[pos + t] = (char)(c | 0x20); }
	}
	if (n_run > 0 && 8 * n_run < n) for (int g = 1; g <= 2; g++) memset(seq + g * (n / 3) - n_run / 2, 'N', (size_t)n_run);
	return total;
}

// query = mutated copy of ref; returns the length written (<= cap), or -1 if cap is too small
int64_t gsah_c_synth_mutate(const char *ref, int64_t n, double d, uint64_t seed, char *out, int64_t cap)
{
	return mutate_into(ref, n, d, seed, 100, out, cap);
}

} // extern "C"
