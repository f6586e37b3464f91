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
inline int code_of(char c) { switch (c) { case 'A': return 0; case 'C': return 1; case 'G': return 2; default: return 3; } }

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
		if (ev < t_sub) { o[w++] = ACGT[(code_of(src[i]) + 1 + (int)((r & 0xffff) % 3)) & 3]; i++; }
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

// query = mutated copy of ref; returns the length written (<= cap), or -1 if cap is too small
int64_t gsah_c_synth_mutate(const char *ref, int64_t n, double d, uint64_t seed, char *out, int64_t cap)
{
	return mutate_into(ref, n, d, seed, 100, out, cap);
}

} // extern "C"
