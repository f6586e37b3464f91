// gsalign_amd/csrc/host/exact_sort.h -- std::sort's result, computed on many threads.
//
// Why: the reference orders VarVec with std::sort on an INCOMPLETE key -- (chromosome, position) -- so the order of variants that share a
// position is whatever libstdc++'s introsort leaves (SeqVariant.cpp:6-10,126), and "bit-identical VCF" means reproducing
// that.  A different sort (stable, parallel merge, radix) orders ties differently.  30 M variants of a human genome take std::sort ~3 s on one
// thread.
//
// How: introsort is a deterministic function of the comparator's answers.  Its quicksort phase (median of first+1 / middle / last-1 moved to the
// front, unguarded Hoare partition, recursion on the right part and iteration on the left, 2 log2(n) levels before the heapsort fallback, ranges
// of at most 16 left for the final insertion pass) splits a range into two that never interact again, so the two parts can be sorted by different
// threads; and the final insertion pass never moves an element across a partition cut (everything left of a cut is <= everything right of it and
// the insertion only passes strictly greater elements), so it can run per leaf.  exact_sort() performs exactly those partition steps -- level by
// level on the pool while the ranges are large, then one task per range -- and calls std::partial_sort for a range whose depth budget runs out,
// as introsort does.  tests/test_host_components.py compares it with std::sort itself, element for element, on inputs full of ties.
#ifndef GSA_EXACT_SORT_H
#define GSA_EXACT_SORT_H
#include <algorithm>
#include <utility>
#include <vector>
#include "par.h"

namespace exact_sort_detail {

template <class T, class C> inline void median_to_first(T *result, T *a, T *b, T *c, C comp)
{
	if (comp(*a, *b)) {
		if (comp(*b, *c)) std::iter_swap(result, b);
		else if (comp(*a, *c)) std::iter_swap(result, c);
		else std::iter_swap(result, a);
	} else if (comp(*a, *c)) std::iter_swap(result, a);
	else if (comp(*b, *c)) std::iter_swap(result, c);
	else std::iter_swap(result, b);
}

template <class T, class C> inline T *partition_pivot(T *first, T *last, C comp)
{
	T *mid = first + (last - first) / 2;
	median_to_first(first, first + 1, mid, last - 1, comp);
	T *pivot = first; T *lo = first + 1, *hi = last;
	for (;;) {
		while (comp(*lo, *pivot)) ++lo;
		--hi;
		while (comp(*pivot, *hi)) --hi;
		if (!(lo < hi)) return lo;
		std::iter_swap(lo, hi);
		++lo;
	}
}

template <class T, class C> inline void insertion(T *first, T *last, C comp)
{
	if (first == last) return;
	for (T *i = first + 1; i != last; ++i) {
		T val = std::move(*i);
		if (comp(val, *first)) { std::move_backward(first, i, i + 1); *first = std::move(val); }
		else { T *p = i, *n = i - 1; while (comp(val, *n)) { *p = std::move(*n); p = n; --n; } *p = std::move(val); }
	}
}

// the serial loop on one range with the depth budget it inherited; leaves are finished on the spot
template <class T, class C> void loop(T *first, T *last, long depth, C comp)
{
	while (last - first > 16) {
		if (depth == 0) { std::partial_sort(first, last, last, comp); return; }
		--depth;
		T *cut = partition_pivot(first, last, comp);
		loop(cut, last, depth, comp);
		last = cut;
	}
	insertion(first, last, comp);
}

inline long lg(size_t n) { long k = 0; while (n > 1) { n >>= 1; k++; } return k; }

} // namespace exact_sort_detail

// [first, last) ordered exactly as std::sort(first, last, comp) orders it
template <class T, class C> void exact_sort(T *first, T *last, C comp, size_t grain = (size_t)1 << 16)
{
	using namespace exact_sort_detail;
	const size_t n = (size_t)(last - first);
	if (n < 2) return;
	struct Range { T *b, *e; long depth; bool done; };
	std::vector<Range> cur(1, Range{ first, last, lg(n) * 2, false });
	HostPool &pool = HostPool::global();
	const size_t want = (size_t)pool.threads() * 8;
	if (pool.threads() > 1 && n > 2 * grain) {
		// breadth first: one partition step on every range that is still large, all of them at once
		for (;;) {
			std::vector<size_t> big;
			for (size_t i = 0; i < cur.size(); i++) if (!cur[i].done && (size_t)(cur[i].e - cur[i].b) > grain) big.push_back(i);
			if (big.empty() || cur.size() >= want) break;
			std::vector<Range> right(big.size());
			pool.run(big.size(), [&](size_t j) {
				Range &r = cur[big[j]];
				if (r.depth == 0) { std::partial_sort(r.b, r.e, r.e, comp); r.done = true; right[j] = Range{ r.e, r.e, 0, true }; return; }
				r.depth--;
				T *cut = partition_pivot(r.b, r.e, comp);
				right[j] = Range{ cut, r.e, r.depth, false };
				r.e = cut;
			});
			for (const Range &r : right) if (r.e > r.b) cur.push_back(r);
		}
	}
	// largest first: the pool hands ranges out in order
	std::sort(cur.begin(), cur.end(), [](const Range &a, const Range &b) { return (a.e - a.b) > (b.e - b.b); });
	pool.run(cur.size(), [&](size_t i) { if (!cur[i].done) loop(cur[i].b, cur[i].e, cur[i].depth, comp); });
}

#endif
