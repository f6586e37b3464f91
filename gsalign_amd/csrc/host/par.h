// gsalign_amd/csrc/host/par.h -- host-side parallelism of the CPU components (loaders, emitters): one process-wide pool of
// worker threads and an ordered writer.  The reference does all of this on one thread (LoadQueryFile main.cpp:82-114, OutputMAF
// tools.cpp:149-220, OutputSequenceVariants SeqVariant.cpp:121-143); at human scale that is ~10 GB of text behind a hot path of 0.1 s.
// Every parallel form here produces the bytes of the serial form (tests/test_host_components.py runs the goldens through both).
#ifndef GSA_HOST_PAR_H
#define GSA_HOST_PAR_H
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

// run(n, fn): fn(i) for every i in [0, n) on the pool's threads and the caller's, returns when all are done.  One run at a time per pool
// (callers serialise on `gate`); fn must not call run() on the same pool.
class HostPool {
public:
	explicit HostPool(int threads) { resize(threads); }
	~HostPool() { stop_all(); }
	int threads() const { return (int)th.size() + 1; }
	void resize(int threads)
	{
		stop_all();
		quit = false;
		for (int k = 1; k < threads; k++) th.emplace_back([this] { work(); });
	}
	void run(size_t n, const std::function<void(size_t)> &fn)
	{
		if (n == 0) return;
		if (n == 1 || th.empty()) { for (size_t i = 0; i < n; i++) fn(i); return; }
		std::lock_guard<std::mutex> one(gate);
		Job j; j.fn = &fn; j.n = n; j.left.store(n);
		{ std::lock_guard<std::mutex> g(mu); cur = &j; epoch++; }
		cv.notify_all();
		drain(j);
		std::unique_lock<std::mutex> g(mu);
		done.wait(g, [&] { return j.left.load() == 0 && j.busy == 0; });
		cur = nullptr;             // (under mu: a worker that wakes late for this epoch finds no job and goes back to sleep)
	}
	// the process-wide pool: GSA_HOST_THREADS, or -t of the CLI (set_threads), default min(hardware threads, 32)
	static HostPool &global()
	{
		static HostPool p(default_threads());
		return p;
	}
	static int default_threads()
	{
		if (const char *e = getenv("GSA_HOST_THREADS")) { const int v = atoi(e); if (v >= 1) return v > 256 ? 256 : v; }
		const unsigned hw = std::thread::hardware_concurrency();
		return hw == 0 ? 4 : (hw > 32 ? 32 : (int)hw);
	}

private:
	// One run()'s state, on run()'s stack.  A worker takes the pointer under `mu` (and counts itself in `busy` there), so it can never
	// pair an index drawn from one job with the size or function of the next: run() does not return -- and the object does not die --
	// while a worker is counted in.
	struct Job {
		const std::function<void(size_t)> *fn = nullptr; size_t n = 0;
		std::atomic<size_t> next{0}, left{0};
		int busy = 0;              // (guarded by mu)
	};
	static void drain(Job &j)
	{
		for (;;) {
			const size_t i = j.next.fetch_add(1);
			if (i >= j.n) return;
			(*j.fn)(i);
			j.left.fetch_sub(1);
		}
	}
	void work()
	{
		unsigned long long seen = 0;
		for (;;) {
			Job *j;
			{
				std::unique_lock<std::mutex> g(mu);
				cv.wait(g, [&] { return quit || epoch != seen; });
				if (quit) return;
				seen = epoch; j = cur;
				if (!j) continue;      // that run() has already finished without us
				j->busy++;
			}
			drain(*j);
			{ std::lock_guard<std::mutex> g(mu); j->busy--; }
			done.notify_all();
		}
	}
	void stop_all()
	{
		{ std::lock_guard<std::mutex> g(mu); quit = true; }
		cv.notify_all();
		for (std::thread &t : th) if (t.joinable()) t.join();
		th.clear();
	}
	std::vector<std::thread> th;
	std::mutex gate, mu; std::condition_variable cv, done;
	Job *cur = nullptr;            // (guarded by mu)
	unsigned long long epoch = 0; bool quit = false;
};

// [0, n) cut into at most `parts` contiguous ranges of at least `grain` items; fn(begin, end) per range on the global pool
template <class F> static inline void par_ranges(size_t n, size_t grain, const F &fn)
{
	HostPool &p = HostPool::global();
	size_t parts = (size_t)p.threads() * 4;
	if (grain == 0) grain = 1;
	if (parts > (n + grain - 1) / grain) parts = (n + grain - 1) / grain;
	if (parts <= 1) { if (n) fn((size_t)0, n); return; }
	p.run(parts, [&](size_t k) { const size_t b = n * k / parts, e = n * (k + 1) / parts; if (e > b) fn(b, e); });
}

// A move-only byte buffer without zero-fill (a 250 Mb block is two MAF lines of 250 MB: std::string::resize would memset them first).
struct OutBuf {
	char *p = nullptr; size_t n = 0, cap = 0;
	OutBuf() {}
	explicit OutBuf(size_t c) { reserve(c); }
	OutBuf(const OutBuf &) = delete; OutBuf &operator=(const OutBuf &) = delete;
	OutBuf(OutBuf &&o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr; o.n = o.cap = 0; }
	OutBuf &operator=(OutBuf &&o) noexcept { if (this != &o) { free(p); p = o.p; n = o.n; cap = o.cap; o.p = nullptr; o.n = o.cap = 0; } return *this; }
	~OutBuf() { free(p); }
	void reserve(size_t c) { if (c > cap) { c = c < 256 ? 256 : c; char *q = (char *)realloc(p, c); if (!q) abort(); p = q; cap = c; } }
	void append(const char *s, size_t len) { if (n + len > cap) reserve((n + len) + (n + len) / 2); memcpy(p + n, s, len); n += len; }
	void append(const std::string &s) { append(s.data(), s.size()); }
};

// Buffers written to one file descriptor in the order they were handed over, by a thread of its own: formatting the next block
// overlaps the write(2) of the previous one.  `budget` bytes may be in flight; push() blocks beyond that.  Large buffers come back
// through take(): a fresh 250 MB allocation is 60 000 page faults, a recycled one none.
class OrderedWriter {
public:
	OrderedWriter(int fd_, size_t budget_ = (size_t)3 << 30) : fd(fd_), budget(budget_) { th = std::thread([this] { loop(); }); }
	~OrderedWriter() { close(); }
	void push(OutBuf &&b)
	{
		if (b.n == 0) { recycle(std::move(b)); return; }
		std::unique_lock<std::mutex> g(mu);
		room.wait(g, [&] { return inflight == 0 || inflight + b.n <= budget; });
		inflight += b.n; q.push_back(std::move(b));
		cv.notify_one();
	}
	void push(const std::string &s) { OutBuf b(s.size()); b.append(s); push(std::move(b)); }
	// a buffer of at least `c` bytes (n = 0): a recycled one when one fits
	OutBuf take(size_t c)
	{
		{
			std::lock_guard<std::mutex> g(fmu);
			size_t best = free_.size();
			for (size_t i = 0; i < free_.size(); i++) if (free_[i].cap >= c && (best == free_.size() || free_[i].cap < free_[best].cap)) best = i;
			if (best < free_.size() && free_[best].cap <= 2 * c + ((size_t)1 << 20)) { OutBuf b = std::move(free_[best]); free_.erase(free_.begin() + (long)best); b.n = 0; return b; }
		}
		return OutBuf(c);
	}
	// waits for everything handed over so far; false after a failed write
	bool close()
	{
		if (th.joinable()) {
			{ std::lock_guard<std::mutex> g(mu); fin = true; }
			cv.notify_one(); th.join();
		}
		return ok;
	}
	double write_seconds() const { return wsec; }
	unsigned long long bytes() const { return nbytes; }

private:
	void recycle(OutBuf &&b)
	{
		if (b.cap < ((size_t)1 << 20)) return;
		std::lock_guard<std::mutex> g(fmu);
		if (free_.size() < 6) { b.n = 0; free_.push_back(std::move(b)); }
	}
	void loop()
	{
		for (;;) {
			OutBuf s;
			{
				std::unique_lock<std::mutex> g(mu);
				cv.wait(g, [&] { return fin || !q.empty(); });
				if (q.empty()) return;
				s = std::move(q.front()); q.pop_front();
			}
			const auto t0 = std::chrono::steady_clock::now();
			const char *p = s.p; size_t n = s.n;
			while (n > 0 && ok) { const ssize_t w = ::write(fd, p, n > ((size_t)1 << 30) ? ((size_t)1 << 30) : n); if (w <= 0) { ok = false; break; } p += w; n -= (size_t)w; }
			wsec += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); nbytes += s.n;
			{ std::lock_guard<std::mutex> g(mu); inflight -= s.n; }
			recycle(std::move(s));
			room.notify_all();
		}
	}
	int fd; size_t budget, inflight = 0; bool fin = false, ok = true; double wsec = 0; unsigned long long nbytes = 0;
	std::deque<OutBuf> q; std::vector<OutBuf> free_; std::mutex mu, fmu; std::condition_variable cv, room; std::thread th;
};

#endif
