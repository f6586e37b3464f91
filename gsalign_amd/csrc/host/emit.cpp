// gsalign_amd/csrc/host/emit.cpp -- query FASTA loader and the MAF / ALN / VCF
// emitters.  CPU code by design (north_star keeps MAF/VCF emission on the host).
//
// Follows, quirk for quirk (SURVEY.md App. A.7, App. B #4,#15-#21):
//   LoadQueryFile / TrimChromosomeName / CheckQuerySeq   reference src/main.cpp:35-114
//   OutputMAF / OutputAlignment / SelfComplementarySeq   src/tools.cpp:3-44,149-286
//   VariantIdentification / OutputSequenceVariants       src/SeqVariant.cpp:6-143
#include <algorithm>
#include <cstring>
#include <fstream>
#include "gsa_host.h"

namespace {

inline int nt4(unsigned char c)
{
	switch (c | 0x20) { case 'a': return 0; case 'c': return 1; case 'g': return 2; case 't': return 3; default: return 4; }
}

// tools.cpp:3-31: ACGTN/acgtn -> upper-case complement, U/u -> A, '-' stays, everything else NUL
inline char rev_map(char c)
{
	switch (c) {
	case 'A': case 'a': return 'T'; case 'C': case 'c': return 'G'; case 'G': case 'g': return 'C'; case 'T': case 't': return 'A';
	case 'U': case 'u': return 'A'; case 'N': case 'n': return 'N'; case '-': return '-';
	default: return '\0';
	}
}

void self_complement(std::string &s, int len)          // SelfComplementarySeq (tools.cpp:33-44)
{
	int i, j;
	for (j = len - 1, i = 0; i < j; i++, j--) { char a = s[i], b = s[j]; s[i] = rev_map(b); s[j] = rev_map(a); }
	if (i == j) s[i] = rev_map(s[i]);
}

int count_gaps(const std::string &s, int i, int stop) { int n = 0; for (; i < stop; i++) if (s[i] == '-') n++; return n; }

// the two text lines of a block, built exactly like OutputMAF does (tools.cpp:168-184)
void block_text(const QueryContig &q, const ContigResult &r, const gsa_block &b, std::string &t1, std::string &t2)
{
	t1.clear(); t2.clear();
	for (int k = 0; k < b.n_frag; k++) {
		const gsa_frag &f = r.frags[b.frag_off + k];
		if (f.bseed) { t1.append(q.seq, f.qpos, f.qlen); t2.append(q.seq, f.qpos, f.qlen); }     // seeds print the QUERY text on both lines (App. B #4)
		else { t1.append(r.aln1, f.aln_off, f.aln_len); t2.append(r.aln2, f.aln_off, f.aln_len); }
	}
}

// iExtension (tools.cpp:192-202): a block running past the end of its chromosome copy is trimmed
int extension(const HostIndex &ix, const gsa_block &b, const gsa_frag &last)
{
	const int64_t end = last.rpos + last.rlen;
	const int64_t lim = (b.bdir ? ix.chr_fwd[b.chr] : ix.chr_rev[b.chr]) + ix.chr_len[b.chr];
	return end > lim ? (int)(end - lim) : 0;
}

} // namespace

void ContigResult::assign(const gsa_result &r)
{
	blocks.assign(r.blocks, r.blocks + r.n_blocks);
	frags.resize((size_t)r.n_frags);                      // FragPair_t-shaped copies of the 16-byte records
	gsa_expand_frags(r.recs, r.n_frags, frags.data());
	aln1.assign(r.aln1, (size_t)r.n_aln); aln2.assign(r.aln2, (size_t)r.n_aln);
}

bool gsah_load_query(const std::string &path, std::vector<QueryContig> &out, std::string &err)
{
	std::ifstream file(path.c_str());
	if (!file.is_open()) { err = "cannot open " + path; return false; }
	std::string str; bool first = true;
	while (std::getline(file, str)) {
		if (first) { first = false; if (str.empty() || str[0] != '>') { err = "not a FASTA file: " + path; return false; } }
		if (str.empty()) continue;
		if (str[0] == '>') {
			// TrimChromosomeName (main.cpp:35-47): '|' -> '-', cut at space # : = tab
			std::string name = str.substr(1); size_t i;
			for (i = 0; i < name.size(); i++) {
				if (name[i] == '|') name[i] = '-';
				else if (name[i] == ' ' || name[i] == '#' || name[i] == ':' || name[i] == '=' || name[i] == '\t') break;
			}
			QueryContig qc; qc.name = name.substr(0, i); out.push_back(qc);
		} else {
			if (!str.empty() && str[str.size() - 1] == '\r') str.resize(str.size() - 1);          // CheckQuerySeq (main.cpp:66-80)
			for (size_t i = 0; i < str.size(); i++) if (!isalpha((unsigned char)str[i])) { err = "The query sequence contains non-alphabet characters!"; return false; }
			if (out.empty()) { err = "sequence before the first header"; return false; }
			out.back().seq.append(str);
		}
	}
	if (out.empty()) { err = "no sequence in " + path; return false; }
	return true;
}

void Emitter::maf(FILE *fp, bool first, const QueryContig &q, ContigResult &r) const
{
	if (first) fprintf(fp, "##maf version=1\n");
	std::string t1, t2;
	for (size_t bi = 0; bi < r.blocks.size(); bi++) {
		gsa_block &b = r.blocks[bi];
		if (!allow_dup && b.bdup) continue;
		block_text(q, r, b, t1, t2);
		gsa_frag &last = r.frags[b.frag_off + b.n_frag - 1];
		const int ext = extension(*idx, b, last);
		if (ext > 0) { b.aln_len -= ext; b.score -= ext; last.rlen -= ext; last.qlen -= ext; }
		t1.resize(b.aln_len); t2.resize(b.aln_len);
		const std::string &rname = idx->chr_name[b.chr];
		std::string qname = q.name;
		if (qname.size() <= rname.size()) qname.append(rname.size() - qname.size(), ' ');           // only the query name is printed padded (App. B #18)
		const int clen = idx->chr_len[b.chr]; const unsigned qlen = (unsigned)q.seq.size();
		fprintf(fp, "a score=%d\n", b.bdup ? 1 : b.score);
		if (b.bdir) {
			fprintf(fp, "s ref.%s %d %d + %d %s\n", rname.c_str(), b.gpos - 1, b.aln_len - count_gaps(t1, 0, b.aln_len), clen, t1.c_str());
			fprintf(fp, "s qry.%s %d %d + %d %s\n\n", qname.c_str(), r.frags[b.frag_off].qpos, b.aln_len - count_gaps(t2, 0, b.aln_len), qlen, t2.c_str());
		} else {
			const int64_t rpos = last.rpos + last.rlen - 1;
			self_complement(t1, b.aln_len); self_complement(t2, b.aln_len);
			int d, c, g; idx->coordinate(rpos, &d, &c, &g);
			fprintf(fp, "s ref.%s %d %d + %d %s\n", rname.c_str(), g - 1, b.aln_len - count_gaps(t1, 0, b.aln_len), clen, t1.c_str());
			fprintf(fp, "s qry.%s %d %d - %d %s\n\n", qname.c_str(), qlen - (last.qpos + last.qlen), b.aln_len - count_gaps(t2, 0, b.aln_len), qlen, t2.c_str());
		}
	}
}

void Emitter::aln(FILE *fp, const QueryContig &q, ContigResult &r) const
{
	std::string t1, t2;
	for (size_t bi = 0; bi < r.blocks.size(); bi++) {
		gsa_block &b = r.blocks[bi];
		if (!allow_dup && b.bdup) continue;
		block_text(q, r, b, t1, t2);
		const unsigned full = (unsigned)t1.size();
		gsa_frag &last = r.frags[b.frag_off + b.n_frag - 1];
		const int ext = extension(*idx, b, last);
		if (ext > 0) { b.aln_len -= ext; b.score -= ext; last.rlen -= ext; last.qlen -= ext; t1[b.aln_len] = t2[b.aln_len] = '\0'; }
		std::string rname = idx->chr_name[b.chr], qname = q.name;
		if (qname.size() > rname.size()) rname.append(qname.size() - rname.size(), ' '); else qname.append(rname.size() - qname.size(), ' ');
		fprintf(fp, "#Identity = %d / %d (%.2f%%) Orientation = %s\n\n", b.score, b.aln_len, (int)(1000 * (1.0 * b.score / b.aln_len)) / 10.0, b.bdir ? "Forward" : "Reverse");
		unsigned pos = 0; int qp = r.frags[b.frag_off].qpos + 1; long long rp = b.gpos;
		while (pos < full) {                                         // the reference loops over the UNtrimmed length (tools.cpp:275)
			const unsigned stop = pos + 80 > full ? full : pos + 80;
			const int p = 80 - count_gaps(t1, pos, stop), qn = 80 - count_gaps(t2, pos, stop);
			fprintf(fp, "ref.%s\t%12lld\t%.80s\nqry.%s\t%12d\t%.80s\n\n", rname.c_str(), rp, t1.c_str() + pos, qname.c_str(), qp, t2.c_str() + pos);
			pos += 80; rp += (b.bdir ? p : 0 - p); qp += qn;
		}
		fprintf(fp, "%s\n", std::string(100, '*').c_str());
	}
}

bool Emitter::dotplot(const std::string &gp_path, const std::string &out_prefix, const QueryContig &q, const ContigResult &r, std::vector<std::string> *data_files) const
{
	static const char *colors[10] = { "red", "blue", "web-green", "dark-magenta", "orange", "yellow", "turquoise", "dark-yellow", "violet", "dark-grey" };
	if (r.blocks.empty()) return false;
	const int nchr = (int)idx->chr_len.size();
	std::vector<int> sum((size_t)nchr, 0);
	for (const gsa_block &b : r.blocks) if (b.score > 0) sum[(size_t)b.chr] += b.score;
	std::vector<std::pair<int, int64_t> > top;                          // reference sequences with >= 1000 identical columns, best five
	for (int i = 0; i < nchr; i++) if (sum[(size_t)i] >= 1000) top.push_back(std::make_pair(i, (int64_t)sum[(size_t)i]));
	if (top.empty()) return false;
	std::sort(top.begin(), top.end(), [](const std::pair<int, int64_t> &a, const std::pair<int, int64_t> &b) { return a.second > b.second; });
	if (top.size() > 5) top.resize(5);
	FILE *gp = fopen(gp_path.c_str(), "w"); if (!gp) return false;
	const std::string data = out_prefix + "." + q.name;
	std::vector<FILE *> fh((size_t)nchr, (FILE *)NULL);
	for (size_t i = 0; i < top.size(); i++) {
		const std::string fn = data + "vs" + idx->chr_name[(size_t)top[i].first];
		fh[(size_t)top[i].first] = fopen(fn.c_str(), "w");
		if (fh[(size_t)top[i].first]) { fprintf(fh[(size_t)top[i].first], "0 0\n0 0\n\n"); if (data_files) data_files->push_back(fn); }
	}
	fprintf(gp, "set terminal postscript color solid 'Courier' 15\nset output '%s-%s.ps'\nset grid\nset border 1\n", out_prefix.c_str(), q.name.c_str());
	for (size_t i = 0; i < top.size(); i++) fprintf(gp, "set style line %d lw 4 pt 0 ps 0.5 lc '%s'\n", (int)i + 1, colors[i]);
	fprintf(gp, "set xrange[1:*]\nset yrange[1:*]\nset xlabel 'Query (%s)'\nset ylabel 'Ref'\n", q.name.c_str());
	fprintf(gp, "plot ");
	for (size_t i = 0; i < top.size(); i++) {
		const std::string &cn = idx->chr_name[(size_t)top[i].first];
		fprintf(gp, "'%svs%s' title '%s' with lp ls %d%s", data.c_str(), cn.c_str(), cn.c_str(), (int)i + 1, i + 1 != top.size() ? ", " : "\n\n");
	}
	for (const gsa_block &b : r.blocks) {
		if (b.score <= 0 || !fh[(size_t)b.chr]) continue;
		const gsa_frag &first = r.frags[b.frag_off], &last = r.frags[b.frag_off + b.n_frag - 1];
		int d, c, g0, g1;
		idx->coordinate(first.rpos, &d, &c, &g0); idx->coordinate(last.rpos + last.rlen - 1, &d, &c, &g1);
		fprintf(fh[(size_t)b.chr], "%d %d\n%d %d\n\n", first.qpos + 1, g0, last.qpos + last.qlen, g1);
	}
	for (FILE *f : fh) if (f) fclose(f);
	fclose(gp);
	return true;
}

void Emitter::variants(int query_idx, const QueryContig &q, const ContigResult &r)
{
	const std::string &ref = idx->ref;
	for (size_t bi = 0; bi < r.blocks.size(); bi++) {
		const gsa_block &b = r.blocks[bi];
		if (b.bdup) continue;
		Variant v; v.chr_idx = b.chr; v.query_idx = query_idx;
		int d, c, g;
		for (int k = 0; k < b.n_frag; k++) {
			const gsa_frag &f = r.frags[b.frag_off + k];
			if (f.bseed) continue;
			if (f.qlen == 0 && f.rlen == 0) continue;
			if (f.qlen == 0) {                                        // pure deletion (:36-45)
				n_del++;
				v.type = 2; idx->coordinate(f.rpos - 1, &d, &c, &g); v.pos = g;
				v.ref_frag = ref.substr(f.rpos - 1, f.rlen + 1); v.alt_frag.assign(1, q.seq[f.qpos - 1]);
				vars.push_back(v);
			} else if (f.rlen == 0) {                                 // pure insertion (:46-55)
				n_ins++;
				v.type = 1; idx->coordinate(f.rpos - 1, &d, &c, &g); v.pos = g;
				v.ref_frag.assign(1, ref[f.rpos - 1]); v.alt_frag = q.seq.substr(f.qpos - 1, f.qlen + 1);
				vars.push_back(v);
			} else if (f.qlen == 1 && f.rlen == 1) {                  // 1x1 (:56-67)
				const char a1 = r.aln1[f.aln_off], a2 = r.aln2[f.aln_off];
				if (nt4(a1) != nt4(a2) && nt4(a2) != 4) {
					n_snv++;
					v.type = 0; idx->coordinate(f.rpos, &d, &c, &g); v.pos = g;
					v.ref_frag.assign(1, a1); v.alt_frag.assign(1, a2);
					vars.push_back(v);
				}
			} else {                                                  // walk the aligned columns (:68-115)
				const char *x1 = r.aln1.data() + f.aln_off, *x2 = r.aln2.data() + f.aln_off;
				const int L = f.aln_len; int64_t rp = f.rpos; int qp = f.qpos;
				for (int i = 0; i < L; i++) {
					if (x1[i] == '-') {
						n_ins++;
						int n = 1; while (i + n < L && x1[i + n] == '-') n++;
						const std::string fr = q.seq.substr(qp - 1, n + 1);
						v.type = 1; idx->coordinate(rp - 1, &d, &c, &g); v.pos = g;
						v.ref_frag.assign(1, fr[0]); v.alt_frag = fr;           // REF anchor comes from the QUERY (App. B #15)
						vars.push_back(v);
						qp += n; i += n - 1;
					} else if (x2[i] == '-') {
						n_del++;
						int n = 1; while (i + n < L && x2[i + n] == '-') n++;
						const std::string fr = ref.substr(rp - 1, n + 1);
						v.type = 2; idx->coordinate(rp - 1, &d, &c, &g); v.pos = g;
						v.ref_frag = fr; v.alt_frag.assign(1, fr[0]);
						vars.push_back(v);
						rp += n; i += n - 1;
					} else if (nt4(x1[i]) != nt4(x2[i])) {
						if (nt4(x2[i]) != 4) {
							n_snv++;
							v.type = 0; idx->coordinate(rp, &d, &c, &g); v.pos = g;
							v.ref_frag.assign(1, x1[i]); v.alt_frag.assign(1, x2[i]);
							vars.push_back(v);
						}
						rp++; qp++;
					} else { rp++; qp++; }
				}
			}
		}
	}
}

void Emitter::vcf(FILE *fp, const std::string &reference_label)
{
	static const char *MutType[3] = { "SUBSTITUTE", "INSERT", "DELETE" };
	struct ByPos { bool operator()(const Variant &a, const Variant &b) const { return a.chr_idx == b.chr_idx ? a.pos < b.pos : a.chr_idx < b.chr_idx; } };
	std::sort(vars.begin(), vars.end(), ByPos());                    // same std::sort, same incomplete key (App. B #17)
	fprintf(fp, "##fileformat=VCFv4.1\n");
	fprintf(fp, "##reference=%s\n", reference_label.c_str());
	fprintf(fp, "##source=GSAlign %s\n", "1.0.22");
	fprintf(fp, "##INFO=<ID=TYPE,Number=1,Type=String,Description=\"The type of allele, either SUBSTITUTE, INSERT, or DELETE.\">\n");
	for (size_t i = 0; i < idx->chr_name.size(); i++) fprintf(fp, "##contig=<ID=%s,length=%d>\n", idx->chr_name[i].c_str(), idx->chr_len[i]);
	fprintf(fp, "#CHROM	POS	ID	REF	ALT	QUAL	FILTER	INFO\n");
	for (size_t i = 0; i < vars.size(); i++)
		fprintf(fp, "%s\t%d\t.\t%s\t%s\t100\t*\tTYPE=%s\n", idx->chr_name[vars[i].chr_idx].c_str(), vars[i].pos, vars[i].ref_frag.c_str(), vars[i].alt_frag.c_str(), MutType[vars[i].type]);
}
