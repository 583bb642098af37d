// rescue.cpp -- stage (4), the BWT rescue of short uncovered read regions (see rescue.h).  Follows the reference's
// src/bwt_aln.c (bwt_aln_remain :398, bwt_aln_core :306, bwt_cluster_seed :177, bwt_aln_res :200), its region bookkeeping
// (get_reg / get_remain_reg, src/lamsa_aln.c:550-637) and the FM-index queries of src/bwt.c, including their quirks.
#include "rescue.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace lamsa {

// ------------------------------------------------------------------ FM index
static bool read_all(FILE *fp, void *dst, size_t bytes) { return bytes == 0 || fread(dst, 1, bytes, fp) == bytes; }

bool FmIndex::load(const std::string &prefix, std::string &err)
{
    FILE *fp = fopen((prefix + ".bwt").c_str(), "rb");
    if (!fp) { err = "cannot open " + prefix + ".bwt"; return false; }
    fseek(fp, 0, SEEK_END);
    const long sz = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (sz < 40) { fclose(fp); err = "short .bwt"; return false; }
    const size_t words = ((size_t)sz - 40) >> 2;
    bwt.assign(words + 4, 0);
    bool ok = read_all(fp, &primary, 8) && read_all(fp, L2 + 1, 32) && read_all(fp, bwt.data(), words << 2);
    fclose(fp);
    if (!ok) { err = "short .bwt"; return false; }
    seq_len = L2[4];
    fp = fopen((prefix + ".sa").c_str(), "rb");
    if (!fp) { err = "cannot open " + prefix + ".sa"; return false; }
    uint64_t p2 = 0, skipped[4], sl = 0;
    ok = read_all(fp, &p2, 8) && read_all(fp, skipped, 32) && read_all(fp, &sa_intv, 8) && read_all(fp, &sl, 8);
    if (!ok || p2 != primary || sl != seq_len || sa_intv == 0 || (sa_intv & (sa_intv - 1))) { fclose(fp); err = ".sa does not belong to .bwt"; return false; }
    n_sa = (seq_len + sa_intv) / sa_intv;
    sa.assign((size_t)n_sa, 0);
    sa[0] = (uint64_t)-1;
    ok = read_all(fp, sa.data() + 1, 8 * (size_t)(n_sa - 1));
    fclose(fp);
    if (!ok) { err = "short .sa"; return false; }
    return true;
}

static inline int occ_aux(uint64_t y, int c)
{   // symbols equal to c among the 32 two-bit symbols of y
    y = ((c & 2) ? y : ~y) >> 1 & ((c & 1) ? y : ~y) & 0x5555555555555555ull;
    y = (y & 0x3333333333333333ull) + (y >> 2 & 0x3333333333333333ull);
    return (int)(((y + (y >> 4)) & 0xf0f0f0f0f0f0f0full) * 0x101010101010101ull >> 56);
}

uint64_t FmIndex::occ(uint64_t k, int c) const
{
    if (k == seq_len) return L2[c + 1] - L2[c];
    if (k == (uint64_t)-1) return 0;
    k -= (k >= primary);                                        // the sentinel is not stored
    const uint32_t *p = bwt.data() + ((k >> 7) << 4);           // block of 128 symbols: 4 counts (u64), then 8 words of symbols
    uint64_t n; memcpy(&n, p + 2 * c, 8);
    p += 8;
    const uint32_t *end = p + (((k >> 5) - ((k & ~127ull) >> 5)) << 1);
    for (; p < end; p += 2) n += (uint64_t)occ_aux((uint64_t)p[0] << 32 | p[1], c);
    n += (uint64_t)occ_aux(((uint64_t)p[0] << 32 | p[1]) & ~((1ull << ((~k & 31) << 1)) - 1), c);
    if (c == 0) n -= ~k & 31;                                   // the masked-out symbols read as 0
    return n;
}

uint64_t FmIndex::sa_at(uint64_t k) const
{
    uint64_t s = 0; const uint64_t mask = sa_intv - 1;
    while (k & mask) {
        ++s;
        if (k == primary) { k = 0; continue; }                  // bwt_invPsi, src/bwt.c:53-59
        uint64_t x = k - (k > primary);
        const int c = (int)(bwt[(size_t)(((x >> 7) << 4) + 8 + ((x & 0x7f) >> 4))] >> ((~x & 0xf) << 1) & 3);
        k = L2[c] + occ(k, c);
    }
    return s + sa[(size_t)(k / sa_intv)];
}

// occ(k, c) and occ(l, c) for k <= l: one walk over the block when both fall into the same 128-symbol block (bwt_2occ :132)
static inline void occ2(const FmIndex &fm, uint64_t k, uint64_t l, int c, uint64_t *ok, uint64_t *ol)
{
    const uint64_t kk = k >= fm.primary ? k - 1 : k, ll = l >= fm.primary ? l - 1 : l;
    if ((ll >> 7) != (kk >> 7) || k == (uint64_t)-1 || l == (uint64_t)-1 || l == fm.seq_len || k == fm.seq_len) { *ok = fm.occ(k, c); *ol = fm.occ(l, c); return; }
    const uint32_t *p = fm.bwt.data() + ((kk >> 7) << 4);
    uint64_t n; memcpy(&n, p + 2 * c, 8);
    p += 8;
    uint64_t i = kk & ~127ull;
    for (const uint64_t j = kk >> 5 << 5; i < j; i += 32, p += 2) n += (uint64_t)occ_aux((uint64_t)p[0] << 32 | p[1], c);
    uint64_t m = n;
    n += (uint64_t)occ_aux(((uint64_t)p[0] << 32 | p[1]) & ~((1ull << ((~kk & 31) << 1)) - 1), c);
    if (c == 0) n -= ~kk & 31;
    *ok = n;
    for (const uint64_t j = ll >> 5 << 5; i < j; i += 32, p += 2) m += (uint64_t)occ_aux((uint64_t)p[0] << 32 | p[1], c);
    m += (uint64_t)occ_aux(((uint64_t)p[0] << 32 | p[1]) & ~((1ull << ((~ll & 31) << 1)) - 1), c);
    if (c == 0) m -= ~ll & 31;
    *ol = m;
}

uint64_t FmIndex::match(int len, const uint8_t *s, uint64_t *k0, uint64_t *l0) const
{
    uint64_t k = *k0, l = *l0;
    for (int i = len - 1; i >= 0; --i) {
        const int c = s[i];
        if (c > 3) return 0;
        uint64_t ok, ol;
        occ2(*this, k - 1, l, c, &ok, &ol);
        k = L2[c] + ok + 1; l = L2[c] + ol;
        if (k > l) return 0;
    }
    *k0 = k; *l0 = l;
    return l - k + 1;
}

// ------------------------------------------------------------------ regions (get_reg / get_remain_reg)
namespace {
struct RegB { int is_rev, chr; int64_t pos; };
struct Reg { int beg, end; std::vector<RegB> rb, re; };

int ref_in_cigar(const std::vector<int32_t> &c)
{   // refInCigar, src/frag_check.c:205 (H counts as reference there)
    int n = 0;
    for (int32_t w : c) { const int op = w & 0xf; if (op == 0 || op == 2 || op == 5) n += w >> 4; }
    return n;
}

void collect_regs(const ReadResult &R, std::vector<Reg> &regs)
{
    for (int st = 0; st < 2; ++st)
        for (const Line &la : R.stage[st]) {
            if (la.tol_score < 0) continue;
            for (const Rec &r : la.rec) {
                Reg g; g.beg = r.reg_beg; g.end = r.reg_end;
                RegB b, e; b.chr = e.chr = r.chr; b.is_rev = e.is_rev = 1 - r.nstrand;
                const int64_t last = r.offset + ref_in_cigar(r.cigar) - 1;
                if (r.nstrand == 1) { b.pos = r.offset; e.pos = last; } else { e.pos = r.offset; b.pos = last; }
                g.rb.push_back(b); g.re.push_back(e);
                regs.push_back(std::move(g));
            }
        }
}

void remain_regs(std::vector<Reg> &a, int read_len, int merge_thd, int min_thd, int max_thd, std::vector<Reg> &out)
{
    if (a.empty()) {
        if (min_thd < read_len && read_len <= max_thd) { Reg g; g.beg = 1; g.end = read_len; out.push_back(g); }
        return;
    }
    std::stable_sort(a.begin(), a.end(), [](const Reg &x, const Reg &y) { return x.beg < y.beg; });      // glibc qsort on these sizes is a merge sort
    size_t cur = 0;
    for (size_t i = 1; i < a.size(); ++i) {
        if (a[i].beg - a[cur].end - 1 < merge_thd) {
            if (a[i].end > a[cur].end) a[cur].end = a[i].end;
            a[cur].rb.insert(a[cur].rb.end(), a[i].rb.begin(), a[i].rb.end());
            a[cur].re.insert(a[cur].re.end(), a[i].re.begin(), a[i].re.end());
        } else {
            ++cur;
            if (cur != i) a[cur] = a[i];
        }
    }
    a.resize(cur + 1);
    if (a[0].beg > min_thd && a[0].beg - 1 <= max_thd) { Reg g; g.beg = 1; g.end = a[0].beg - 1; g.re = a[0].rb; out.push_back(g); }
    size_t i;
    for (i = 1; i < a.size(); ++i)
        if (a[i].beg - a[i - 1].end > min_thd && a[i].beg - 1 - a[i - 1].end <= max_thd) {
            Reg g; g.beg = a[i - 1].end + 1; g.end = a[i].beg - 1; g.rb = a[i - 1].re; g.re = a[i].rb; out.push_back(g);
        }
    if (read_len - a[i - 1].end > min_thd && read_len - a[i - 1].end <= max_thd) { Reg g; g.beg = a[i - 1].end + 1; g.end = read_len; g.rb = a[i - 1].re; out.push_back(g); }
}

bool near_anchor(const Reg &reg, int ref_id, int is_rev, int64_t pos, int sv_thd)
{
    for (const RegB &b : reg.rb) if (ref_id == b.chr - 1 && is_rev == b.is_rev && abs((int)(pos - b.pos)) < sv_thd) return true;
    for (const RegB &b : reg.re) if (ref_id == b.chr - 1 && is_rev == b.is_rev && abs((int)(pos - b.pos)) < sv_thd) return true;
    return false;
}

// ------------------------------------------------------------------ seeds and their chains
struct Loc { int ref_id, is_rev; int64_t ref_pos; int from_x, from_y, score, NM, node_n, track; };
struct Seed { int pos; std::vector<Loc> loc; };
const int MAX_HIT = 100;

void set_seed(Seed &s, int ref_id, int is_rev, int64_t ref_pos)
{   // bwt_set_seed :35: a full list is emptied, the hit that found it full is dropped
    if ((int)s.loc.size() == MAX_HIT) { s.loc.clear(); return; }
    Loc l; l.ref_id = ref_id; l.is_rev = is_rev; l.ref_pos = ref_pos; l.from_x = -1; l.from_y = 0; l.score = 1; l.NM = 0; l.node_n = 1; l.track = 0;
    s.loc.push_back(l);
}

int pos2rid(const Index &ix, int64_t pos_f)
{   // bns_pos2rid, src/bntseq.c:366
    const int n = (int)ix.off.size();
    if (pos_f >= ix.l_pac) return -1;
    int left = 0, mid = 0, right = n;
    while (left < right) {
        mid = (left + right) >> 1;
        if (pos_f >= ix.off[(size_t)mid]) {
            if (mid == n - 1) break;
            if (pos_f < ix.off[(size_t)mid + 1]) break;
            left = mid + 1;
        } else right = mid;
    }
    return mid;
}

struct Heap {                        // node_score as bwt_backtrack uses it: heap_add_node, src/lamsa_dp_con.c:44; node_pop, src/lamsa_heap.c:5
    int max_n; std::vector<int> x, y, score, NM;
    void sift(int i) {
        const int n = (int)x.size();
        for (;;) {
            const int l = 2 * i + 1, r = 2 * (i + 1); int m = i;
            if (l < n && (score[l] < score[i] || (score[l] == score[i] && NM[l] > NM[i]))) m = l;
            if (r < n && (score[r] < score[m] || (score[r] == score[m] && NM[r] > NM[m]))) m = r;
            if (m == i) return;
            std::swap(x[i], x[m]); std::swap(y[i], y[m]); std::swap(score[i], score[m]); std::swap(NM[i], NM[m]);
            i = m;
        }
    }
    void add(int nx, int ny, int s, int nm) {
        const int n = (int)x.size();
        if (n < max_n) {
            x.push_back(nx); y.push_back(ny); score.push_back(s); NM.push_back(nm);
            if (n == max_n - 1) for (int i = (max_n - 1) / 2; i >= 0; --i) sift(i);
        } else if (score[0] < s || (score[0] == s && NM[0] > nm)) { score[0] = s; NM[0] = nm; x[0] = nx; y[0] = ny; sift(0); }
    }
};

}  // namespace

int RescueJobs::add(const uint8_t *q, int ql, const uint8_t *t, int tl, int kind_, int w_, int h0_)
{
    const int id = (int)qlen.size();
    q_off.push_back((int64_t)seq.size()); seq.insert(seq.end(), q, q + ql);
    t_off.push_back((int64_t)seq.size()); seq.insert(seq.end(), t, t + tl);
    qlen.push_back(ql); tlen.push_back(tl); kind.push_back(kind_); w.push_back(w_); h0.push_back(h0_);
    return id;
}

static bool fetch_ref(const Index &ix, int chr, int64_t start0, int *len, std::vector<uint8_t> &dst)
{   // pac2fa_core, src/bntseq.c:465-477 (false where the reference exits)
    const int32_t clen = ix.len[(size_t)chr - 1];
    if (start0 > clen || start0 < 0) return false;
    if (start0 + *len > clen) *len = (int)(clen - start0);
    if (*len < 0) return false;
    dst.resize((size_t)*len);
    const int64_t k0 = ix.off[(size_t)chr - 1] + start0;
    for (int i = 0; i < *len; ++i) { const int64_t k = k0 + i; dst[(size_t)i] = ix.pac[(size_t)(k >> 2)] >> ((~k & 3) << 1) & 3; }
    return true;
}

void rescue_plan(const ReadResult &R, const uint8_t *bseq, int read_len, const Index &ix, const FmIndex &fm, const lamsa_hp_para &P, RescuePlan &plan, RescueJobs &jobs)
{
    plan.lines.clear(); plan.region_of.clear();
    std::vector<Reg> regs, remain;
    collect_regs(R, regs);
    remain_regs(regs, read_len, P.bwt_seed_len, P.bwt_min_len, P.bwt_max_len, remain);
    const int seed_len = P.bwt_seed_len;
    for (size_t ri = 0; ri < remain.size(); ++ri) {
        const Reg &reg = remain[ri];
        const int reg_beg = reg.beg, reg_len = reg.end - reg_beg + 1, seed_n = reg_len - seed_len + 1;
        if (seed_n <= 0) continue;
        std::vector<Seed> sv((size_t)seed_n);
        // exact seeds (bwt_aln_core :316-363)
        for (int i = 0; i < seed_n; ++i) {
            sv[(size_t)i].pos = i + 1;
            uint64_t k = 0, l = fm.seq_len;
            if (!fm.match(seed_len, bseq + reg_beg - 1 + i, &k, &l)) continue;
            const uint64_t cnt = l - k + 1;
            if (cnt > 5 * (uint64_t)MAX_HIT) continue;
            int kept = 0;
            for (uint64_t m = k; m <= l; ++m) {
                const uint64_t rp = fm.sa_at(m);
                const int is_rev = (int64_t)rp >= ix.l_pac;
                const int64_t pos = is_rev ? (ix.l_pac << 1) - 1 - (int64_t)rp : (int64_t)rp;
                const int ref_id = pos2rid(ix, pos);
                if (ref_id < 0) continue;
                const int64_t sp = pos - ix.off[(size_t)ref_id] + 1 - (is_rev ? seed_len - 1 : 0);
                if (sp + seed_len - 1 > ix.len[(size_t)ref_id] || sp < 0) continue;
                if (cnt <= (uint64_t)MAX_HIT) set_seed(sv[(size_t)i], ref_id, is_rev, sp);
                else {                                          // many hits: only those near an anchor of the region, at most 100
                    if (!near_anchor(reg, ref_id, is_rev, sp, P.SV_len_thd)) continue;
                    set_seed(sv[(size_t)i], ref_id, is_rev, sp);
                    if (++kept == MAX_HIT) break;
                }
            }
        }
        // chain (bwt_update_dp :74): the nearest earlier seed hit within one base of the diagonal
        for (int i = 1; i < seed_n; ++i)
            for (Loc &a : sv[(size_t)i].loc) {
                bool done = false;
                for (int m = i - 1; m >= 0 && !done; --m)
                    for (size_t n = 0; n < sv[(size_t)m].loc.size(); ++n) {
                        const Loc &b = sv[(size_t)m].loc[n];
                        if (a.is_rev != b.is_rev || a.ref_id != b.ref_id) continue;
                        const int dis = (int)((a.is_rev ? -1 : 1) * (a.ref_pos - b.ref_pos) - (sv[(size_t)i].pos - sv[(size_t)m].pos));
                        if (abs(dis) > 1) continue;
                        a.from_x = m; a.from_y = (int)n; a.score = b.score + 1; a.NM = b.NM + abs(dis); a.node_n = b.node_n + 1;
                        done = true; break;
                    }
            }
        // chain ends (bwt_backtrack :97)
        Heap hp; hp.max_n = P.res_mul_max;
        int max_score = 0;
        for (int i = seed_n - 1; i >= 0; --i)
            for (size_t j = 0; j < sv[(size_t)i].loc.size(); ++j) {
                Loc &a = sv[(size_t)i].loc[j];
                if (a.track) continue;
                if (near_anchor(reg, a.ref_id, a.is_rev, a.ref_pos, P.SV_len_thd)) a.score += a.score / 2;
                if (a.score > max_score) { max_score = a.score; hp.add(i, (int)j, max_score, a.NM); }
                else if (a.score >= max_score / 2) hp.add(i, (int)j, a.score, a.NM);
                for (int fx = i, fy = (int)j; fx != -1;) { Loc &t = sv[(size_t)fx].loc[(size_t)fy]; t.track = 1; const int nx = t.from_x, ny = t.from_y; fx = nx; fy = ny; }
            }
        std::vector<std::vector<std::pair<int, int>>> lines;
        while (!hp.x.empty()) {                                 // node_pop takes the LAST array element, not the heap top
            const int rx = hp.x.back(), ry = hp.y.back();
            hp.x.pop_back(); hp.y.pop_back(); hp.score.pop_back(); hp.NM.pop_back();
            if (sv[(size_t)rx].loc[(size_t)ry].score < max_score / 2) continue;
            bool used = false;
            for (int fx = rx, fy = ry; fx != -1;) { const Loc &t = sv[(size_t)fx].loc[(size_t)fy]; if (t.track == 2) { used = true; break; } const int nx = t.from_x, ny = t.from_y; fx = nx; fy = ny; }
            if (used) continue;
            std::vector<std::pair<int, int>> ln((size_t)sv[(size_t)rx].loc[(size_t)ry].node_n);
            int i = (int)ln.size() - 1;
            for (int fx = rx, fy = ry; fx != -1; --i) { Loc &t = sv[(size_t)fx].loc[(size_t)fy]; if (i >= 0) ln[(size_t)i] = {fx, fy}; t.track = 2; const int nx = t.from_x, ny = t.from_y; fx = nx; fy = ny; }
            lines.push_back(std::move(ln));
        }
        // one alignment per chain (bwt_set_bound :183, bwt_aln_res :200-246)
        for (const auto &ln : lines) {
            RescueLine L;
            const Loc &first = sv[(size_t)ln.front().first].loc[(size_t)ln.front().second], &last = sv[(size_t)ln.back().first].loc[(size_t)ln.back().second];
            const int pos_first = sv[(size_t)ln.front().first].pos, pos_last = sv[(size_t)ln.back().first].pos;
            L.ref_id = first.ref_id; L.is_rev = first.is_rev; L.reg_beg = reg_beg; L.reg_len = reg_len;
            if (first.is_rev) {
                L.right.read_pos = reg_len + 1 - pos_first; L.left.read_pos = reg_len + 1 - (pos_last + seed_len - 1);
                L.right.ref_pos = first.ref_pos + seed_len - 1; L.left.ref_pos = last.ref_pos;
            } else {
                L.left.read_pos = pos_first; L.right.read_pos = pos_last + seed_len - 1;
                L.left.ref_pos = first.ref_pos; L.right.ref_pos = last.ref_pos + seed_len - 1;
            }
            const int ext = 100;
            const int after = read_len - (reg_beg + reg_len - 1), before = reg_beg - 1;
            if (L.is_rev) { L.left_eta = after > ext ? ext : after; L.extra_end = reg_beg + reg_len + L.left_eta - 1; L.right_eta = before > ext ? ext : before; L.extra_beg = reg_beg - L.right_eta; }
            else { L.left_eta = before > ext ? ext : before; L.extra_beg = reg_beg - L.left_eta; L.right_eta = after > ext ? ext : after; L.extra_end = reg_beg + reg_len + L.right_eta - 1; }
            const int extra_len = reg_len + L.left_eta + L.right_eta;
            L.query.resize((size_t)extra_len);
            for (int i = 0; i < extra_len; ++i) {
                const uint8_t c = bseq[L.extra_beg - 1 + i];
                if (L.is_rev) L.query[(size_t)(extra_len - 1 - i)] = c < 4 ? 3 - c : 4; else L.query[(size_t)i] = c;
            }
            const int64_t want = L.left.ref_pos - (L.left.read_pos - 1 + L.left_eta) - seed_len;
            L.ref_start = want < 1 ? 1 : want;
            L.ref_len = (int)(L.right.ref_pos + reg_len - L.right.read_pos + L.right_eta + seed_len - L.ref_start + 1);
            if (L.ref_len < 0 || !fetch_ref(ix, L.ref_id + 1, L.ref_start - 1, &L.ref_len, L.target)) continue;      // the reference exits here
            const int mq = L.right.read_pos - L.left.read_pos + 1, mt = (int)(L.right.ref_pos - L.left.ref_pos + 1);
            const int64_t t0 = L.left.ref_pos - L.ref_start;
            if (mq < 0 || mt < 0 || t0 < 0 || t0 + mt > L.ref_len) continue;
            L.job_mid = jobs.add(L.query.data() + (L.left.read_pos - 1 + L.left_eta), mq, L.target.data() + t0, mt, 0, P.band_w, 0);
            if (L.left.read_pos > 1 || L.left_eta > 0) {
                L.left_qlen = L.left.read_pos - 1 + L.left_eta;
                const int tl = (int)t0;
                L.lq.resize((size_t)L.left_qlen); L.lt.resize((size_t)tl);
                for (int i = 0; i < L.left_qlen; ++i) L.lq[(size_t)i] = L.query[(size_t)(L.left_qlen - 1 - i)];
                for (int i = 0; i < tl; ++i) L.lt[(size_t)i] = L.target[(size_t)(tl - 1 - i)];
                L.job_left = jobs.add(L.lq.data(), L.left_qlen, L.lt.data(), tl, 1, P.band_w, seed_len * P.match);
            }
            if (L.right.read_pos < reg_len || L.right_eta > 0) {
                L.right_qlen = reg_len - L.right.read_pos + L.right_eta;
                const int64_t tl = L.ref_start + L.ref_len - 1 - L.right.ref_pos;
                if (tl < 0 || tl > L.ref_len) continue;         // ksw_extend_core exits on a negative length
                L.job_right = jobs.add(L.query.data() + L.right.read_pos + L.left_eta, L.right_qlen, L.target.data() + (L.ref_len - tl), (int)tl, 1, P.band_w, seed_len * P.match);
            }
            plan.lines.push_back(std::move(L)); plan.region_of.push_back((int)ri);
        }
    }
}

// ------------------------------------------------------------------ finish
static inline void push1(std::vector<int32_t> &c, int32_t w)
{   // _push_cigar1, src/frag_check.h:153
    if ((w >> 4) == 0) return;
    if (!c.empty() && (c.back() & 0xf) == (w & 0xf)) { c.back() += (w >> 4) << 4; return; }
    c.push_back(w);
}
static inline void push_n(std::vector<int32_t> &c, const int32_t *w, int n)
{   // _push_cigar, src/frag_check.h:158: only the first word may merge (same operation, or I next to S -> S); the rest is appended as it is
    if (n == 0) return;
    int j = 0;
    if (!c.empty()) {
        const int a = c.back() & 0xf, b = w[0] & 0xf;
        if (a == b) { c.back() += (w[0] >> 4) << 4; j = 1; }
        else if ((a == 1 && b == 4) || (a == 4 && b == 1)) { c.back() = (((c.back() >> 4) + (w[0] >> 4)) << 4) | 4; j = 1; }
    }
    for (; j < n; ++j) c.push_back(w[j]);
}

void rescue_finish(ReadResult &R, const uint8_t *bseq, int read_len, const Index &ix, const lamsa_hp_para &P, RescuePlan &plan, const DpResults &dp)
{
    std::vector<Line> &out = R.stage[2];
    out.clear();
    std::vector<uint8_t> rc;
    int carry_score = 0, carry_NM = 0;                          // tol_score / tol_NM of a slot survive a rejected chain (:379-382)
    for (RescueLine &L : plan.lines) {
        auto job_cig = [&](int j, const int32_t *&w, int &n) { const int64_t g = dp.base + j; w = dp.cigar + dp.cig_off[g]; n = (int)(dp.cig_off[g + 1] - dp.cig_off[g]); };
        Line la; la.line_score = 0; la.tol_score = carry_score; la.tol_NM = carry_NM;
        Rec r; r.chr = L.ref_id + 1; r.nstrand = 1 - L.is_rev;
        push1(r.cigar, ((L.is_rev ? read_len - L.extra_end : L.extra_beg - 1) << 4) | 4);
        if (L.job_left >= 0) {
            const int32_t *w; int n; job_cig(L.job_left, w, n);
            if (n > 0) {
                const int qle = dp.qle[dp.base + L.job_left], tle = dp.tle[dp.base + L.job_left];
                L.left.read_pos -= qle; L.left.ref_pos -= tle;
                push1(r.cigar, ((L.left_qlen - qle) << 4) | 4);
                std::vector<int32_t> inv(w, w + n); std::reverse(inv.begin(), inv.end());
                push_n(r.cigar, inv.data(), n);
            } else push1(r.cigar, (L.left_qlen << 4) | 4);
        }
        r.offset = L.left.ref_pos;
        { const int32_t *w; int n; job_cig(L.job_mid, w, n); push_n(r.cigar, w, n); }
        if (L.job_right >= 0) {
            const int32_t *w; int n; job_cig(L.job_right, w, n);
            if (n > 0) { push_n(r.cigar, w, n); L.right.read_pos += dp.qle[dp.base + L.job_right]; L.right.ref_pos += dp.tle[dp.base + L.job_right]; }
            push1(r.cigar, ((L.reg_len - L.right.read_pos + L.right_eta) << 4) | 4);
        }
        push1(r.cigar, ((L.is_rev ? L.extra_beg - 1 : read_len - L.extra_end) << 4) | 4);
        // lamsa_res_aux, src/frag_check.c:793-848, on this one record
        const uint8_t *rd = bseq;
        if (L.is_rev) {
            if (rc.empty()) { rc.resize((size_t)read_len); for (int i = 0; i < read_len; ++i) rc[(size_t)i] = bseq[read_len - 1 - i] < 4 ? 3 - bseq[read_len - 1 - i] : 4; }
            rd = rc.data();
        }
        int ref_len = ref_in_cigar(r.cigar);
        std::vector<uint8_t> ref;
        bool ok = fetch_ref(ix, r.chr, r.offset - 1, &ref_len, ref);
        int ref_i = 0, read_i = 0, n_mm = 0, n_m = 0, n_io = 0, n_ie = 0, n_do = 0, n_de = 0;
        for (size_t i = 0; ok && i < r.cigar.size(); ++i) {
            const int op = r.cigar[i] & 0xf, len = r.cigar[i] >> 4;
            if (op == 0) { if (read_i + len > read_len || ref_i + len > ref_len) { ok = false; break; } int mm = 0; for (int j = 0; j < len; ++j) mm += rd[read_i++] != ref[(size_t)ref_i++]; n_m += len - mm; n_mm += mm; }
            else if (op == 1) { read_i += len; n_ie += len; ++n_io; }
            else if (op == 2) { ref_i += len; n_de += len; ++n_do; }
            else if (op == 4) read_i += len;
            else ok = false;
        }
        if (!ok || read_i != read_len || ref_i != ref_len) { R.status |= LAMSA_HP_ST_REFEXIT; return; }      // the reference exits with "Unmatched length"
        r.NM = n_mm + n_ie + n_de;
        r.score = n_m * P.match - n_mm * P.mis - n_io * P.ins_gapo - n_ie * P.ins_gape - n_do * P.del_gapo - n_de * P.del_gape;
        int cur_res_n = 0;
        if (r.score < 0) cur_res_n = -1; else { la.tol_score += r.score; la.tol_NM += r.NM; }
        if (cur_res_n < 0) la.tol_score = -1; else la.tol_score -= cur_res_n * P.split_pen;
        // covered interval of the record, push_reg_res :571
        const int32_t c0 = r.cigar.front(), c1 = r.cigar.back();
        if (r.nstrand == 1) { r.reg_beg = (c0 & 0xf) == 4 ? (c0 >> 4) + 1 : 1; r.reg_end = (c1 & 0xf) == 4 ? read_len - (c1 >> 4) : read_len; }
        else { r.reg_beg = (c1 & 0xf) == 4 ? (c1 >> 4) + 1 : 1; r.reg_end = (c0 & 0xf) == 4 ? read_len - (c0 >> 4) : read_len; }
        bool accept = true;
        if (P.read_type > 0) {                                  // solid_readInCigar(...) > 0.5 * reg_len, :376
            int solid = 0;
            for (int32_t w : r.cigar) { const int op = w & 0xf; if (op == 0 || op == 1) solid += w >> 4; }
            accept = (double)solid > 0.5 * L.reg_len;
        }
        if (accept) {
            if (cur_res_n >= 0) la.rec.push_back(std::move(r));
            out.push_back(std::move(la));
            carry_score = carry_NM = 0;                         // the next chain gets a fresh slot
        } else { carry_score = la.tol_score; carry_NM = la.tol_NM; }
    }
}

}  // namespace lamsa
