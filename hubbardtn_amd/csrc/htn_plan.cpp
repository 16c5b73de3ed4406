// Contraction compiler of the two-site DMRG path: sector tables + reduced MPO -> task lists for the grouped GEMM,
// SVD staging, the global truncation rule, post-SVD finalisation.  Host C++ (no HIP).
//
// This is the MI355X-first replacement for what TensorKit does with fusion trees and tree transformers around every
// contraction: instead of permuting tensors between matricisations at run time,
// every contraction is compiled ONCE per bond geometry into (output tile, segment) records; recoupling coefficients
// (closed-form 9j) become the segments' alpha.  tests/test_cplan_cpu.py holds the Python statement of the same
// planner (tests/ref_planner.py) and asserts byte-identical task lists.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <mutex>

#include "htn_core.h"

namespace htn {

// ---- error string ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

// ---- symmetry ----------------------------------------------------------------------------------------------------
bool Sym::triangle(int a, int b, int c) const {
    if (!su2()) return a + b == c;
    return abs(a - b) <= c && c <= a + b && ((a + b + c) & 1) == 0;
}
void Sym::fuse(Sec sec, int s, std::vector<Sec>& out) const {
    out.clear();
    const Sec m = site[s];
    const int N = wrapN(sec.N + m.N);
    if (!su2()) {
        out.push_back({N, sec.j + m.j});
        return;
    }
    for (int jj = abs(sec.j - m.j); jj <= sec.j + m.j; jj += 2) out.push_back({N, jj});
}
void Sym::split(Sec sec, int s, std::vector<Sec>& out) const {
    out.clear();
    const Sec m = site[s];
    if (kind != HTN_SYM_SU2 && sec.N < m.N) return;
    const int N = wrapN(sec.N - m.N);
    if (!su2()) {
        out.push_back({N, sec.j - m.j});
        return;
    }
    for (int jj = abs(sec.j - m.j); jj <= sec.j + m.j; jj += 2) out.push_back({N, jj});
}

// ---- Wigner symbols (Racah formula; validated against brute-force CG contraction, tests/test_wigner.py) ---------
static double fact(int n) {
    static double tab[171];
    static std::once_flag once;
    std::call_once(once, [] {
        tab[0] = 1.0;
        for (int i = 1; i <= 170; ++i) tab[i] = tab[i - 1] * i;
    });
    return tab[n];
}
static inline bool tri(int a, int b, int c) { return abs(a - b) <= c && c <= a + b && ((a + b + c) & 1) == 0; }
static double delta(int a, int b, int c) {
    return fact((a + b - c) / 2) * fact((a - b + c) / 2) * fact((-a + b + c) / 2) / fact((a + b + c) / 2 + 1);
}
static std::mutex g_wig_mu;
static std::unordered_map<uint64_t, double> g_6j, g_9j, g_cl, g_cr;
static inline uint64_t pack(std::initializer_list<int> v) {
    uint64_t h = 0;
    for (int x : v) h = (h << 7) | (uint64_t)(x & 127);
    return h;
}
double wigner6j(int j1, int j2, int j3, int j4, int j5, int j6) {
    if (!(tri(j1, j2, j3) && tri(j1, j5, j6) && tri(j4, j2, j6) && tri(j4, j5, j3))) return 0.0;
    const uint64_t key = pack({j1, j2, j3, j4, j5, j6});
    {
        std::lock_guard<std::mutex> lk(g_wig_mu);
        auto it = g_6j.find(key);
        if (it != g_6j.end()) return it->second;
    }
    const double pref = sqrt(delta(j1, j2, j3) * delta(j1, j5, j6) * delta(j4, j2, j6) * delta(j4, j5, j3));
    const int a1 = (j1 + j2 + j3) / 2, a2 = (j1 + j5 + j6) / 2, a3 = (j4 + j2 + j6) / 2, a4 = (j4 + j5 + j3) / 2;
    const int b1 = (j1 + j2 + j4 + j5) / 2, b2 = (j2 + j3 + j5 + j6) / 2, b3 = (j3 + j1 + j6 + j4) / 2;
    double s = 0.0;
    for (int t = std::max(std::max(a1, a2), std::max(a3, a4)); t <= std::min(b1, std::min(b2, b3)); ++t)
        s += ((t & 1) ? -1.0 : 1.0) * fact(t + 1) /
             (fact(t - a1) * fact(t - a2) * fact(t - a3) * fact(t - a4) * fact(b1 - t) * fact(b2 - t) * fact(b3 - t));
    const double r = pref * s;
    std::lock_guard<std::mutex> lk(g_wig_mu);
    g_6j[key] = r;
    return r;
}
double wigner9j(int j1, int j2, int j3, int j4, int j5, int j6, int j7, int j8, int j9) {
    if (!(tri(j1, j2, j3) && tri(j4, j5, j6) && tri(j7, j8, j9) && tri(j1, j4, j7) && tri(j2, j5, j8) && tri(j3, j6, j9)))
        return 0.0;
    const uint64_t key = pack({j1, j2, j3, j4, j5, j6, j7, j8, j9});
    {
        std::lock_guard<std::mutex> lk(g_wig_mu);
        auto it = g_9j.find(key);
        if (it != g_9j.end()) return it->second;
    }
    const int lo = std::max(std::max(abs(j1 - j9), abs(j4 - j8)), abs(j2 - j6));
    const int hi = std::min(std::min(j1 + j9, j4 + j8), j2 + j6);
    double s = 0.0;
    for (int x = lo; x <= hi; x += 2)
        s += ((x & 1) ? -1.0 : 1.0) * (x + 1) * wigner6j(j1, j4, j7, j8, j9, x) * wigner6j(j2, j5, j8, j4, x, j6) *
             wigner6j(j3, j6, j9, x, j1, j2);
    std::lock_guard<std::mutex> lk(g_wig_mu);
    g_9j[key] = s;
    return s;
}
// L'[a',w',a] += coef * A[b',s',a']^+ L[b',w,b] W[w,s',s,w'] A[b,s,a]
double coef_left(int jbp, int k, int jb, int jsp, int js, int kop, int jap, int kp, int ja) {
    return sqrt((double)((jbp + 1) * (jsp + 1) * (ja + 1) * (kp + 1))) * wigner9j(jb, k, jbp, js, kop, jsp, ja, kp, jap);
}
// R[c',w,c] += coef * conj(B[c',s',b']) W[w,s',s,w'] R[b',w',b] B[c,s,b]
double coef_right(int jcp, int k, int jc, int jsp, int js, int kop, int jbp, int kp, int jb) {
    return wigner9j(jc, k, jcp, js, kop, jsp, jb, kp, jbp) * sqrt((double)((kp + 1) * (jsp + 1))) * (jc + 1) * (jbp + 1) /
           sqrt((double)((jb + 1) * (jcp + 1)));
}
// y[a',s1',c',s2',b'] += coef * L[a',w,a] theta[a,s1,c,s2,b] R[b',w',b]^T
double coef_apply(int ja, int jap, int k, int js1, int js1p, int kop1, int km, int jc, int jcp, int js2, int js2p,
                  int kop2, int kp, int jb, int jbp) {
    return coef_left(jap, k, ja, js1p, js1, kop1, jcp, km, jc) * coef_left(jcp, km, jc, js2p, js2, kop2, jbp, kp, jb) *
           (jbp + 1) / (jb + 1);
}

// ---- bonds ---------------------------------------------------------------------------------------------------------
Bond::Bond(std::vector<std::pair<Sec, int>> items) {
    std::sort(items.begin(), items.end(), [](const std::pair<Sec, int>& a, const std::pair<Sec, int>& b) { return a.first < b.first; });
    for (auto& it : items) {
        if (it.second <= 0) continue;
        index[skey(it.first)] = (int)secs.size();
        secs.push_back(it.first);
        dims.push_back(it.second);
        int32_t rec[3] = {it.first.N, it.first.j, it.second};
        key.append((const char*)rec, sizeof(rec));
        seckey.append((const char*)rec, 2 * sizeof(int32_t));
    }
}
int64_t Bond::dim_full(const Sym& sym) const {
    int64_t d = 0;
    for (size_t i = 0; i < secs.size(); ++i) d += (int64_t)sym.qdim(secs[i]) * dims[i];
    return d;
}
int Bond::multiplets() const {
    int d = 0;
    for (int v : dims) d += v;
    return d;
}

// ---- layouts -------------------------------------------------------------------------------------------------------
SiteLayoutP build_site_layout(const Sym& sym, char kind, BondP bl, BondP br) {
    auto lay = std::make_shared<SiteLayout>();
    lay->kind = kind;
    lay->bl = bl;
    lay->br = br;
    int64_t off = 0;
    std::vector<Sec> tmp;
    auto add_block = [&](Sec l, int s, Sec r, BlockRec rec) {
        const Key k = mk(l.N, l.j, s, r.N, r.j);
        lay->bidx[k] = (int)lay->blocks.size();
        lay->bkeys.push_back(k);
        lay->blocks.push_back(rec);
    };
    if (kind == 'L') {
        for (size_t ri = 0; ri < br->secs.size(); ++ri) {
            const Sec r = br->secs[ri];
            std::vector<Grp> groups;
            for (int s = 0; s < sym.n_site; ++s) {
                sym.split(r, s, tmp);
                for (Sec l : tmp)
                    if (bl->has(l)) groups.push_back({l, s});
            }
            std::sort(groups.begin(), groups.end(), [](const Grp& a, const Grp& b) { return a.sec < b.sec || (a.sec == b.sec && a.s < b.s); });
            int rows = 0;
            for (auto& g : groups) rows += bl->dim(g.sec);
            if (rows == 0) continue;
            const int n = br->dims[ri];
            int ro = 0;
            for (auto& g : groups) {
                add_block(g.sec, g.s, r, {off + ro, bl->dim(g.sec), n, rows});
                ro += bl->dim(g.sec);
            }
            lay->midx[skey(r)] = (int)lay->mats.size();
            lay->mats.push_back({r, off, rows, n, groups});
            off += (int64_t)rows * n;
        }
    } else {
        for (size_t li = 0; li < bl->secs.size(); ++li) {
            const Sec l = bl->secs[li];
            std::vector<Grp> groups;
            for (int s = 0; s < sym.n_site; ++s) {
                sym.fuse(l, s, tmp);
                for (Sec r : tmp)
                    if (br->has(r)) groups.push_back({r, s});
            }
            std::sort(groups.begin(), groups.end(), [](const Grp& a, const Grp& b) { return a.s < b.s || (a.s == b.s && a.sec < b.sec); });
            int cols = 0;
            for (auto& g : groups) cols += br->dim(g.sec);
            if (cols == 0) continue;
            const int m = bl->dims[li];
            int co = 0;
            for (auto& g : groups) {
                add_block(l, g.s, g.sec, {off + (int64_t)co * m, m, br->dim(g.sec), m});
                co += br->dim(g.sec);
            }
            lay->midx[skey(l)] = (int)lay->mats.size();
            lay->mats.push_back({l, off, m, cols, groups});
            off += (int64_t)m * cols;
        }
    }
    lay->size = off;
    return lay;
}

ThetaLayoutP build_theta_layout(const Sym& sym, BondP bl, BondP br) {
    auto lay = std::make_shared<ThetaLayout>();
    lay->bl = bl;
    lay->br = br;
    std::map<Sec, std::vector<Grp>> rg, cg;
    std::vector<Sec> tmp;
    for (Sec a : bl->secs)
        for (int s1 = 0; s1 < sym.n_site; ++s1) {
            sym.fuse(a, s1, tmp);
            for (Sec c : tmp) rg[c].push_back({a, s1});
        }
    for (Sec b : br->secs)
        for (int s2 = 0; s2 < sym.n_site; ++s2) {
            sym.split(b, s2, tmp);
            for (Sec c : tmp) cg[c].push_back({b, s2});
        }
    int64_t off = 0;
    for (auto& kv : rg) {
        const Sec c = kv.first;
        auto ic = cg.find(c);
        if (ic == cg.end()) continue;
        ThetaLayout::Mat M;
        M.c = c;
        M.rows_g = kv.second;
        M.cols_g = ic->second;
        std::sort(M.rows_g.begin(), M.rows_g.end(), [](const Grp& a, const Grp& b) { return a.sec < b.sec || (a.sec == b.sec && a.s < b.s); });
        std::sort(M.cols_g.begin(), M.cols_g.end(), [](const Grp& a, const Grp& b) { return a.s < b.s || (a.s == b.s && a.sec < b.sec); });
        M.roffs.push_back(0);
        for (auto& g : M.rows_g) M.roffs.push_back(M.roffs.back() + bl->dim(g.sec));
        M.coffs.push_back(0);
        for (auto& g : M.cols_g) M.coffs.push_back(M.coffs.back() + br->dim(g.sec));
        M.rows = M.roffs.back();
        M.cols = M.coffs.back();
        M.off = off;
        for (size_t i = 0; i < M.rows_g.size(); ++i)
            for (size_t j = 0; j < M.cols_g.size(); ++j) {
                const Sec a = M.rows_g[i].sec, b = M.cols_g[j].sec;
                const Key k = mk(a.N, a.j, M.rows_g[i].s, c.N, c.j, M.cols_g[j].s, b.N, b.j);
                lay->bidx[k] = (int)lay->blocks.size();
                lay->bkeys.push_back(k);
                lay->blocks.push_back({off + M.roffs[i] + (int64_t)M.coffs[j] * M.rows, bl->dim(a), br->dim(b), M.rows});
            }
        off += (int64_t)M.rows * M.cols;
        lay->midx[skey(c)] = (int)lay->mats.size();
        lay->mats.push_back(std::move(M));
    }
    lay->size = off;
    return lay;
}

EnvLayoutP build_env_layout(const Sym& sym, char side, BondP bond, const std::vector<Lvl>& levels) {
    auto lay = std::make_shared<EnvLayout>();
    lay->side = side;
    lay->bond = bond;
    lay->levels = levels;
    lay->ident = side == 'L' ? 0 : (int)levels.size() - 1;
    int64_t off = 0;
    for (int w = 0; w < (int)levels.size(); ++w) {
        if (w == lay->ident) continue;
        for (size_t ki = 0; ki < bond->secs.size(); ++ki) {
            const Sec ket = bond->secs[ki];
            for (size_t bi = 0; bi < bond->secs.size(); ++bi) {
                const Sec bra = bond->secs[bi];
                if (!sym.connects(ket, levels[w].dN, levels[w].k, bra)) continue;
                Key k;
                EnvLayout::Blk rec;
                if (side == 'L') {
                    k = mk(bra.N, bra.j, w, ket.N, ket.j);
                    rec = {off, bond->dims[bi], bond->dims[ki]};
                } else {
                    k = mk(ket.N, ket.j, w, bra.N, bra.j);
                    rec = {off, bond->dims[ki], bond->dims[bi]};
                }
                lay->bidx[k] = (int)lay->blocks.size();
                lay->bkeys.push_back(k);
                lay->blocks.push_back(rec);
                lay->by_ket[mk(w, ket.N, ket.j)].push_back(bra);
                off += (int64_t)bond->dims[bi] * bond->dims[ki];
            }
        }
    }
    lay->size = off;
    return lay;
}

// ---- task-list assembly ------------------------------------------------------------------------------------------
namespace {

struct SegRec {
    int32_t type, buf_a;
    int64_t a_off;
    int32_t lda, op_a, buf_b;
    int64_t b_off;
    int32_t ldb, op_b, k;
    cplx alpha;
    bool same(const SegRec& o) const {
        return type == o.type && buf_a == o.buf_a && a_off == o.a_off && lda == o.lda && op_a == o.op_a && buf_b == o.buf_b &&
               b_off == o.b_off && ldb == o.ldb && op_b == o.op_b && k == o.k;
    }
};
struct TLBlock {
    int32_t buf;
    int64_t off;
    int32_t m, n, ld;
    std::vector<SegRec> segs;
};
struct BRow {
    int64_t off;
    int32_t buf, ld, m, n;
    int64_t seg_begin, seg_count, ncopy, ksum;
};

// segment records + block table -> Tasks: K pre-split of the GEMM segments into 16-deep slabs (tile.pad[1] = 1: the
// kernel's quads walk the slab list without reading segment records) and all tiles, longest first (LPT order)
void emit_tasks(std::vector<htn_seg>& segs, bool have_segs, std::vector<BRow>& B, int64_t pos, int64_t flops, Tasks& out) {
    if (have_segs) {
        std::vector<int64_t> cum(segs.size() + 1, 0);
        for (size_t i = 0; i < segs.size(); ++i)
            cum[i + 1] = cum[i] + (segs[i].type == HTN_SEG_GEMM ? (segs[i].k + 15) / 16 : 1);
        std::vector<htn_seg> rep((size_t)cum.back());
        for (size_t i = 0; i < segs.size(); ++i) {
            const int64_t nch = cum[i + 1] - cum[i];
            for (int64_t c = 0; c < nch; ++c) {
                htn_seg s = segs[i];
                if (s.type == HTN_SEG_GEMM) {
                    const int64_t k0 = c * 16;
                    s.a_off += s.op_a == HTN_OP_N ? k0 * s.lda : k0;
                    s.b_off += s.op_b == HTN_OP_N ? k0 : k0 * s.ldb;
                    s.k = (int32_t)std::min<int64_t>(16, s.k - k0);
                }
                rep[(size_t)(cum[i] + c)] = s;
            }
        }
        for (auto& b : B) {
            const int64_t e = cum[(size_t)(b.seg_begin + b.seg_count)], s0 = cum[(size_t)b.seg_begin];
            b.seg_count = e - s0;
            b.seg_begin = s0;
        }
        segs.swap(rep);
        pos = (int64_t)segs.size();
    }
    std::vector<htn_tile> tiles;
    std::vector<int64_t> work;
    for (auto& b : B) {
        const int ntr = (b.m + HTN_TILE - 1) / HTN_TILE, ntc = (b.n + HTN_TILE - 1) / HTN_TILE;
        for (int loc = 0; loc < ntr * ntc; ++loc) {
            const int r0 = (loc / ntc) * HTN_TILE, c0 = (loc % ntc) * HTN_TILE;
            htn_tile t;
            memset(&t, 0, sizeof(t));
            t.c_off = b.off;
            t.buf_c = b.buf;
            t.ldc = b.ld;
            t.m = std::min(HTN_TILE, b.m - r0);
            t.n = std::min(HTN_TILE, b.n - c0);
            t.row0 = r0;
            t.col0 = c0;
            t.seg_begin = (int32_t)b.seg_begin;
            t.seg_count = (int32_t)b.seg_count;
            t.pad[0] = (int32_t)b.ncopy;
            t.pad[1] = 1;
            tiles.push_back(t);
            work.push_back((int64_t)t.m * t.n * (b.ksum + 1));
        }
    }
    std::vector<int> order(tiles.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return work[a] > work[b]; });
    out.tiles.resize(tiles.size());
    for (size_t i = 0; i < order.size(); ++i) out.tiles[i] = tiles[order[i]];
    out.ntiles = (int32_t)tiles.size();
    out.segs.swap(segs);
    out.nsegs = (int32_t)pos;
    out.flops = flops;
    if (out.segs.empty()) {
        htn_seg z;
        memset(&z, 0, sizeof(z));
        out.segs.push_back(z);
    }
    if (out.tiles.empty()) {
        htn_tile z;
        memset(&z, 0, sizeof(z));
        out.tiles.push_back(z);
    }
}

class TaskList {
public:
    std::vector<TLBlock> blocks;
    KeyMap idx;
    int block(const Key& key, int buf, int64_t off, int m, int n, int ld) {
        auto it = idx.find(key);
        if (it != idx.end()) return it->second;
        const int i = (int)blocks.size();
        idx[key] = i;
        blocks.push_back({buf, off, m, n, ld, {}});
        return i;
    }
    int find(const Key& key) const {
        auto it = idx.find(key);
        return it == idx.end() ? -1 : it->second;
    }
    void add(int bi, const SegRec& r) {
        auto& v = blocks[bi].segs;
        for (auto& s : v)
            if (s.same(r)) {          // segments with identical operands are merged (alpha summed), first position kept
                s.alpha += r.alpha;
                return;
            }
        v.push_back(r);
    }
    void gemm(int bi, int buf_a, int64_t a_off, int lda, int op_a, int buf_b, int64_t b_off, int ldb, int op_b, int k, cplx alpha) {
        add(bi, {HTN_SEG_GEMM, buf_a, a_off, lda, op_a, buf_b, b_off, ldb, op_b, k, alpha});
    }
    void copy(int bi, int buf_b, int64_t b_off, int ldb, cplx alpha) {
        add(bi, {HTN_SEG_COPY, 0, 0, 1, HTN_OP_N, buf_b, b_off, ldb, HTN_OP_N, 0, alpha});
    }
    void finalize(Tasks& out) {
        std::vector<htn_seg> segs;
        std::vector<BRow> B;
        int64_t pos = 0, flops = 0;
        for (auto& b : blocks) {
            int64_t ncopy = 0, ksum = 0, cnt = 0;
            for (int pass = 0; pass < 2; ++pass)            // GEMM segments first, COPY segments last (stable)
                for (auto& s : b.segs) {
                    if (s.alpha == cplx(0.0, 0.0)) continue;
                    if ((s.type == HTN_SEG_GEMM) != (pass == 0)) continue;
                    htn_seg r;
                    memset(&r, 0, sizeof(r));
                    r.a_off = s.a_off;
                    r.b_off = s.b_off;
                    r.buf_a = s.buf_a;
                    r.buf_b = s.buf_b;
                    r.lda = s.lda;
                    r.ldb = s.ldb;
                    r.k = s.k;
                    r.op_a = s.op_a;
                    r.op_b = s.op_b;
                    r.type = s.type;
                    r.alpha_re = s.alpha.real();
                    r.alpha_im = s.alpha.imag() == 0.0 ? 0.0 : s.alpha.imag();      // never -0.0 for real coefficients
                    segs.push_back(r);
                    ++cnt;
                    if (s.type == HTN_SEG_GEMM) ksum += s.k;
                    else ++ncopy;
                }
            flops += 8 * (int64_t)b.m * b.n * ksum;
            B.push_back({b.off, b.buf, b.ld, b.m, b.n, pos, cnt, ncopy, ksum});
            pos += cnt;
        }
        if (B.empty()) {
            out = Tasks();
            htn_tile zt;
            memset(&zt, 0, sizeof(zt));
            out.tiles.push_back(zt);
            out.ntiles = 0;
            if (segs.empty()) {
                htn_seg z;
                memset(&z, 0, sizeof(z));
                segs.push_back(z);
            }
            out.segs.swap(segs);
            out.nsegs = (int32_t)pos;
            out.flops = flops;
            return;
        }
        const bool have = !segs.empty();
        emit_tasks(segs, have, B, pos, flops, out);
    }
};

inline Key tkey(Sec a, int s1, Sec c, int s2, Sec b) { return mk(a.N, a.j, s1, c.N, c.j, s2, b.N, b.j); }

}  // namespace

// ---- load balance of a launch: split-K across workgroups + placement-aware order ----------------------------------------
// Cost model of k_grouped_gemm_z (htn_gemm.hip): a tile is one 4-wave workgroup, one wave per SIMD; a tile with q = 1, 2
// or 4 quadrants splits its S slabs 4 / q ways, so it keeps each SIMD of its CU busy for ceil(S q / 4) slab units (16 MFMAs,
// ~0.6 us at the clock the chip holds under MFMA load).  A launch is bound by the most loaded CU, and where a workgroup lands
// is deterministic while every CU still has a free slot (measured, tools/gemm_prof.py): workgroup p goes to XCD p % 8 and,
// inside the XCD, to its CUs in a fixed rotation -- workgroup 256 + j lands on the CU of workgroup j.  Hence:
//   1. tiles whose load exceeds the cap (a fraction of the balanced share of a CU) are cut: first spatially into their
//      quadrants, then -- if one quadrant is still too long -- into parts along K (split-K across workgroups: the parts run
//      on different CUs and meet through the workspace, see the kernel; that hand-off costs several us);
//   2. the parts are dealt in LAYERS of n_cus: layer 0 = the n_cus longest, one per CU; every later layer gives each CU
//      one more part, longest part to the least loaded CU -- preferably a CU of the XCD where the part's output strip
//      already lives (operands then come out of that XCD's L2); GEMM_RESIDENT layers are co-resident (occupancy);
//   3. whatever does not fit those layers follows longest first: it is dispatched dynamically as slots free up.
#define GEMM_RESIDENT 6      // workgroups of k_grouped_gemm_z co-resident per CU (its launch bounds: 6 waves per SIMD)
int balance_tiles(Tasks& t, int n_cus) {
    const int nt = t.ntiles;
    if (nt <= 0) return 0;
    static const int env_cap = getenv("HTN_GEMM_CAP") ? atoi(getenv("HTN_GEMM_CAP")) : 0;
    static const int env_capk = getenv("HTN_GEMM_CAPK") ? atoi(getenv("HTN_GEMM_CAPK")) : 0;
    static const int env_xcd = getenv("HTN_GEMM_XCD") ? atoi(getenv("HTN_GEMM_XCD")) : 1;
    static const int env_layers = getenv("HTN_GEMM_LAYERS") ? atoi(getenv("HTN_GEMM_LAYERS")) : GEMM_RESIDENT;
    // Grouped (sector-run -> XCD) placement of the H_eff apply's tiles: OFF by default.  Measured at chi = 1024 (tools/
    // gemm_groups_ab.sh, PMC FETCH_SIZE pass + bench.py): fabric traffic per launch 23.5 MB -> 12.7 MB (3.3x -> 1.78x the
    // algorithmic bytes), launch time 22.1 -> 23.6 us.  The re-reads are served by the Infinity Cache and are not what binds the
    // kernel; concentrating a sector's tiles on one XCD costs more (per-XCD composition of long and short tiles) than the
    // shorter operand latency buys.  HTN_GEMM_GROUPS=1 switches it on.
    static const int env_groups = getenv("HTN_GEMM_GROUPS") ? atoi(getenv("HTN_GEMM_GROUPS")) : 0;
    const bool use_groups = env_groups && !t.group_bounds.empty();
    auto slabs = [](const htn_tile& T) { return std::max(0, T.seg_count - T.pad[0]); };
    auto ways = [](const htn_tile& T) { return 4 / ((T.m > 16 ? 2 : 1) * (T.n > 16 ? 2 : 1)); };
    auto load = [&](const htn_tile& T) { const int g = ways(T); return (slabs(T) + g - 1) / g + 1; };   // + 1: prologue / epilogue
    int64_t total = 0;
    for (int i = 0; i < nt; ++i) total += load(t.tiles[i]);
    const int64_t share = (total + n_cus - 1) / n_cus;        // balanced load of one CU
    // spatial cut: no hand-off, so a small fraction of the share -- but only while the launch has fewer workgroups than the
    // CUs can hold (4 per CU): beyond that more workgroups add prologues and operand re-reads, not parallelism
    const int slots = GEMM_RESIDENT * n_cus;                   // workgroups the chip holds at once
    int cap = nt >= slots ? (1 << 30) : std::max((int)(2 * share / 5), 4);
    int cap_k = std::max((int)share, 12);                       // K cut: the hand-off costs several us
    if (env_cap > 0) cap = env_cap;
    if (env_capk > 0) cap_k = env_capk;
    // 1a. spatial cut: a tile of 2 or 4 quadrants whose load exceeds the cap becomes 2 or 4 one-quadrant tiles sharing its
    //     segment list (same MFMA work, spread over 2 or 4 CUs; every one of them splits K four ways; nothing to reduce).
    //     (Holding the cuts back so that the launch fits the chip at once -- 1024 workgroups -- was measured slower, 26.3 vs
    //     24.4 us at chi = 1024: the tiles left whole cost more than the ~60 workgroups that start late.)
    std::vector<char> cut((size_t)nt, 0);
    {
        std::vector<int> cand;
        for (int i = 0; i < nt; ++i) {
            const htn_tile& T = t.tiles[i];
            const int g = ways(T);
            if (g < 4 && (slabs(T) + g - 1) / g > cap) cand.push_back(i);
        }
        std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return load(t.tiles[a]) > load(t.tiles[b]); });
        int64_t count = nt;
        for (int i : cand) {
            cut[i] = 1;
            count += 4 / ways(t.tiles[i]) - 1;
        }
    }
    std::vector<htn_tile> src;
    src.reserve((size_t)nt * 2);
    for (int i = 0; i < nt; ++i) {
        const htn_tile& T = t.tiles[i];
        if (!cut[i]) {
            src.push_back(T);
            continue;
        }
        for (int r0 = 0; r0 < T.m; r0 += 16)
            for (int c0 = 0; c0 < T.n; c0 += 16) {
                htn_tile Q = T;
                Q.row0 = T.row0 + r0, Q.col0 = T.col0 + c0;
                Q.m = std::min(16, T.m - r0), Q.n = std::min(16, T.n - c0);
                src.push_back(Q);
            }
    }
    std::vector<htn_tile> out;
    out.reserve(src.size() + 64);
    int ws = 0, tickets = 0;
    for (size_t i = 0; i < src.size(); ++i) {
        htn_tile T = src[i];
        T.part = 0, T.nparts = 1, T.ws_slot = 0, T.ticket = 0;
        const int S = slabs(T), g = ways(T), ld = (S + g - 1) / g;
        const int np = ld > cap_k + cap_k / 4 ? (ld + cap_k - 1) / cap_k : 1;
        if (np <= 1 || T.pad[1] == 0 || tickets >= HTN_WS_TICKET_ELEMS * 4 - 1) {
            out.push_back(T);
            continue;
        }
        const int base = ld / np, extra = ld % np;         // slab units per part: the first `extra` parts take one more
        int pos = 0;
        for (int p = 0; p < np; ++p) {
            const int u = base + (p < extra ? 1 : 0);
            const int cnt = p == np - 1 ? S - pos : std::min(g * u, S - pos);
            htn_tile P = T;
            P.seg_begin = T.seg_begin + pos;
            P.seg_count = cnt + (p == np - 1 ? T.pad[0] : 0);     // COPY segments ride with the last part
            P.pad[0] = p == np - 1 ? T.pad[0] : 0;
            P.part = p, P.nparts = np, P.ws_slot = ws, P.ticket = tickets;
            out.push_back(P);
            pos += cnt;
        }
        ws += np;
        ++tickets;
    }
    const int n = (int)out.size();
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return load(out[a]) > load(out[b]); });
    const int NX = (env_xcd && n_cus % 8 == 0) ? 8 : 1, per = n_cus / NX;       // CU (x, k) <-> launch position 8 k + x of a layer
    const int n_layers = std::min(std::max(env_layers, 1), (n + n_cus - 1) / n_cus);
    std::vector<int64_t> cu_load((size_t)n_cus, 0), x_load(NX, 0);
    std::vector<std::vector<int>> cu_tiles((size_t)n_cus);
    std::unordered_map<Key, int, KeyHash> home;
    int next = 0;
    std::vector<std::vector<int>> layers;       // tile indices of every resident layer, with the XCD each one goes to
    std::vector<std::vector<int>> layer_x;
    std::vector<char> used((size_t)n, 0);
    if (use_groups && NX > 1) {
        // GROUPED placement (H_eff apply): the coupled-sector matrices of the output are cut, in their sorted order, into NX
        // contiguous runs of about equal load and every run is given to one XCD -- a tile reads the theta matrix of its own
        // sector, those of the neighbouring sectors (the two-site terms move one electron across the centre bond) and the
        // environment blocks of its sector's outer labels, so neighbours in the sorted order share most of what they read and
        // each XCD's L2 fetches it once.  Every XCD then fills its share of each layer from its OWN longest-first list; what
        // an XCD cannot place (its list ran out) is filled from the others' leftovers at the end.
        const int ng = (int)t.group_bounds.size() + 1;
        std::vector<int64_t> gload((size_t)ng, 0);
        std::vector<int> grp((size_t)n);
        int64_t tot = 0;
        for (int i = 0; i < n; ++i) {
            grp[i] = (int)(std::upper_bound(t.group_bounds.begin(), t.group_bounds.end(), out[i].c_off) - t.group_bounds.begin());
            gload[grp[i]] += load(out[i]);
            tot += load(out[i]);
        }
        // all tiles in the order (group, block, row strip, column): cut into NX runs of equal load (a large sector is shared
        // by two neighbouring XCDs rather than unbalancing one)
        std::vector<int> seq(n);
        for (int i = 0; i < n; ++i) seq[i] = i;
        std::stable_sort(seq.begin(), seq.end(), [&](int a, int b) {
            if (grp[a] != grp[b]) return grp[a] < grp[b];
            if (out[a].c_off != out[b].c_off) return out[a].c_off < out[b].c_off;
            if (out[a].row0 != out[b].row0) return out[a].row0 < out[b].row0;
            return out[a].col0 < out[b].col0;
        });
        std::vector<int> thome((size_t)n, 0);
        {
            // weight of a tile in the XCD partition: its slab units plus a fixed cost (start-up, reduction, store) that
            // the per-CU balance above can afford to underestimate (it averages out over a CU's six tiles) but a partition
            // by sector cannot: the outer sectors are all small tiles
            static const int pw = getenv("HTN_GEMM_PW") ? atoi(getenv("HTN_GEMM_PW")) : 3;
            auto wgt = [&](int i) { return (int64_t)load(out[i]) - 1 + pw; };
            int64_t wtot = 0, acc = 0;
            for (int i = 0; i < n; ++i) wtot += wgt(i);
            for (int q = 0; q < n; ++q) {
                const int i = seq[q];
                const int64_t mid = acc + wgt(i) / 2;
                thome[i] = (int)std::min<int64_t>(NX - 1, mid * NX / std::max<int64_t>(wtot, 1));
                acc += wgt(i);
            }
        }
        (void)gload;
        std::vector<std::vector<int>> xq(NX);       // per XCD, longest first (order[] is sorted)
        for (int j = 0; j < n; ++j) xq[thome[order[j]]].push_back(order[j]);
        std::vector<size_t> xpos(NX, 0);
        for (int l = 0; l < n_layers; ++l) {
            const int cnt = std::min(n_cus, n - next);
            std::vector<int> room(NX, per);
            if (cnt < n_cus)
                for (int x = 0; x < NX; ++x) room[x] = cnt / NX + (x < cnt % NX ? 1 : 0);
            std::vector<int> li, lx;
            for (int x = 0; x < NX; ++x)
                while (room[x] > 0 && xpos[x] < xq[x].size()) {
                    li.push_back(xq[x][xpos[x]++]);
                    lx.push_back(x);
                    --room[x];
                }
            for (int x = 0; x < NX; ++x)                     // slots an XCD could not fill itself: the longest leftover of the fullest queue
                while (room[x] > 0) {
                    int y = -1;
                    for (int c = 0; c < NX; ++c)
                        if (xpos[c] < xq[c].size() && (y < 0 || xq[c].size() - xpos[c] > xq[y].size() - xpos[y])) y = c;
                    if (y < 0) break;
                    li.push_back(xq[y][xpos[y]++]);
                    lx.push_back(x);
                    --room[x];
                }
            for (int i : li) used[i] = 1;
            next += (int)li.size();
            layers.push_back(li);
            layer_x.push_back(lx);
        }
    } else {
        for (int l = 0; l < n_layers; ++l) {
            const int cnt = std::min(n_cus, n - next);
            // the XCD of every part of this layer: its strip's home if that XCD still has room in the layer
            std::vector<int> room(NX, per), xs(cnt), li(cnt);
            if (cnt < n_cus)            // last, partial layer: launch positions are filled in order, XCD x gets ceil / floor
                for (int x = 0; x < NX; ++x) room[x] = cnt / NX + (x < cnt % NX ? 1 : 0);
            for (int j = 0; j < cnt; ++j) {
                const htn_tile& T = out[order[next + j]];
                const Key k = mk(T.buf_c, (int32_t)(T.c_off & 0x7fffffff), (int32_t)(T.c_off >> 31), T.row0);
                auto it = home.find(k);
                int x = it == home.end() ? -1 : it->second;
                if (x < 0 || room[x] == 0) {
                    int y = -1;
                    for (int c = 0; c < NX; ++c)
                        if (room[c] > 0 && (y < 0 || x_load[c] < x_load[y])) y = c;
                    if (x < 0) home[k] = y;
                    x = y;
                }
                --room[x];
                xs[j] = x;
                li[j] = order[next + j];
                used[li[j]] = 1;
                x_load[x] += load(T);
            }
            next += cnt;
            layers.push_back(li);
            layer_x.push_back(xs);
        }
    }
    for (size_t l = 0; l < layers.size(); ++l) {
        // inside the XCD: longest part of the layer to the least loaded CU that has no part of this layer yet
        std::vector<int> idx(layers[l].size());
        for (size_t j = 0; j < idx.size(); ++j) idx[j] = (int)j;
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return load(out[layers[l][a]]) > load(out[layers[l][b]]); });
        std::vector<char> taken((size_t)n_cus, 0);
        for (int j : idx) {
            const int i = layers[l][j], x = layer_x[l][j];
            int best = -1;
            for (int k = 0; k < per; ++k) {
                const int c = k * NX + x;
                if (!taken[c] && (best < 0 || cu_load[c] < cu_load[best])) best = c;
            }
            taken[best] = 1;
            cu_tiles[best].push_back(i);
            cu_load[best] += load(out[i]);
        }
    }
    // launch order: CUs that take part in the last (possibly partial) layer come first inside their XCD -- the rotation of
    // an XCD restarts at its first CU with every layer
    std::vector<std::vector<int>> cus(NX);
    for (int x = 0; x < NX; ++x) {
        for (int k = 0; k < per; ++k) cus[x].push_back(k * NX + x);
        std::stable_sort(cus[x].begin(), cus[x].end(),
                         [&](int a, int b) { return cu_tiles[a].size() > cu_tiles[b].size(); });
    }
    std::vector<htn_tile> fin;
    fin.reserve((size_t)n);
    for (int l = 0; l < n_layers; ++l)
        for (int k = 0; k < per; ++k)
            for (int x = 0; x < NX; ++x) {
                const int c = cus[x][k];
                if ((int)cu_tiles[c].size() > l) fin.push_back(out[cu_tiles[c][l]]);
            }
    for (int j = 0; j < n; ++j)
        if (!used[order[j]]) fin.push_back(out[order[j]]);
    t.tiles.swap(fin);
    t.ntiles = (int32_t)t.tiles.size();
    return ws;
}

// ---- y = H_eff x ----------------------------------------------------------------------------------
void plan_apply(const Mpo& mpo, const ThetaLayout& tl, const EnvLayout& Ll, const EnvLayout& Rl, const MpoSite& W1,
                const MpoSite& W2, ApplyPlan& out) {
    const Sym& sym = mpo.sym;
    const int nfin = (int)W2.right.size() - 1;
    std::unordered_map<int, std::vector<const MpoEntry*>> w2;
    for (auto& e : W2.entries) w2[e.wl].push_back(&e);
    TaskList ty, tz;
    for (size_t i = 0; i < tl.blocks.size(); ++i) {
        const BlockRec& r = tl.blocks[i];
        ty.block(tl.bkeys[i], BUF_Y, r.off, r.m, r.n, r.ld);
    }
    int64_t zoff = 0;
    KeyMap zblocks;
    int64_t nterms = 0;
    std::vector<Sec> cps;
    const std::vector<Sec> none;
    for (size_t xi = 0; xi < tl.blocks.size(); ++xi) {
        const Key& beta = tl.bkeys[xi];
        const BlockRec& X = tl.blocks[xi];
        const Sec a{beta[0], beta[1]}, c{beta[3], beta[4]}, b{beta[6], beta[7]};
        const int s1 = beta[2], s2 = beta[5];
        const int js1 = sym.jeff(sym.site[s1].j), js2 = sym.jeff(sym.site[s2].j);
        for (auto& e1 : W1.entries) {
            const int w = e1.wl, wm = e1.wr;
            const SiteOp& o1 = mpo.ops[e1.op];
            const int kw = W1.left[w].k, kmid = W1.right[wm].k;
            std::vector<Sec> one_a{a};
            const std::vector<Sec>* aps = w == 0 ? &one_a : Ll.kets(w, a);
            if (!aps || aps->empty()) continue;
            for (int s1p = 0; s1p < sym.n_site; ++s1p) {
                const double r1 = o1.red[s1p][s1];
                if (r1 == 0.0) continue;
                auto iw = w2.find(wm);
                if (iw == w2.end()) continue;
                for (const MpoEntry* e2 : iw->second) {
                    const int wp = e2->wr;
                    const SiteOp& o2 = mpo.ops[e2->op];
                    const int kwp = W2.right[wp].k;
                    std::vector<Sec> one_b{b};
                    const std::vector<Sec>* bps = wp == nfin ? &one_b : Rl.kets(wp, b);
                    if (!bps || bps->empty()) continue;
                    for (int s2p = 0; s2p < sym.n_site; ++s2p) {
                        const double r2 = o2.red[s2p][s2];
                        if (r2 == 0.0) continue;
                        for (Sec ap : *aps) {
                            sym.fuse(ap, s1p, cps);
                            for (Sec cp : cps)
                                for (Sec bp : *bps) {
                                    const Key betap = tkey(ap, s1p, cp, s2p, bp);
                                    const int oi = ty.find(betap);
                                    if (oi < 0) continue;
                                    const double cf = coef_apply(sym.jeff(a.j), sym.jeff(ap.j), sym.jeff(kw), js1,
                                                                 sym.jeff(sym.site[s1p].j), sym.jeff(o1.k), sym.jeff(kmid),
                                                                 sym.jeff(c.j), sym.jeff(cp.j), js2, sym.jeff(sym.site[s2p].j),
                                                                 sym.jeff(o2.k), sym.jeff(kwp), sym.jeff(b.j), sym.jeff(bp.j));
                                    const cplx alpha = cf * r1 * r2 * e1.coef * e2->coef;
                                    if (alpha == cplx(0.0, 0.0)) continue;
                                    ++nterms;
                                    const bool hasL = w != 0, hasR = wp != nfin;
                                    if (!hasL && !hasR) {
                                        ty.copy(oi, BUF_X, X.off, X.ld, alpha);
                                    } else if (hasL && !hasR) {
                                        const auto& Lb = Ll.blocks[Ll.block(ap, w, a)];
                                        ty.gemm(oi, BUF_L, Lb.off, Lb.m, HTN_OP_N, BUF_X, X.off, X.ld, HTN_OP_N, Lb.n, alpha);
                                    } else if (hasR && !hasL) {
                                        const auto& Rb = Rl.blocks[Rl.block(b, wp, bp)];
                                        ty.gemm(oi, BUF_X, X.off, X.ld, HTN_OP_N, BUF_R, Rb.off, Rb.m, HTN_OP_N, Rb.m, alpha);
                                    } else {
                                        const auto& Lb = Ll.blocks[Ll.block(ap, w, a)];
                                        const auto& Rb = Rl.blocks[Rl.block(b, wp, bp)];
                                        const Key zk = mk(oi, wp, b.N, b.j);
                                        int zi;
                                        auto iz = zblocks.find(zk);
                                        if (iz == zblocks.end()) {
                                            zi = tz.block(zk, BUF_Z, zoff, Lb.m, X.n, Lb.m);
                                            zblocks[zk] = zi;
                                            ty.gemm(oi, BUF_Z, zoff, Lb.m, HTN_OP_N, BUF_R, Rb.off, Rb.m, HTN_OP_N, Rb.m, 1.0);
                                            zoff += (int64_t)Lb.m * X.n;
                                        } else
                                            zi = iz->second;
                                        tz.gemm(zi, BUF_L, Lb.off, Lb.m, HTN_OP_N, BUF_X, X.off, X.ld, HTN_OP_N, Lb.n, alpha);
                                    }
                                }
                        }
                    }
                }
            }
        }
    }
    out.has_z = !zblocks.empty();
    if (out.has_z) tz.finalize(out.tz);
    ty.finalize(out.ty);
    for (const auto& M : tl.mats) out.ty.group_bounds.push_back(M.off);      // placement hint: one group per coupled-sector matrix of y
    out.zsize = zoff;
    out.nterms = nterms;
}

// ---- theta = T1 . T2 ------------------------------------------------------------------------------------------------
// mode "RR": both right layout (centre on i); "LL": both left layout (centre on i+1); "LR": left x right layout.
// buffers: BUF_S1 = site i, BUF_S2 = site i+1, output BUF_Y.
void plan_theta(const char* mode, const SiteLayout& lay1, const SiteLayout& lay2, const ThetaLayout& tl, Tasks& out) {
    TaskList t;
    const bool LR = !strcmp(mode, "LR"), RR = !strcmp(mode, "RR");
    for (auto& M : tl.mats) {
        const Sec c = M.c;
        if (LR) {
            const int bi = t.block(mk(0, c.N, c.j), BUF_Y, M.off, M.rows, M.cols, M.rows);
            const int m1 = lay1.mat(c), m2 = lay2.mat(c);
            if (m1 >= 0 && m2 >= 0) {
                const auto& A = lay1.mats[m1];
                const auto& Bm = lay2.mats[m2];
                t.gemm(bi, BUF_S1, A.off, A.rows, HTN_OP_N, BUF_S2, Bm.off, Bm.rows, HTN_OP_N, A.cols, 1.0);
            }
        } else if (RR) {
            const int m2 = lay2.mat(c);
            for (size_t i = 0; i < M.rows_g.size(); ++i) {
                const Sec a = M.rows_g[i].sec;
                const int s1 = M.rows_g[i].s;
                const int bi = t.block(mk(1, a.N, a.j, s1, c.N, c.j), BUF_Y, M.off + M.roffs[i], tl.bl->dim(a), M.cols, M.rows);
                const int blk = lay1.block(a, s1, c);
                if (blk >= 0 && m2 >= 0) {
                    const BlockRec& r = lay1.blocks[blk];
                    const auto& Bm = lay2.mats[m2];
                    t.gemm(bi, BUF_S1, r.off, r.ld, HTN_OP_N, BUF_S2, Bm.off, Bm.rows, HTN_OP_N, r.n, 1.0);
                }
            }
        } else {
            const int m1 = lay1.mat(c);
            for (size_t j = 0; j < M.cols_g.size(); ++j) {
                const Sec b = M.cols_g[j].sec;
                const int s2 = M.cols_g[j].s;
                const int bi = t.block(mk(2, c.N, c.j, s2, b.N, b.j), BUF_Y, M.off + (int64_t)M.coffs[j] * M.rows, M.rows,
                                       tl.br->dim(b), M.rows);
                const int blk = lay2.block(c, s2, b);
                if (blk >= 0 && m1 >= 0) {
                    const BlockRec& r = lay2.blocks[blk];
                    const auto& A = lay1.mats[m1];
                    t.gemm(bi, BUF_S1, A.off, A.rows, HTN_OP_N, BUF_S2, r.off, r.ld, HTN_OP_N, r.m, 1.0);
                }
            }
        }
    }
    t.finalize(out);
}

// ---- environments (a10) -------------------------------------------------------------------------------------------
// GL[i+1] from GL[i], left-layout site tensor (BUF_S1), MPO site W.  stage 1 -> BUF_Z (Y panels), stage 2 -> BUF_Y.
void plan_left_env(const Mpo& mpo, const EnvLayout& Ll, const SiteLayout& lay, const MpoSite& W, const EnvLayout& Lnew,
                   EnvPlan& out) {
    const Sym& sym = mpo.sym;
    TaskList t1, t2;
    int64_t zoff = 0;
    KeyMap ypanel;
    for (size_t q = 0; q < Lnew.blocks.size(); ++q) {
        const Key& k = Lnew.bkeys[q];
        const Sec cp{k[0], k[1]}, c{k[3], k[4]};
        const int wp = k[2];
        const auto& nb = Lnew.blocks[q];
        const int b2 = t2.block(k, BUF_Y, nb.off, nb.m, nb.n, nb.m);
        const int mcp = lay.mat(cp), mc = lay.mat(c);
        if (mcp < 0 || mc < 0) continue;                       // structurally zero block
        const auto& Mcp = lay.mats[mcp];
        ypanel[k] = 1;
        t2.gemm(b2, BUF_S1, Mcp.off, Mcp.rows, HTN_OP_C, BUF_Z, zoff, Mcp.rows, HTN_OP_N, Mcp.rows, 1.0);
        int ro = 0;
        for (auto& g : Mcp.groups) {
            t1.block(mk(cp.N, cp.j, wp, c.N, c.j, g.sec.N, g.sec.j, g.s), BUF_Z, zoff + ro, lay.bl->dim(g.sec), nb.n, Mcp.rows);
            ro += lay.bl->dim(g.sec);
        }
        zoff += (int64_t)Mcp.rows * nb.n;
    }
    std::vector<Sec> cps;
    for (auto& e : W.entries) {
        const int wl = e.wl, wr = e.wr;
        if (wr == 0 && W.right.size() > 1) continue;           // 'start' level stays the implicit identity
        const SiteOp& o = mpo.ops[e.op];
        const int kl = W.left[wl].k, kr = W.right[wr].k;
        for (size_t bi = 0; bi < lay.blocks.size(); ++bi) {
            const Key& bk = lay.bkeys[bi];
            const Sec a{bk[0], bk[1]}, c{bk[3], bk[4]};
            const int s = bk[2];
            const BlockRec& Ab = lay.blocks[bi];
            for (int sp = 0; sp < sym.n_site; ++sp) {
                const double r = o.red[sp][s];
                if (r == 0.0) continue;
                std::vector<Sec> one{a};
                const std::vector<Sec>* aps = wl == 0 ? &one : Ll.kets(wl, a);
                if (!aps) continue;
                for (Sec ap : *aps) {
                    sym.fuse(ap, sp, cps);
                    for (Sec cp : cps) {
                        if (!ypanel.count(mk(cp.N, cp.j, wr, c.N, c.j)) || lay.block(ap, sp, cp) < 0) continue;
                        const double cf = coef_left(sym.jeff(ap.j), sym.jeff(kl), sym.jeff(a.j), sym.jeff(sym.site[sp].j),
                                                    sym.jeff(sym.site[s].j), sym.jeff(o.k), sym.jeff(cp.j), sym.jeff(kr),
                                                    sym.jeff(c.j));
                        const cplx alpha = cf * r * e.coef;
                        if (alpha == cplx(0.0, 0.0)) continue;
                        const int b1 = t1.find(mk(cp.N, cp.j, wr, c.N, c.j, ap.N, ap.j, sp));
                        if (wl == 0)
                            t1.copy(b1, BUF_S1, Ab.off, Ab.ld, alpha);
                        else {
                            const auto& Lb = Ll.blocks[Ll.block(ap, wl, a)];
                            t1.gemm(b1, BUF_L, Lb.off, Lb.m, HTN_OP_N, BUF_S1, Ab.off, Ab.ld, HTN_OP_N, Lb.n, alpha);
                        }
                    }
                }
            }
        }
    }
    t1.finalize(out.t1);
    t2.finalize(out.t2);
    out.zsize = zoff;
}

// GR[i] (stored transposed: block (c, w, c') = [n_c, n_c']) from GR[i+1], right-layout site tensor (BUF_S1), MPO site W
void plan_right_env(const Mpo& mpo, const EnvLayout& Rl, const SiteLayout& lay, const MpoSite& W, const EnvLayout& Rnew,
                    EnvPlan& out) {
    const Sym& sym = mpo.sym;
    TaskList t1, t2;
    const int nfin_r = (int)W.right.size() - 1, nfin_l = (int)W.left.size() - 1;
    int64_t zoff = 0;
    KeyMap ypanel;
    for (size_t q = 0; q < Rnew.blocks.size(); ++q) {
        const Key& k = Rnew.bkeys[q];
        const Sec c{k[0], k[1]}, cp{k[3], k[4]};
        const int w = k[2];
        const auto& nb = Rnew.blocks[q];
        const int b2 = t2.block(k, BUF_Y, nb.off, nb.m, nb.n, nb.m);
        const int mcp = lay.mat(cp), mc = lay.mat(c);
        if (mcp < 0 || mc < 0) continue;
        const auto& Mcp = lay.mats[mcp];
        ypanel[k] = 1;
        // Rt[c, w, c'] (n_c x n_c') = Y (n_c x cols') . B_{c'}^H (cols' x n_c')
        t2.gemm(b2, BUF_Z, zoff, nb.m, HTN_OP_N, BUF_S1, Mcp.off, Mcp.rows, HTN_OP_C, Mcp.cols, 1.0);
        int co = 0;
        for (auto& g : Mcp.groups) {
            t1.block(mk(c.N, c.j, w, cp.N, cp.j, g.s, g.sec.N, g.sec.j), BUF_Z, zoff + (int64_t)co * nb.m, nb.m, lay.br->dim(g.sec), nb.m);
            co += lay.br->dim(g.sec);
        }
        zoff += (int64_t)nb.m * Mcp.cols;
    }
    std::vector<Sec> cps;
    for (auto& e : W.entries) {
        const int wl = e.wl, wr = e.wr;
        if (wl == nfin_l && W.left.size() > 1) continue;       // 'final' level stays the implicit identity
        const SiteOp& o = mpo.ops[e.op];
        const int kl = W.left[wl].k, kr = W.right[wr].k;
        for (size_t bi = 0; bi < lay.blocks.size(); ++bi) {
            const Key& bk = lay.bkeys[bi];
            const Sec c{bk[0], bk[1]}, b{bk[3], bk[4]};
            const int s = bk[2];
            const BlockRec& Bb = lay.blocks[bi];
            for (int sp = 0; sp < sym.n_site; ++sp) {
                const double r = o.red[sp][s];
                if (r == 0.0) continue;
                std::vector<Sec> one{b};
                const std::vector<Sec>* bps = wr == nfin_r ? &one : Rl.kets(wr, b);
                if (!bps) continue;
                for (Sec bp : *bps) {
                    sym.split(bp, sp, cps);
                    for (Sec cp : cps) {
                        if (!ypanel.count(mk(c.N, c.j, wl, cp.N, cp.j)) || lay.block(cp, sp, bp) < 0) continue;
                        const double cf = coef_right(sym.jeff(cp.j), sym.jeff(kl), sym.jeff(c.j), sym.jeff(sym.site[sp].j),
                                                     sym.jeff(sym.site[s].j), sym.jeff(o.k), sym.jeff(bp.j), sym.jeff(kr),
                                                     sym.jeff(b.j));
                        const cplx alpha = cf * r * e.coef;
                        if (alpha == cplx(0.0, 0.0)) continue;
                        const int b1 = t1.find(mk(c.N, c.j, wl, cp.N, cp.j, sp, bp.N, bp.j));
                        if (wr == nfin_r)
                            t1.copy(b1, BUF_S1, Bb.off, Bb.ld, alpha);
                        else {
                            const auto& Rb = Rl.blocks[Rl.block(b, wr, bp)];
                            t1.gemm(b1, BUF_S1, Bb.off, Bb.ld, HTN_OP_N, BUF_R, Rb.off, Rb.m, HTN_OP_N, Rb.m, alpha);
                        }
                    }
                }
            }
        }
    }
    t1.finalize(out.t1);
    t2.finalize(out.t2);
    out.zsize = zoff;
}

// ---- SVD staging ---------------------------------------------------------------------------------------------------
// Default (QRCP): stage G0 = M^H ('right') or M ('left'); the kernel preconditions it by pivoted QR and runs Jacobi on
// R^H, returning (isometry x Sigma) directly.  Blocks with a side > 512 fall back to the plain staging: mode A stages
// the block so that the normalised Jacobi output IS the wanted isometry; mode B (block much wider than tall in that
// orientation) orthogonalises the short side instead and accumulates the rotation J, which then is the isometry.
int plan_svd(const ThetaLayout& tl, bool right, SvdPlan& out) {
    const size_t n = tl.mats.size();
    out = SvdPlan();
    out.desc.resize(std::max<size_t>(n, 1));
    out.stage.resize(std::max<size_t>(n, 1));
    memset(out.desc.data(), 0, sizeof(htn_svd_block) * out.desc.size());
    memset(out.stage.data(), 0, sizeof(htn_copy_item) * out.stage.size());
    int64_t go = 0, vo = 0, so = 0;
    for (size_t i = 0; i < n; ++i) {
        const auto& M = tl.mats[i];
        out.mids.push_back(M.c);
        const int rows = M.rows, cols = M.cols;
        const int mA = right ? rows : cols, nA = right ? cols : rows;
        htn_svd_block& d = out.desc[i];
        htn_copy_item& st = out.stage[i];
        st.src_off = M.off;
        st.idx_off = -1;
        st.scl_off = -1;
        st.lds = rows;
        st.gather_dim = 0;
        st.scale_dim = -1;
        st.inv_norm = 0;
        const int64_t mm = std::max(rows, cols), kk = std::min(rows, cols);
        out.flops += 4 * (4 * mm * kk * kk + 8 * kk * kk * kk);
        if (std::max(rows, cols) <= 512) {
            const int m0 = nA, n0 = mA;                       // G0 is m0 x n0; Jacobi works on R^H: n0 x r
            const int r = std::min(m0, n0);
            d.g_off = go;
            d.v_off = vo;
            d.s_off = so;
            d.m = n0;
            d.n = r;
            d.flags = HTN_SVD_QRCP;
            d.pad = m0;
            st.dst_off = go;
            st.rows = m0;
            st.cols = n0;
            st.ldd = m0;
            st.op = right ? HTN_OP_C : HTN_OP_N;
            out.transposed.push_back(!right);
            out.accumulate.push_back(0);
            go += (int64_t)m0 * n0;
            vo += (int64_t)((n0 + 63) / 64 * 64) * r;         // padded leading dimension of the kernel's R^H workspace
            so += r;
            out.max_m = std::max(out.max_m, std::max(m0, n0));
            continue;
        }
        const bool modeA = (nA <= 1.25 * mA && mA <= 512) || nA > 512;
        int m, nn;
        bool tr, acc;
        if (modeA) {
            m = mA, nn = nA, tr = !right, acc = false;
        } else {
            m = nA, nn = mA, tr = right, acc = true;
        }
        if (m > 512) return set_error("coupled block taller than 512 rows in both orientations (%d x %d)", rows, cols);
        d.g_off = go;
        d.v_off = vo;
        d.s_off = so;
        d.m = m;
        d.n = nn;
        d.flags = acc ? HTN_SVD_ACCUMULATE : 0;
        d.pad = 0;
        st.dst_off = go;
        st.rows = m;
        st.cols = nn;
        st.ldd = m;
        st.op = tr ? HTN_OP_C : HTN_OP_N;
        out.transposed.push_back(tr);
        out.accumulate.push_back(acc);
        out.any_accumulate |= acc;
        go += (int64_t)m * nn;
        vo += acc ? (int64_t)nn * nn : 0;
        so += nn;
        out.max_m = std::max(out.max_m, m);
    }
    out.g_size = go;
    out.v_size = vo;
    out.s_size = so;
    return 0;
}

// ---- global truncation ----------------------------------------------------------------------------
//   cutoff   -> truncbelow(10^-svalue), src:1007-1010 (keep Schmidt values > cutoff)
//   chi_full -> truncdim(D), src:1363-1365 (largest values while sum (2S+1) kept <= D)
// Order at the cut: by sqrt(2S+1) * Schmidt value (= tilde value) for weighting 0, by the Schmidt value for 1; ties
// broken by sector then index, so kept sets are prefixes per sector; the largest multiplet is always kept.
void truncate(const std::vector<double>& vals, const std::vector<int>& lens, const std::vector<int>& qdims, int chi_full,
              double cutoff, int weighting, std::vector<int>& counts, double& trunc_weight, double& kept_norm) {
    const double rel_floor = 1e-14;
    const size_t nsec = lens.size();
    counts.assign(nsec, 0);
    double smax = 0.0, total = 0.0;
    for (double v : vals) {
        smax = std::max(smax, v);
        total += v * v;
    }
    // The candidates of every sector are a PREFIX of its descending list (both filters are monotone inside a sector), so the
    // global order "key descending, ties by sector then index" is a k-way merge of at most a few dozen sorted runs: a heap
    // of one cursor per sector instead of a sort of every value (this runs on the host with the GPU idle, once per bond).
    struct Cur {
        double key;
        int sid, idx;
    };
    auto worse = [](const Cur& a, const Cur& b) {        // heap order: the BEST cursor on top
        if (a.key != b.key) return a.key < b.key;
        if (a.sid != b.sid) return a.sid > b.sid;
        return a.idx > b.idx;
    };
    std::vector<size_t> start(nsec);
    std::vector<Cur> heap;
    size_t p = 0;
    for (size_t k = 0; k < nsec; ++k) {
        start[k] = p;
        p += (size_t)lens[k];
    }
    auto admissible = [&](size_t k, int i) {
        if (i >= lens[k]) return false;
        const double v = vals[start[k] + i], schmidt = v / sqrt((double)qdims[k]);
        return schmidt > cutoff && v > rel_floor * smax;
    };
    auto key_of = [&](size_t k, int i) { return weighting == 0 ? vals[start[k] + i] : vals[start[k] + i] / sqrt((double)qdims[k]); };
    for (size_t k = 0; k < nsec; ++k)
        if (admissible(k, 0)) heap.push_back({key_of(k, 0), (int)k, 0});
    std::make_heap(heap.begin(), heap.end(), worse);
    int64_t tot = 0;
    double kept_w = 0.0;
    size_t kept = 0;
    while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), worse);
        const Cur c = heap.back();
        heap.pop_back();
        if (chi_full > 0) {
            tot += qdims[c.sid];
            if (tot > chi_full && kept > 0) break;      // (the largest multiplet is always kept)
            if (tot > chi_full) {                        // the very first candidate alone exceeds the limit: keep it, stop
                counts[c.sid] += 1;
                const double v = vals[start[c.sid] + c.idx];
                kept_w += v * v;
                ++kept;
                break;
            }
        }
        counts[c.sid] += 1;
        const double v = vals[start[c.sid] + c.idx];
        kept_w += v * v;
        ++kept;
        if (admissible((size_t)c.sid, c.idx + 1)) {
            heap.push_back({key_of((size_t)c.sid, c.idx + 1), c.sid, c.idx + 1});
            std::push_heap(heap.begin(), heap.end(), worse);
        }
    }
    trunc_weight = total > 0.0 ? (total - kept_w) / total : 0.0;
    kept_norm = sqrt(kept_w);
}

// ---- post-SVD finalisation -------------------------------------------------------------------------------------------
// Writes the truncated A (left layout) and B (right layout) into ONE output buffer (A at offA, B at offB).  keep[i] =
// kept count of block i of sp.  The per-update column-index array (kept columns in value order, block after block) is
// supplied at run time; here only its offsets are fixed.
//   iso_g : isometry columns taken from G' and divided by sigma           (global scale 1)
//   cen_g : centre tensor taken from G' (mode B; sigma cancels)           (global scale 1/nrm)
//   iso_v : isometry taken from the accumulated rotation J (mode B)       (global scale 1)
//   cen   : mode-A centres, U^H M or M V, as grouped-GEMM segments with alpha = 1
//           buffers: BUF_X = theta (scaled by 1/nrm by the caller), BUF_S1 = BUF_Y = the output buffer
void plan_finalize(const ThetaLayout& tl, const SvdPlan& sp, const std::vector<int>& keep, const SiteLayout& layA,
                   const SiteLayout& layB, bool right, int64_t offA, int64_t offB, FinalizePlan& out) {
    out = FinalizePlan();
    TaskList cen;
    int64_t ioff = 0;
    for (size_t i = 0; i < sp.mids.size(); ++i) {
        const int k = keep[i];
        if (k == 0) continue;
        const Sec c = sp.mids[i];
        const auto& M = tl.mats[i];
        const htn_svd_block& d = sp.desc[i];
        const int64_t oA = offA + layA.mats[layA.mat(c)].off;
        const int64_t oB = offB + layB.mats[layB.mat(c)].off;
        const bool acc = sp.accumulate[i];
        htn_copy_item itA, itB;
        memset(&itA, 0, sizeof(itA));
        memset(&itB, 0, sizeof(itB));
        // A (rows x k, ld rows): gather columns ; B (k x cols, ld k): conjugate-transposed gather of columns
        itA.dst_off = oA, itA.rows = M.rows, itA.cols = k, itA.ldd = M.rows;
        itA.idx_off = ioff, itA.gather_dim = 1, itA.op = HTN_OP_N, itA.scl_off = d.s_off;
        itB.dst_off = oB, itB.rows = k, itB.cols = M.cols, itB.ldd = k;
        itB.idx_off = ioff, itB.gather_dim = 0, itB.op = HTN_OP_C, itB.scl_off = d.s_off;
        ioff += k;
        if (right && !acc) {                 // mode A: G = M, G' = U Sigma; centre S V^H = U^H M
            itA.src_off = d.g_off, itA.lds = d.m, itA.scale_dim = 1, itA.inv_norm = 1;
            out.iso_g.push_back(itA);
            const int bi = cen.block(mk(c.N, c.j), BUF_Y, oB, k, M.cols, k);
            cen.gemm(bi, BUF_S1, oA, M.rows, HTN_OP_C, BUF_X, M.off, M.rows, HTN_OP_N, M.rows, 1.0);
        } else if (right && acc) {           // mode B: G = M^H, G' = V Sigma, J = U
            itA.src_off = d.v_off, itA.lds = d.n, itA.scale_dim = -1, itA.inv_norm = 0;
            out.iso_v.push_back(itA);
            itB.src_off = d.g_off, itB.lds = d.m, itB.scale_dim = -1, itB.inv_norm = 0;
            out.cen_g.push_back(itB);        // S V^H = conj(G')^T
        } else if (!right && !acc) {         // mode A: G = M^H, G' = V Sigma; centre U S = M V
            itB.src_off = d.g_off, itB.lds = d.m, itB.scale_dim = 0, itB.inv_norm = 1;
            out.iso_g.push_back(itB);
            const int bi = cen.block(mk(c.N, c.j), BUF_Y, oA, M.rows, k, M.rows);
            cen.gemm(bi, BUF_X, M.off, M.rows, HTN_OP_N, BUF_S1, oB, k, HTN_OP_C, M.cols, 1.0);
        } else {                             // mode B: G = M, G' = U Sigma, J = V
            itB.src_off = d.v_off, itB.lds = d.n, itB.scale_dim = -1, itB.inv_norm = 0;
            out.iso_v.push_back(itB);
            itA.src_off = d.g_off, itA.lds = d.m, itA.scale_dim = -1, itA.inv_norm = 0;
            out.cen_g.push_back(itA);        // U S = G'
        }
    }
    out.n_idx = ioff;
    out.has_cen = !cen.blocks.empty();
    if (out.has_cen) cen.finalize(out.cen);
}

}  // namespace htn
