// Host-side core of the two-site DMRG engine: sector layouts, recoupling coefficients, the contraction compiler
// (sector tables + reduced MPO -> htn_tile / htn_seg task lists), the global truncation rule and the bond-update /
// sweep driver.  Pure C++17, no HIP: the same translation units are linked into libhubbardtn_hip.so (device work =
// hand-written gfx950 kernels through htn::Backend) and into the CPU baseline library built from oracle/cpu_backend.
//
// Stands in for what TensorKit 0.14.6 (fusion trees, tree transformers, tsvd!, truncation), KrylovKit 0.9.5 (eigsolve)
// and MPSKit 0.13.1 (two-site sweep, environments) do below `find_groundstate(psi0, H, alg)`
// (src/HubbardFunctions.jl:1010); SURVEY.md 8a a6-a10, App. A.
#pragma once
#include <stdint.h>
#include <string.h>

#include <array>
#include <complex>
#include <memory>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "hubbardtn_hip.h"

namespace htn {

typedef std::complex<double> cplx;

// buffer-table slots shared by all plans
enum { BUF_X = 0, BUF_Y, BUF_L, BUF_R, BUF_Z, BUF_S1, BUF_S2, BUF_AUX };

struct Sec {
    int32_t N, j;
};
inline bool operator<(const Sec& a, const Sec& b) { return a.N < b.N || (a.N == b.N && a.j < b.j); }
inline bool operator==(const Sec& a, const Sec& b) { return a.N == b.N && a.j == b.j; }
inline bool operator!=(const Sec& a, const Sec& b) { return !(a == b); }
inline uint64_t skey(Sec s) { return ((uint64_t)(uint32_t)s.N << 32) | (uint32_t)s.j; }

// ---- symmetry ------------------------------------------------------------------------------------------------
struct Sym {
    int kind = HTN_SYM_SU2_U1;
    int n_site = 3;
    Sec site[HTN_MAX_SITE] = {{0, 0}, {1, 1}, {2, 0}, {0, 0}};
    bool su2() const { return kind != HTN_SYM_U1_U1; }
    int wrapN(int N) const { return kind == HTN_SYM_SU2 ? (N & 1) : N; }
    int qdim(Sec c) const { return su2() ? c.j + 1 : 1; }             // quantum dimension 2S + 1
    int jeff(int j) const { return su2() ? j : 0; }                   // spin entering the recoupling coefficients
    bool triangle(int a, int b, int c) const;                         // may (a, b) couple to c
    bool connects(Sec ket, int dN, int k, Sec bra) const { return bra.N == wrapN(ket.N + dN) && triangle(ket.j, k, bra.j); }
    void fuse(Sec sec, int s, std::vector<Sec>& out) const;           // sectors reachable by adding site multiplet s
    void split(Sec sec, int s, std::vector<Sec>& out) const;          // sectors c with c (x) s -> sec
};

// ---- closed-form SU(2) recoupling coefficients (doubled spins) -------------------------------------------------
double wigner6j(int j1, int j2, int j3, int j4, int j5, int j6);
double wigner9j(int j1, int j2, int j3, int j4, int j5, int j6, int j7, int j8, int j9);
double coef_left(int jbp, int k, int jb, int jsp, int js, int kop, int jap, int kp, int ja);
double coef_right(int jcp, int k, int jc, int jsp, int js, int kop, int jbp, int kp, int jb);
double coef_apply(int ja, int jap, int k, int js1, int js1p, int kop1, int km, int jc, int jcp, int js2, int js2p,
                  int kop2, int kp, int jb, int jbp);

// ---- generic small keys ------------------------------------------------------------------------------------------
typedef std::array<int32_t, 12> Key;
struct KeyHash {
    size_t operator()(const Key& k) const {
        uint64_t h = 1469598103934665603ull;
        for (int32_t v : k) {
            h ^= (uint32_t)v;
            h *= 1099511628211ull;
        }
        return (size_t)h;
    }
};
typedef std::unordered_map<Key, int, KeyHash> KeyMap;
inline Key mk(int32_t a = -9, int32_t b = -9, int32_t c = -9, int32_t d = -9, int32_t e = -9, int32_t f = -9, int32_t g = -9,
              int32_t h = -9, int32_t i = -9, int32_t j = -9, int32_t k = -9, int32_t l = -9) {
    return Key{a, b, c, d, e, f, g, h, i, j, k, l};
}

// ---- bonds and layouts ------------------------------------------------------------------------------------------
struct Bond {
    std::vector<Sec> secs;                 // sorted
    std::vector<int32_t> dims;
    std::unordered_map<uint64_t, int> index;
    std::string key;                       // bytes of (N, j, dim) triples: identifies the table in plan caches
    std::string seckey;                    // labels only
    Bond() {}
    explicit Bond(std::vector<std::pair<Sec, int>> items);
    int find(Sec s) const {
        auto it = index.find(skey(s));
        return it == index.end() ? -1 : it->second;
    }
    bool has(Sec s) const { return index.count(skey(s)) != 0; }
    int dim(Sec s) const {
        const int i = find(s);
        return i < 0 ? 0 : dims[i];
    }
    int64_t dim_full(const Sym& sym) const;
    int multiplets() const;
};
typedef std::shared_ptr<const Bond> BondP;

struct BlockRec {
    int64_t off;
    int32_t m, n, ld;
};
struct Grp {
    Sec sec;
    int s;
};

struct SiteLayout {              // kind 'L': matrices per right sector [(l, s) rows ; n_r]; 'R': per left sector [n_l ; (s, r) cols]
    char kind;
    BondP bl, br;
    std::vector<Key> bkeys;      // (l.N, l.j, s, r.N, r.j) in layout order
    std::vector<BlockRec> blocks;
    KeyMap bidx;
    struct Mat {
        Sec c;
        int64_t off;
        int32_t rows, cols;
        std::vector<Grp> groups;
    };
    std::vector<Mat> mats;
    std::unordered_map<uint64_t, int> midx;
    int64_t size = 0;
    int block(Sec l, int s, Sec r) const {
        auto it = bidx.find(mk(l.N, l.j, s, r.N, r.j));
        return it == bidx.end() ? -1 : it->second;
    }
    int mat(Sec c) const {
        auto it = midx.find(skey(c));
        return it == midx.end() ? -1 : it->second;
    }
};
typedef std::shared_ptr<const SiteLayout> SiteLayoutP;
SiteLayoutP build_site_layout(const Sym& sym, char kind, BondP bl, BondP br);

struct ThetaLayout {             // per mid sector c one dense matrix M_c[(a, s1) rows ; (s2, b) cols]
    BondP bl, br;
    struct Mat {
        Sec c;
        int64_t off;
        int32_t rows, cols;
        std::vector<Grp> rows_g, cols_g;       // (a, s1) sorted by (a, s1); (b, s2) sorted by (s2, b)
        std::vector<int32_t> roffs, coffs;
    };
    std::vector<Mat> mats;                     // in sorted order of c (= mids)
    std::unordered_map<uint64_t, int> midx;
    std::vector<Key> bkeys;                    // (a, s1, c, s2, b)
    std::vector<BlockRec> blocks;
    KeyMap bidx;
    int64_t size = 0;
    int block(Sec a, int s1, Sec c, int s2, Sec b) const {
        auto it = bidx.find(mk(a.N, a.j, s1, c.N, c.j, s2, b.N, b.j));
        return it == bidx.end() ? -1 : it->second;
    }
};
typedef std::shared_ptr<const ThetaLayout> ThetaLayoutP;
ThetaLayoutP build_theta_layout(const Sym& sym, BondP bl, BondP br);

struct Lvl {
    int32_t dN, k;
};
struct EnvLayout {               // side 'L': key (bra, w, ket) -> [n_bra, n_ket]; 'R': key (ket, w, bra) -> [n_ket, n_bra]
    char side;
    BondP bond;
    std::vector<Lvl> levels;
    int ident;
    std::vector<Key> bkeys;      // (x.N, x.j, w, y.N, y.j)
    struct Blk {
        int64_t off;
        int32_t m, n;
    };
    std::vector<Blk> blocks;
    KeyMap bidx;
    std::unordered_map<Key, std::vector<Sec>, KeyHash> by_ket;      // (w, ket) -> [bra]
    int64_t size = 0;
    int block(Sec x, int w, Sec y) const {
        auto it = bidx.find(mk(x.N, x.j, w, y.N, y.j));
        return it == bidx.end() ? -1 : it->second;
    }
    const std::vector<Sec>* kets(int w, Sec ket) const {
        auto it = by_ket.find(mk(w, ket.N, ket.j));
        return it == by_ket.end() ? nullptr : &it->second;
    }
};
typedef std::shared_ptr<const EnvLayout> EnvLayoutP;
EnvLayoutP build_env_layout(const Sym& sym, char side, BondP bond, const std::vector<Lvl>& levels);

// ---- MPO ---------------------------------------------------------------------------------------------------------
struct MpoEntry {
    int32_t wl, wr, op;
    cplx coef;
};
struct MpoSite {
    std::vector<Lvl> left, right;
    std::vector<MpoEntry> entries;
    std::string key;             // bytes identifying (left, right, entries) in plan caches
};
struct SiteOp {
    int k, dN;
    double red[HTN_MAX_SITE][HTN_MAX_SITE];
};
struct Mpo {
    Sym sym;
    std::vector<SiteOp> ops;
    std::vector<MpoSite> sites;
};

// ---- task lists --------------------------------------------------------------------------------------------------
struct Tasks {
    std::vector<htn_tile> tiles;
    int32_t ntiles = 0;
    std::vector<htn_seg> segs;
    int32_t nsegs = 0;
    int64_t flops = 0;           // algorithmic complex128 flops (8 per MAC) of the GEMM segments
    // optional placement hint for balance_tiles: ascending start offsets (in the output buffer) of groups of output blocks
    // whose tiles read the same operands (H_eff apply: one group per coupled-sector matrix of theta); tiles of a group are
    // steered to one XCD so that its L2 fetches those operands once.  Empty: tiles are grouped by output row strip.
    std::vector<int64_t> group_bounds;
};

// Load balance of one grouped-GEMM launch (HIP backend): the launch is bound by its longest tile's dependent K loop,
// so tiles with more than ~(total slabs / CUs) K slabs are cut into parts (htn_tile.part / nparts) that run on
// different workgroups and meet through the split-K workspace; the records are then ordered longest first and dealt
// so that tiles sharing operand panels (same output block and row strip) land on the same XCD's L2 under the
// round-robin workgroup placement.  Returns the number of workspace slabs the list uses (0: nothing was split).
int balance_tiles(Tasks& t, int n_cus);

struct ApplyPlan {
    Tasks tz, ty;
    bool has_z = false;
    int64_t zsize = 0;
    int64_t nterms = 0;
};
void plan_apply(const Mpo& mpo, const ThetaLayout& tl, const EnvLayout& Ll, const EnvLayout& Rl, const MpoSite& W1,
                const MpoSite& W2, ApplyPlan& out);
void plan_theta(const char* mode, const SiteLayout& lay1, const SiteLayout& lay2, const ThetaLayout& tl, Tasks& out);
struct EnvPlan {
    Tasks t1, t2;
    int64_t zsize = 0;
};
void plan_left_env(const Mpo& mpo, const EnvLayout& Ll, const SiteLayout& lay, const MpoSite& W, const EnvLayout& Lnew,
                   EnvPlan& out);
void plan_right_env(const Mpo& mpo, const EnvLayout& Rl, const SiteLayout& lay, const MpoSite& W, const EnvLayout& Rnew,
                    EnvPlan& out);

struct SvdPlan {
    std::vector<htn_svd_block> desc;
    std::vector<htn_copy_item> stage;
    std::vector<Sec> mids;
    std::vector<char> transposed, accumulate;
    int64_t g_size = 0, v_size = 0, s_size = 0;
    int32_t max_m = 0;
    int64_t flops = 0;           // LAPACK-equivalent flops, SURVEY 8(d)
    bool any_accumulate = false;
};
int plan_svd(const ThetaLayout& tl, bool right, SvdPlan& out);          // != 0: block too tall (htn error set)

// global truncation over sectors.  vals: per sector DESCENDING tilde singular values, concatenated
// in sector order; lens[k] values in sector k; qdims[k] = 2S+1.  -> counts[k], discarded weight, norm of the kept part
void truncate(const std::vector<double>& vals, const std::vector<int>& lens, const std::vector<int>& qdims, int chi_full,
              double cutoff, int weighting, std::vector<int>& counts, double& trunc_weight, double& kept_norm);

struct FinalizePlan {
    std::vector<htn_copy_item> iso_g, cen_g, iso_v;
    Tasks cen;
    bool has_cen = false;
    int64_t n_idx = 0;           // entries of the per-update column-index array (kept columns, block after block)
};
void plan_finalize(const ThetaLayout& tl, const SvdPlan& sp, const std::vector<int>& keep, const SiteLayout& layA,
                   const SiteLayout& layB, bool right, int64_t offA, int64_t offB, FinalizePlan& out);

// ---- device backend ----------------------------------------------------------------------------------------------
struct Backend {
    virtual ~Backend() {}
    virtual int kind() const = 0;
    virtual int activate() { return 0; }                 // make this backend's device current for the calling thread
    virtual void* alloc(size_t bytes) = 0;               // device memory, reuse is stream ordered; nullptr on failure
    virtual void release(void* p) = 0;
    virtual int upload(void* dst, const void* src_host, size_t bytes) = 0;      // source staged before return
    virtual int download(void* dst_host, const void* src, size_t bytes) = 0;    // synchronous
    virtual int zero(void* p, size_t bytes) = 0;
    virtual int sync() = 0;
    virtual int grouped_gemm(const void* const* bufs, const htn_tile* tiles, int32_t n_tiles, const htn_seg* segs) = 0;
    virtual int lanczos(const htn_gemm_launch* stages, int n_stages, int x_slot, int y_slot, void* V, int64_t n,
                        int krylovdim, double tol, int max_restart, int zero_y, htn_exchange2_fn exchange, void* user,
                        double* eig, int* n_matvec, double* residual, double* matvec_ms) = 0;
    virtual int jacobi_svd(void* G, void* Vj, double* S, const htn_svd_block* desc_dev, const htn_svd_block* desc_host,
                           int n_blocks, int max_m, int max_sweeps, double tol, int32_t* info_dev,
                           const htn_svd_opts* opts) = 0;
    virtual int batched_copy(void* dst, const void* src, const int32_t* idx, const double* scl,
                             const htn_copy_item* items_dev, int n_items, double gscale) = 0;
    virtual int scale(void* x, int64_t n, double f) = 0;                         // complex vector x *= f
    // multi-rank reduction of y (sum over ranks) enqueued on the stream; default: not available
    virtual int set_comm(int rank, int world, const void* id) { return 1; }
    virtual bool has_comm() const { return false; }
    virtual int allreduce(void* y, int64_t n) { return 1; }
    bool timing = false;
};

int set_error(const char* fmt, ...);        // writes the thread-local error string, returns 1
char* err_buf();

}  // namespace htn
