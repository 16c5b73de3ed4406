// Bond-update / sweep driver behind the C ABI (include/hubbardtn_hip.h, "Bond-update / sweep level").
//
// Stands in for MPSKit's two-site sweep body reached from
//     find_groundstate(psi0, H, IDMRG2(; trscheme, tol))                src/HubbardFunctions.jl:1010
// per bond: form theta, Lanczos lowest eigenpair of the AC2 effective Hamiltonian, per-sector SVD + global truncation,
// write back, move the environment.  Sweep order follows MPSKit's DMRG2: bonds 1..L-1 going
// right, L-2..1 going left (2L-3 updates).  All tensors stay on the device between bonds; the host sees the Lanczos
// tridiagonal coefficients and the singular values (needed for the global truncation rule, App. A.6).
// Host C++ (no HIP): device work goes through htn::Backend.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>

#include "htn_core.h"

namespace htn {
Backend* make_backend(int backend, int device, void* stream);       // one per library (htn_backend_hip.hip / cpu)
}
using namespace htn;

// Handles are reference counted: an htn_mps keeps its context and its MPO alive, so destroying the handles in any order
// (garbage-collected host languages do exactly that) is safe; the last release frees the object.
struct htn_ctx {
    std::atomic<int> refs{1};
    std::unique_ptr<Backend> be;
    int rank = 0, world = 1;
    bool shard = false;              // zero y + reduce after every matvec (world > 1, or forced for tests)
    htn_exchange2_fn exch = nullptr;
    void* exch_user = nullptr;
};
struct htn_mpo {
    std::atomic<int> refs{1};
    htn_ctx* ctx;
    Mpo mpo;
};
static void ctx_release(htn_ctx* c) {
    if (c && --c->refs == 0) delete c;
}
static void mpo_release(htn_mpo* m) {
    if (m && --m->refs == 0) {
        htn_ctx* c = m->ctx;
        delete m;
        ctx_release(c);
    }
}

namespace {

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct DBuf {
    Backend* be;
    void* p;
    size_t bytes;
    DBuf(Backend* b, size_t n) : be(b), p(nullptr), bytes(n) { p = be->alloc(std::max<size_t>(n, 16)); }
    ~DBuf() {
        if (p) be->release(p);
    }
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
};
typedef std::shared_ptr<DBuf> DBufP;
struct DView {                     // complex128 window of a device buffer
    DBufP base;
    int64_t off = 0;
    cplx* ptr() const { return base ? (cplx*)base->p + off : nullptr; }
};

struct DevTasks {
    DBufP mem;
    const htn_tile* tiles = nullptr;
    const htn_seg* segs = nullptr;
    int32_t ntiles = 0, nsegs = 0;
    int32_t ws_slots = 0;          // split-K workspace slabs the list needs (balance_tiles)
    int64_t flops = 0;
};

struct ApplyC {
    DevTasks dz, dy;
    bool has_z = false;
    int64_t zsize = 0, flops = 0;
    int32_t ntiles = 0, nsegs = 0;
};
struct EnvC {
    EnvLayoutP lay;
    DevTasks d1, d2;
    int64_t zsize = 0, flops = 0;
};
struct SvdC {
    SvdPlan sp;
    DBufP stage, desc;
    // sector-sharded SVD (world > 1): the blocks this rank owns (LPT over ~ m n^2), as their own staging / descriptor lists
    std::vector<htn_svd_block> own_desc;
    DBufP own_stage, own_desc_dev;
    int n_own = 0;
};
struct FinC {
    SiteLayoutP layA, layB;
    DBufP items;                   // iso_g | cen_g | iso_v copy items
    const htn_copy_item *ig = nullptr, *cg = nullptr, *iv = nullptr;
    int n_ig = 0, n_cg = 0, n_iv = 0;
    DevTasks cen;
    bool has_cen = false;
};

struct Spectrum {
    std::vector<Sec> secs;
    std::vector<std::vector<double>> vals;
};

}  // namespace

struct htn_mps {
    htn_ctx* ctx;
    htn_mpo* mpo_handle = nullptr;
    Backend* be;
    const Mpo* mpo;
    int L;
    std::vector<BondP> bonds;
    std::vector<SiteLayoutP> site_lay;
    std::vector<DView> site_buf;
    std::vector<EnvLayoutP> Llay, Rlay;
    std::vector<DView> Lbuf, Rbuf;
    std::unordered_map<std::string, std::shared_ptr<const void>> cache;
    int64_t hits = 0, misses = 0;
    double energy = 0.0;
    std::map<int, Spectrum> spectra;
    std::map<int, std::pair<std::pair<int, double>, double>> cut_hint;      // bond -> ((chi, cutoff), smallest kept value)
    std::map<int, int> sweeps_hint;     // bond -> outer Jacobi sweeps its large blocks needed last time (htn_svd_opts.sweeps_hint)
    std::vector<int32_t> idx_host;

    template <class T, class F>
    std::shared_ptr<T> cached(const std::string& key, F build) {
        auto it = cache.find(key);
        if (it != cache.end()) {
            ++hits;
            return std::const_pointer_cast<T>(std::static_pointer_cast<const T>(it->second));
        }
        if (cache.size() > 20000) cache.clear();
        ++misses;
        std::shared_ptr<T> v = build();
        if (v) cache[key] = v;
        return v;
    }
    static std::string ikey(const char* tag, int i) {
        std::string k(tag);
        k.append((const char*)&i, sizeof(i));
        return k;
    }
    DBufP dalloc(size_t bytes) {
        auto b = std::make_shared<DBuf>(be, bytes);
        return b->p ? b : nullptr;
    }
    DView zalloc(int64_t n, bool zero) {
        DView v;
        v.base = dalloc(sizeof(cplx) * (size_t)std::max<int64_t>(n, 1));
        if (v.base && zero) be->zero(v.base->p, sizeof(cplx) * (size_t)std::max<int64_t>(n, 1));
        return v;
    }
    DView ws;                      // split-K workspace of the grouped GEMM: [tickets | slabs], grown on demand
    int64_t ws_slots = -1;
    int ensure_ws(int slots) {
        if (slots <= ws_slots) return 0;
        const int64_t want = std::max<int64_t>(2 * (int64_t)slots, 256);
        ws = zalloc(HTN_WS_ELEMS(want), false);
        if (!ws.base) return set_error("device allocation of the split-K workspace failed");
        if (be->zero(ws.ptr(), sizeof(cplx) * HTN_WS_TICKET_ELEMS)) return 1;      // tickets start at zero; the kernel resets them
        ws_slots = want;
        return 0;
    }
    int upload_tasks(const Tasks& t_in, DevTasks& d) {
        Tasks balanced;
        const Tasks* tp = &t_in;
        d.ws_slots = 0;
        if (be->kind() == HTN_BACKEND_HIP && t_in.ntiles > 0) {        // (the CPU baseline runs tiles as they are)
            balanced = t_in;
            balanced.tiles.resize((size_t)t_in.ntiles);
            d.ws_slots = balance_tiles(balanced, 256);
            tp = &balanced;
        }
        const Tasks& t = *tp;
        const size_t tb = (sizeof(htn_tile) * t.tiles.size() + 63) / 64 * 64, sb = sizeof(htn_seg) * t.segs.size();
        d.mem = dalloc(tb + sb);
        if (!d.mem) return set_error("device allocation of a task list failed");
        if (be->upload(d.mem->p, t.tiles.data(), sizeof(htn_tile) * t.tiles.size())) return 1;
        if (be->upload((char*)d.mem->p + tb, t.segs.data(), sb)) return 1;
        d.tiles = (const htn_tile*)d.mem->p;
        d.segs = (const htn_seg*)((char*)d.mem->p + tb);
        d.ntiles = t.ntiles;
        d.nsegs = t.nsegs;
        d.flops = t.flops;
        return 0;
    }
    // (htn_plan_apply_dump shows the lists BEFORE balancing: those are what the Python statement of the planner emits)
    SiteLayoutP site_layout(char kind, BondP bl, BondP br) {
        return cached<const SiteLayout>(std::string("slay") + kind + bl->key + "|" + br->key,
                                        [&] { return build_site_layout(mpo->sym, kind, bl, br); });
    }
    ThetaLayoutP theta_layout(BondP bl, BondP br) {
        return cached<const ThetaLayout>(std::string("tl") + bl->key + "|" + br->key,
                                         [&] { return build_theta_layout(mpo->sym, bl, br); });
    }
    int gemm(const DevTasks& d, std::initializer_list<std::pair<int, const void*>> bufs) {
        if (d.ntiles == 0) return 0;
        if (ensure_ws(d.ws_slots)) return 1;
        const void* table[HTN_MAX_BUFS] = {nullptr};
        for (auto& kv : bufs) table[kv.first] = kv.second;
        table[HTN_BUF_WS] = ws.ptr();
        return be->grouped_gemm(table, d.tiles, d.ntiles, d.segs);
    }

    int left_env(int i);
    int right_env(int i);
    std::shared_ptr<ApplyC> make_apply(int i, const ThetaLayout& tl);
    int theta_into(int i, const ThetaLayout& tl, cplx* dst);
    int update_bond(int i, int direction, bool right, bool optimise, const htn_sweep_opts& o, htn_bond_stats* st);
    int sweep(const htn_sweep_opts& o, htn_bond_stats* st, double* E);
};

// GL on bond i+1 from GL on bond i and the left-layout tensor of site i
int htn_mps::left_env(int i) {
    const SiteLayout& lay = *site_lay[i];
    if (lay.kind != 'L') return set_error("left_env: site %d is not in left layout", i);
    const MpoSite& W = mpo->sites[i];
    auto c = cached<EnvC>(ikey("lenv", i) + bonds[i]->key + "|" + bonds[i + 1]->key, [&]() -> std::shared_ptr<EnvC> {
        auto e = std::make_shared<EnvC>();
        e->lay = build_env_layout(mpo->sym, 'L', bonds[i + 1], W.right);
        EnvPlan p;
        plan_left_env(*mpo, *Llay[i], lay, W, *e->lay, p);
        if (upload_tasks(p.t1, e->d1) || upload_tasks(p.t2, e->d2)) return nullptr;
        e->zsize = p.zsize;
        e->flops = p.t1.flops + p.t2.flops;
        return e;
    });
    if (!c) return 1;
    DView z = zalloc(c->zsize, false), out = zalloc(c->lay->size, false);
    if (!z.base || !out.base) return set_error("device allocation failed (left environment)");
    if (gemm(c->d1, {{BUF_L, Lbuf[i].ptr()}, {BUF_S1, site_buf[i].ptr()}, {BUF_Z, z.ptr()}})) return 1;
    if (gemm(c->d2, {{BUF_S1, site_buf[i].ptr()}, {BUF_Z, z.ptr()}, {BUF_Y, out.ptr()}})) return 1;
    Llay[i + 1] = c->lay;
    Lbuf[i + 1] = out;
    return 0;
}

// GR on bond i from GR on bond i+1 and the right-layout tensor of site i
int htn_mps::right_env(int i) {
    const SiteLayout& lay = *site_lay[i];
    if (lay.kind != 'R') return set_error("right_env: site %d is not in right layout", i);
    const MpoSite& W = mpo->sites[i];
    auto c = cached<EnvC>(ikey("renv", i) + bonds[i]->key + "|" + bonds[i + 1]->key, [&]() -> std::shared_ptr<EnvC> {
        auto e = std::make_shared<EnvC>();
        e->lay = build_env_layout(mpo->sym, 'R', bonds[i], W.left);
        EnvPlan p;
        plan_right_env(*mpo, *Rlay[i + 1], lay, W, *e->lay, p);
        if (upload_tasks(p.t1, e->d1) || upload_tasks(p.t2, e->d2)) return nullptr;
        e->zsize = p.zsize;
        e->flops = p.t1.flops + p.t2.flops;
        return e;
    });
    if (!c) return 1;
    DView z = zalloc(c->zsize, false), out = zalloc(c->lay->size, false);
    if (!z.base || !out.base) return set_error("device allocation failed (right environment)");
    if (gemm(c->d1, {{BUF_R, Rbuf[i + 1].ptr()}, {BUF_S1, site_buf[i].ptr()}, {BUF_Z, z.ptr()}})) return 1;
    if (gemm(c->d2, {{BUF_S1, site_buf[i].ptr()}, {BUF_Z, z.ptr()}, {BUF_Y, out.ptr()}})) return 1;
    Rlay[i] = c->lay;
    Rbuf[i] = out;
    return 0;
}

// compiled H_eff apply of bond (i, i+1); with a sharded context the Y-stage tiles are dealt round-robin over the ranks
// (tiles are in LPT order, so dealing balances MACs; every rank keeps the full segment table and the full Z stage)
std::shared_ptr<ApplyC> htn_mps::make_apply(int i, const ThetaLayout& tl) {
    return cached<ApplyC>(ikey("apply", i) + bonds[i]->key + "|" + bonds[i + 2]->key, [&]() -> std::shared_ptr<ApplyC> {
        ApplyPlan p;
        plan_apply(*mpo, tl, *Llay[i], *Rlay[i + 2], mpo->sites[i], mpo->sites[i + 1], p);
        auto a = std::make_shared<ApplyC>();
        a->has_z = p.has_z;
        a->zsize = p.zsize;
        a->flops = p.ty.flops + (p.has_z ? p.tz.flops : 0);
        a->ntiles = p.ty.ntiles + (p.has_z ? p.tz.ntiles : 0);
        a->nsegs = p.ty.nsegs + (p.has_z ? p.tz.nsegs : 0);
        if (ctx->world > 1) {
            std::vector<htn_tile> sel;
            for (int t = ctx->rank; t < p.ty.ntiles; t += ctx->world) sel.push_back(p.ty.tiles[t]);
            p.ty.ntiles = (int32_t)sel.size();
            if (sel.empty()) sel.push_back(p.ty.tiles[0]);
            p.ty.tiles.swap(sel);
        }
        if (p.has_z && upload_tasks(p.tz, a->dz)) return nullptr;
        if (upload_tasks(p.ty, a->dy)) return nullptr;
        return a;
    });
}

int htn_mps::theta_into(int i, const ThetaLayout& tl, cplx* dst) {
    const SiteLayout &l1 = *site_lay[i], &l2 = *site_lay[i + 1];
    const char mode[3] = {l1.kind, l2.kind, 0};
    if (strcmp(mode, "RR") && strcmp(mode, "LL") && strcmp(mode, "LR")) return set_error("theta: centre is not on sites (%d, %d)", i, i + 1);
    auto d = cached<DevTasks>(std::string("theta") + mode + bonds[i]->key + "|" + bonds[i + 1]->key + "|" + bonds[i + 2]->key,
                              [&]() -> std::shared_ptr<DevTasks> {
                                  Tasks t;
                                  plan_theta(mode, l1, l2, tl, t);
                                  auto dt = std::make_shared<DevTasks>();
                                  if (upload_tasks(t, *dt)) return nullptr;
                                  return dt;
                              });
    if (!d) return 1;
    return gemm(*d, {{BUF_S1, site_buf[i].ptr()}, {BUF_S2, site_buf[i + 1].ptr()}, {BUF_Y, dst}});
}

static int exchange_tramp(void* y, int64_t n, void* user) {
    htn_ctx* ctx = (htn_ctx*)user;
    if (ctx->exch) return ctx->exch(y, n, ctx->exch_user);
    return ctx->be->allreduce(y, n);
}

int htn_mps::update_bond(int i, int direction, bool right, bool optimise, const htn_sweep_opts& o, htn_bond_stats* st) {
    if (i < 0 || i + 1 >= L) return set_error("htn_bond_update: bond index %d out of range", i);
    const double t0 = now();
    const Sym& sym = mpo->sym;
    BondP bl = bonds[i], br = bonds[i + 2];
    ThetaLayoutP tlp = theta_layout(bl, br);
    const ThetaLayout& tl = *tlp;
    const int64_t n = tl.size;
    const int kd = o.krylovdim > 0 ? o.krylovdim : 30;
    if (n <= 0) return set_error("htn_bond_update: empty two-site tensor on bond %d", i);
    DView V = zalloc((int64_t)(kd + 2) * n, false);
    if (!V.base) return set_error("device allocation of the Krylov basis failed (%lld elements)", (long long)((kd + 2) * n));
    if (theta_into(i, tl, V.ptr())) return 1;                  // theta -> V[0] (the Lanczos driver normalises it)
    auto ap = make_apply(i, tl);
    if (!ap) return 1;
    DView z = zalloc(ap->zsize, false);
    if (ensure_ws(std::max(ap->dy.ws_slots, ap->has_z ? ap->dz.ws_slots : 0))) return 1;
    htn_gemm_launch stages[2];
    memset(stages, 0, sizeof(stages));
    stages[0].bufs[HTN_BUF_WS] = stages[1].bufs[HTN_BUF_WS] = ws.ptr();
    int ns = 0;
    if (ap->has_z) {
        stages[ns].bufs[BUF_L] = Lbuf[i].ptr();
        stages[ns].bufs[BUF_Z] = z.ptr();
        stages[ns].tiles = ap->dz.tiles, stages[ns].segs = ap->dz.segs, stages[ns].n_tiles = ap->dz.ntiles;
        ++ns;
    }
    stages[ns].bufs[BUF_L] = Lbuf[i].ptr();
    stages[ns].bufs[BUF_R] = Rbuf[i + 2].ptr();
    stages[ns].bufs[BUF_Z] = z.ptr();
    stages[ns].tiles = ap->dy.tiles, stages[ns].segs = ap->dy.segs, stages[ns].n_tiles = ap->dy.ntiles;
    ++ns;
    if (o.profile) be->sync();
    const double t_plan = now() - t0;
    double E = 0.0, res = 0.0, mv_ms = 0.0;
    int nmv = 0;
    const bool shard = ctx->shard;
    if (be->lanczos(stages, ns, BUF_X, BUF_Y, V.ptr(), n, kd, optimise ? o.lanczos_tol : 1e300, o.maxrestart, shard ? 1 : 0,
                    shard ? exchange_tramp : nullptr, ctx, &E, &nmv, &res, be->timing ? &mv_ms : nullptr))
        return 1;
    if (o.profile) be->sync();
    const double t_lan = now() - t0 - t_plan;
    cplx* x = V.ptr();
    // ---- SVD + truncation ----
    auto sc = cached<SvdC>(std::string(right ? "svdR" : "svdL") + bl->key + "|" + br->key, [&]() -> std::shared_ptr<SvdC> {
        auto s = std::make_shared<SvdC>();
        if (plan_svd(tl, right, s->sp)) return nullptr;
        s->stage = dalloc(sizeof(htn_copy_item) * s->sp.stage.size());
        s->desc = dalloc(sizeof(htn_svd_block) * s->sp.desc.size());
        if (!s->stage || !s->desc) return nullptr;
        if (be->upload(s->stage->p, s->sp.stage.data(), sizeof(htn_copy_item) * s->sp.stage.size())) return nullptr;
        if (be->upload(s->desc->p, s->sp.desc.data(), sizeof(htn_svd_block) * s->sp.desc.size())) return nullptr;
        if (ctx->world > 1) {
            // owner of every block: longest first onto the least loaded rank (deterministic: every rank computes the same map)
            const int nbk = (int)s->sp.mids.size();
            std::vector<int> ord(nbk);
            for (int b = 0; b < nbk; ++b) ord[b] = b;
            auto cost = [&](int b) { return (double)s->sp.desc[b].m * s->sp.desc[b].n * s->sp.desc[b].n; };
            std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return cost(a) > cost(b); });
            std::vector<double> load(ctx->world, 0.0);
            std::vector<htn_copy_item> st_own;
            for (int b : ord) {
                int r = 0;
                for (int q = 1; q < ctx->world; ++q)
                    if (load[q] < load[r]) r = q;
                load[r] += cost(b);
                if (r == ctx->rank) {
                    s->own_desc.push_back(s->sp.desc[b]);
                    st_own.push_back(s->sp.stage[b]);
                }
            }
            s->n_own = (int)s->own_desc.size();
            if (s->n_own) {
                s->own_stage = dalloc(sizeof(htn_copy_item) * st_own.size());
                s->own_desc_dev = dalloc(sizeof(htn_svd_block) * s->own_desc.size());
                if (!s->own_stage || !s->own_desc_dev) return nullptr;
                if (be->upload(s->own_stage->p, st_own.data(), sizeof(htn_copy_item) * st_own.size())) return nullptr;
                if (be->upload(s->own_desc_dev->p, s->own_desc.data(), sizeof(htn_svd_block) * s->own_desc.size())) return nullptr;
            }
        }
        return s;
    });
    if (!sc) return 1;
    const SvdPlan& sp = sc->sp;
    const int nb = (int)sp.mids.size();
    DView G = zalloc(sp.g_size, false), Vj = zalloc(sp.v_size, false);
    // singular values and per-block sweep counts share one buffer: ONE device-to-host copy (one stream sync) per bond
    const size_t s_elems = ((size_t)std::max<int64_t>(sp.s_size, 1) + 1) & ~(size_t)1, i_elems = (size_t)std::max(nb, 1);
    DBufP S = dalloc(sizeof(double) * s_elems + sizeof(int32_t) * i_elems);
    if (!G.base || !Vj.base || !S) return set_error("device allocation failed (SVD workspace)");
    int32_t* info_dev = (int32_t*)((double*)S->p + s_elems);
    // Sector-sharded SVD (world > 1): every rank stages and decomposes only the blocks it owns; everything else
    // in G / S (and the rotation workspace of accumulate-mode blocks) stays zero and ONE sum over ranks per buffer
    // hands every rank the complete result -- bit-identical everywhere (x + 0 + ... + 0), so the ranks stay in lock step.
    const bool svd_shard = ctx->world > 1 && ctx->shard;
    if (svd_shard) {
        if (be->zero(G.ptr(), sizeof(cplx) * (size_t)std::max<int64_t>(sp.g_size, 1))) return 1;
        if (be->zero(S->p, sizeof(double) * s_elems + sizeof(int32_t) * i_elems)) return 1;
        if (sp.any_accumulate && be->zero(Vj.ptr(), sizeof(cplx) * (size_t)std::max<int64_t>(sp.v_size, 1))) return 1;
        if (sc->n_own && be->batched_copy(G.ptr(), x, nullptr, nullptr, (const htn_copy_item*)sc->own_stage->p, sc->n_own, 1.0)) return 1;
    } else if (be->batched_copy(G.ptr(), x, nullptr, nullptr, (const htn_copy_item*)sc->stage->p, nb, 1.0))
        return 1;
    // Singular directions far below what the truncation keeps need not be resolved (optional, OFF by default).
    // truncbelow(eta): everything below eta goes anyway.  truncdim(D): if the previous update of this bond (same D) was
    // limited by D, its smallest kept value is where the cut will fall again.  x is normalised: values compare across sweeps.
    htn_svd_opts so;
    int32_t jac_used = 0;
    so.split_elems = o.svd_split_elems;
    so.sweeps_hint = 0;
    so.rank_cut = 0.0;
    so.sweeps_used = &jac_used;
    {       // the previous update of this bond tells how many outer sweeps the large blocks will need (speculation bound)
        auto h = sweeps_hint.find(i + 1);
        if (h != sweeps_hint.end()) so.sweeps_hint = h->second;
    }
    if (o.rank_cut > 0.0) {
        double cut = 0.0;
        auto h = cut_hint.find(i + 1);
        if (h != cut_hint.end() && h->second.first.first == o.chi_full && h->second.first.second == o.cutoff)
            cut = o.rank_cut * h->second.second;
        so.rank_cut = std::max(cut, o.rank_cut * o.cutoff);
    }
    if (svd_shard) {
        if (sc->n_own && be->jacobi_svd(G.ptr(), Vj.ptr(), (double*)S->p, (const htn_svd_block*)sc->own_desc_dev->p,
                                        sc->own_desc.data(), sc->n_own, sp.max_m, o.jacobi_max_sweeps > 0 ? o.jacobi_max_sweeps : 40,
                                        o.jacobi_tol > 0.0 ? o.jacobi_tol : 1e-14, info_dev, &so))
            return 1;
        if (exchange_tramp(G.ptr(), std::max<int64_t>(sp.g_size, 1), ctx)) return set_error("sharded SVD: exchange of the blocks failed");
        if (exchange_tramp(S->p, (int64_t)(s_elems / 2), ctx)) return set_error("sharded SVD: exchange of the singular values failed");
        if (sp.any_accumulate && exchange_tramp(Vj.ptr(), std::max<int64_t>(sp.v_size, 1), ctx)) return 1;
    } else if (be->jacobi_svd(G.ptr(), Vj.ptr(), (double*)S->p, (const htn_svd_block*)sc->desc->p, sp.desc.data(), nb, sp.max_m,
                              o.jacobi_max_sweeps > 0 ? o.jacobi_max_sweeps : 40, o.jacobi_tol > 0.0 ? o.jacobi_tol : 1e-14,
                              info_dev, &so))
        return 1;
    if (jac_used > 0) sweeps_hint[i + 1] = jac_used;
    static const bool dbg_timers = getenv("HTN_DEBUG_HOST_TIMERS") != nullptr;
    const double tA = now();
    std::vector<double> s_host(s_elems + (i_elems + 1) / 2);
    if (be->download(s_host.data(), S->p, sizeof(double) * s_elems + sizeof(int32_t) * i_elems)) return 1;
    const double tB = now();
    const int32_t* info_h = (const int32_t*)(s_host.data() + s_elems);
    int jac_sweeps = 0;
    for (int b = 0; b < (svd_shard ? sc->n_own : nb); ++b) {        // (sharded: the counts of this rank's own blocks)
        if (info_h[b] < 0) return set_error("Jacobi SVD did not converge (bond %d, block %d, %d sweeps)", i + 1, b, -info_h[b]);
        jac_sweeps = std::max(jac_sweeps, (int)info_h[b]);
    }
    // per-block descending order (ties: original column index ascending) and the global truncation
    std::vector<int> lens(nb), qd(nb);
    std::vector<std::vector<int>> order(nb);
    std::vector<double> vals;
    vals.reserve((size_t)sp.s_size);
    std::vector<std::pair<double, int>> tmp;
    for (int b = 0; b < nb; ++b) {
        const int len = sp.desc[b].n;
        const double* sv = s_host.data() + sp.desc[b].s_off;
        lens[b] = len;
        qd[b] = sym.qdim(sp.mids[b]);
        tmp.resize(len);
        for (int k = 0; k < len; ++k) tmp[k] = {sv[k], k};
        std::sort(tmp.begin(), tmp.end(), [](const std::pair<double, int>& a, const std::pair<double, int>& c) {
            return a.first != c.first ? a.first > c.first : a.second < c.second;
        });
        order[b].resize(len);
        for (int k = 0; k < len; ++k) {
            order[b][k] = tmp[k].second;
            vals.push_back(tmp[k].first);
        }
    }
    std::vector<int> counts;
    double tw = 0.0, nrm = 0.0;
    truncate(vals, lens, qd, o.chi_full, o.cutoff, o.weighting, counts, tw, nrm);
    std::vector<std::pair<Sec, int>> mid_items;
    int64_t kept_tot = 0, positive = 0;
    for (int b = 0; b < nb; ++b) {
        if (counts[b] > 0) mid_items.push_back({sp.mids[b], counts[b]});
        kept_tot += counts[b];
    }
    for (double v : vals) positive += v > 0.0;
    if (kept_tot == 0 || !(nrm > 0.0)) return set_error("htn_bond_update: nothing kept by the truncation on bond %d", i + 1);
    BondP mid = std::make_shared<Bond>(mid_items);
    const double tC = now();
    // hint for the next visit of this bond: the smallest kept value, valid only if the dimension limit (not the number
    // of available states) ended the kept set
    if (o.chi_full > 0 && tw > 0.0 && kept_tot < positive) {
        double smin = 1e300;
        size_t p = 0;
        for (int b = 0; b < nb; ++b) {
            if (counts[b] > 0) smin = std::min(smin, vals[p + counts[b] - 1]);
            p += lens[b];
        }
        cut_hint[i + 1] = {{o.chi_full, o.cutoff}, smin};
    } else
        cut_hint.erase(i + 1);
    // the finalisation plan depends on the kept COUNTS only (not on which columns carry them): memoised
    std::string ckey((const char*)counts.data(), sizeof(int) * counts.size());
    auto fc = cached<FinC>(std::string(right ? "finR" : "finL") + bl->key + "|" + br->key + "|" + ckey, [&]() -> std::shared_ptr<FinC> {
        auto f = std::make_shared<FinC>();
        f->layA = site_layout('L', bl, mid);
        f->layB = site_layout('R', mid, br);
        FinalizePlan fp;
        plan_finalize(tl, sp, counts, *f->layA, *f->layB, right, 0, f->layA->size, fp);
        f->n_ig = (int)fp.iso_g.size(), f->n_cg = (int)fp.cen_g.size(), f->n_iv = (int)fp.iso_v.size();
        const size_t tot = (size_t)(f->n_ig + f->n_cg + f->n_iv);
        if (tot) {
            std::vector<htn_copy_item> all;
            all.insert(all.end(), fp.iso_g.begin(), fp.iso_g.end());
            all.insert(all.end(), fp.cen_g.begin(), fp.cen_g.end());
            all.insert(all.end(), fp.iso_v.begin(), fp.iso_v.end());
            f->items = dalloc(sizeof(htn_copy_item) * tot);
            if (!f->items || be->upload(f->items->p, all.data(), sizeof(htn_copy_item) * tot)) return nullptr;
            f->ig = (const htn_copy_item*)f->items->p;
            f->cg = f->ig + f->n_ig;
            f->iv = f->cg + f->n_cg;
        }
        f->has_cen = fp.has_cen;
        if (fp.has_cen && upload_tasks(fp.cen, f->cen)) return nullptr;
        return f;
    });
    if (!fc) return 1;
    const double tD = now();
    idx_host.clear();
    for (int b = 0; b < nb; ++b)
        for (int k = 0; k < counts[b]; ++k) idx_host.push_back(order[b][k]);
    DBufP idx_d = dalloc(sizeof(int32_t) * std::max<size_t>(idx_host.size(), 1));
    const int64_t sizeA = fc->layA->size, sizeB = fc->layB->size;
    DView out = zalloc(sizeA + sizeB, true);
    if (!idx_d || !out.base) return set_error("device allocation failed (site tensors)");
    if (be->upload(idx_d->p, idx_host.data(), sizeof(int32_t) * idx_host.size())) return 1;
    const int32_t* idxp = (const int32_t*)idx_d->p;
    const double* Sp = (const double*)S->p;
    if (fc->n_ig && be->batched_copy(out.ptr(), G.ptr(), idxp, Sp, fc->ig, fc->n_ig, 1.0)) return 1;
    if (fc->n_cg && be->batched_copy(out.ptr(), G.ptr(), idxp, Sp, fc->cg, fc->n_cg, 1.0 / nrm)) return 1;
    if (fc->n_iv && be->batched_copy(out.ptr(), Vj.ptr(), idxp, Sp, fc->iv, fc->n_iv, 1.0)) return 1;
    if (fc->has_cen) {
        if (be->scale(x, n, 1.0 / nrm)) return 1;               // centre = U^H (M / nrm)
        if (gemm(fc->cen, {{BUF_X, x}, {BUF_S1, out.ptr()}, {BUF_Y, out.ptr()}})) return 1;
    }
    bonds[i + 1] = mid;
    site_lay[i] = fc->layA;
    site_buf[i] = DView{out.base, out.off};
    site_lay[i + 1] = fc->layB;
    site_buf[i + 1] = DView{out.base, out.off + sizeA};
    if (dbg_timers)
        fprintf(stderr, "bond %d: svd call %.1f us, download %.1f, sort+truncate %.1f, finalize plan %.1f, enqueue %.1f  misses %lld hits %lld nvals %zu\n", i + 1,
                (tA - t0 - t_plan - t_lan) * 1e6, (tB - tA) * 1e6, (tC - tB) * 1e6, (tD - tC) * 1e6, (now() - tD) * 1e6, (long long)misses, (long long)hits, vals.size());
    if (o.profile) be->sync();
    const double t_svd = now() - t0 - t_plan - t_lan;
    if (right ? left_env(i) : right_env(i + 1)) return 1;
    if (o.profile) be->sync();
    const double t_env = now() - t0 - t_plan - t_lan - t_svd;
    energy = E;
    Spectrum spec;
    {
        size_t p = 0;
        for (int b = 0; b < nb; ++b) {
            if (counts[b] > 0) {
                spec.secs.push_back(sp.mids[b]);
                std::vector<double> v(counts[b]);
                const double f = 1.0 / nrm / sqrt((double)qd[b]);
                for (int k = 0; k < counts[b]; ++k) v[k] = vals[p + k] * f;
                spec.vals.push_back(std::move(v));
            }
            p += lens[b];
        }
    }
    spectra[i + 1] = std::move(spec);
    if (st) {
        memset(st, 0, sizeof(*st));
        st->bond = i + 1;
        st->direction = direction;
        st->n_matvec = nmv;
        st->jacobi_sweeps = jac_sweeps;
        st->chi_full = (int32_t)mid->dim_full(sym);
        st->multiplets = mid->multiplets();
        st->n_tiles = ap->ntiles;
        st->n_segs = ap->nsegs;
        st->theta_size = n;
        st->apply_flops = ap->flops;
        st->apply_bytes = 16 * (2 * n + Llay[i]->size + Rlay[i + 2]->size);
        st->svd_flops = sp.flops;
        st->energy = E;
        st->residual = res;
        st->trunc_weight = tw;
        st->t_plan = t_plan;
        st->t_lanczos = t_lan;
        st->t_svd = t_svd;
        st->t_env = t_env;
        st->t_total = now() - t0;
        st->matvec_ms = mv_ms > 0.0 ? mv_ms : 0.0;      // (0: none of this solve's launches fell on the 1-in-8 timing sample)
    }
    return 0;
}

int htn_mps::sweep(const htn_sweep_opts& o, htn_bond_stats* st, double* E) {
    int k = 0;
    for (int i = 0; i < L - 1; ++i, ++k)
        if (update_bond(i, +1, i < L - 2, true, o, st ? st + k : nullptr)) return 1;
    for (int i = L - 3; i >= 0; --i, ++k)
        if (update_bond(i, -1, false, true, o, st ? st + k : nullptr)) return 1;
    if (E) *E = energy;
    return 0;
}

// =====================================================================================================================
// C ABI
// =====================================================================================================================
extern "C" {

const char* htn_last_error(void) { return err_buf(); }
int htn_abi_version(void) { return HTN_ABI_VERSION; }

int htn_ctx_create(int32_t backend, int32_t device, void* stream, htn_ctx** out) {
    if (!out) return set_error("htn_ctx_create: out is NULL");
    Backend* be = make_backend(backend, device, stream);
    if (!be) return 1;
    htn_ctx* c = new htn_ctx();
    c->be.reset(be);
    *out = c;
    return 0;
}
void htn_ctx_destroy(htn_ctx* ctx) { ctx_release(ctx); }
int htn_ctx_backend(const htn_ctx* ctx) { return ctx->be->kind(); }
int htn_ctx_set_timing(htn_ctx* ctx, int32_t on) {
    ctx->be->timing = on != 0;
    return 0;
}
int htn_ctx_set_comm(htn_ctx* ctx, int32_t rank, int32_t world, const void* id_host) {
    if (world < 1 || rank < 0 || rank >= world) return set_error("htn_ctx_set_comm: bad rank %d / world %d", rank, world);
    if (ctx->be->activate()) return 1;
    if (ctx->be->set_comm(rank, world, id_host)) return 1;
    ctx->rank = rank;
    ctx->world = world;
    ctx->shard = true;
    ctx->exch = nullptr;
    return 0;
}
int htn_ctx_set_exchange(htn_ctx* ctx, int32_t rank, int32_t world, htn_exchange2_fn fn, void* user) {
    if (world < 1 || rank < 0 || rank >= world) return set_error("htn_ctx_set_exchange: bad rank %d / world %d", rank, world);
    ctx->rank = rank;
    ctx->world = world;
    ctx->exch = fn;
    ctx->exch_user = user;
    ctx->shard = fn != nullptr;
    return 0;
}

int htn_mpo_create(htn_ctx* ctx, const htn_symmetry* sym, int32_t nsites, const htn_site_op* ops, int32_t n_ops,
                   const int32_t* level_ptr, const int32_t* levels, const int32_t* entry_ptr, const htn_mpo_entry* entries,
                   htn_mpo** out) {
    if (!ctx || !sym || !out || nsites < 2) return set_error("htn_mpo_create: bad arguments");
    if (sym->n_site < 1 || sym->n_site > HTN_MAX_SITE || sym->kind < 0 || sym->kind > 2) return set_error("htn_mpo_create: bad symmetry");
    auto m = std::make_unique<htn_mpo>();
    m->ctx = ctx;
    m->mpo.sym.kind = sym->kind;
    m->mpo.sym.n_site = sym->n_site;
    for (int s = 0; s < sym->n_site; ++s) m->mpo.sym.site[s] = {sym->site_N[s], sym->site_j[s]};
    for (int k = 0; k < n_ops; ++k) {
        SiteOp o;
        o.k = ops[k].k;
        o.dN = ops[k].dN;
        for (int a = 0; a < HTN_MAX_SITE; ++a)
            for (int b = 0; b < HTN_MAX_SITE; ++b) o.red[a][b] = ops[k].red[a * HTN_MAX_SITE + b];
        m->mpo.ops.push_back(o);
    }
    auto lv = [&](int b) {
        std::vector<Lvl> v;
        for (int q = level_ptr[b]; q < level_ptr[b + 1]; ++q) v.push_back({levels[2 * q], levels[2 * q + 1]});
        return v;
    };
    for (int i = 0; i < nsites; ++i) {
        MpoSite s;
        s.left = lv(i);
        s.right = lv(i + 1);
        if (s.left.empty() || s.right.empty()) return set_error("htn_mpo_create: site %d has an empty MPO bond", i);
        for (int q = entry_ptr[i]; q < entry_ptr[i + 1]; ++q) {
            const htn_mpo_entry& e = entries[q];
            if (e.wl < 0 || e.wl >= (int)s.left.size() || e.wr < 0 || e.wr >= (int)s.right.size() || e.op < 0 || e.op >= n_ops)
                return set_error("htn_mpo_create: entry %d of site %d out of range", q - entry_ptr[i], i);
            s.entries.push_back({e.wl, e.wr, e.op, cplx(e.coef_re, e.coef_im)});
        }
        s.key.append((const char*)s.left.data(), sizeof(Lvl) * s.left.size());
        s.key.append("|");
        s.key.append((const char*)s.right.data(), sizeof(Lvl) * s.right.size());
        s.key.append("|");
        for (auto& e : s.entries) {
            int32_t r[3] = {e.wl, e.wr, e.op};
            double c[2] = {e.coef.real(), e.coef.imag()};
            s.key.append((const char*)r, sizeof(r));
            s.key.append((const char*)c, sizeof(c));
        }
        m->mpo.sites.push_back(std::move(s));
    }
    if (m->mpo.sites.front().left.size() != 1 || m->mpo.sites.back().right.size() != 1) {
        // a window inside a larger system (iDMRG) has full-width boundary bonds: allowed, the boundary environments
        // then must be supplied to htn_mps_create
    }
    ++ctx->refs;
    *out = m.release();
    return 0;
}
void htn_mpo_destroy(htn_mpo* mpo) { mpo_release(mpo); }

int htn_mps_create(htn_ctx* ctx, const htn_mpo* mpo, int32_t nsites, const int32_t* bond_ptr, const htn_sector* sectors,
                   const int32_t* sub_ptr, const htn_subblock* subs, const int64_t* data_ptr, const void* data_host,
                   const void* left_env_host, const void* right_env_host, htn_mps** out) {
    if (!ctx || !mpo || !out) return set_error("htn_mps_create: NULL argument");
    if (nsites != (int)mpo->mpo.sites.size()) return set_error("htn_mps_create: %d sites but the MPO has %d", nsites, (int)mpo->mpo.sites.size());
    if (ctx->be->activate()) return 1;
    // (errors below return through the guard: it drops the references the half-built object took)
    struct Guard {
        htn_mps* p;
        ~Guard() {
            if (p) htn_mps_destroy(p);
        }
    } guard{new htn_mps()};
    htn_mps* e = guard.p;
    e->ctx = ctx;
    ++ctx->refs;
    e->mpo_handle = const_cast<htn_mpo*>(mpo);
    ++e->mpo_handle->refs;
    e->be = ctx->be.get();
    e->mpo = &mpo->mpo;
    e->L = nsites;
    const Sym& sym = mpo->mpo.sym;
    for (int b = 0; b <= nsites; ++b) {
        std::vector<std::pair<Sec, int>> items;
        for (int q = bond_ptr[b]; q < bond_ptr[b + 1]; ++q) items.push_back({{sectors[q].N, sectors[q].j}, sectors[q].count});
        e->bonds.push_back(std::make_shared<Bond>(items));
        if (e->bonds.back()->secs.empty()) return set_error("htn_mps_create: bond %d is empty", b);
    }
    e->site_lay.resize(nsites);
    e->site_buf.resize(nsites);
    e->Llay.resize(nsites + 1);
    e->Rlay.resize(nsites + 1);
    e->Lbuf.resize(nsites + 1);
    e->Rbuf.resize(nsites + 1);
    const cplx* data = (const cplx*)data_host;
    std::vector<cplx> flat;
    for (int i = 0; i < nsites; ++i) {
        SiteLayoutP lay = e->site_layout('R', e->bonds[i], e->bonds[i + 1]);
        flat.assign((size_t)std::max<int64_t>(lay->size, 1), cplx(0.0, 0.0));
        for (int q = sub_ptr[i]; q < sub_ptr[i + 1]; ++q) {
            const htn_subblock& sb = subs[q];
            const int bi = lay->block({sb.lN, sb.lj}, sb.s, {sb.rN, sb.rj});
            if (bi < 0) continue;          // a sub-block between sectors the bond tables do not hold
            const BlockRec& r = lay->blocks[bi];
            if (sb.ld < r.m) return set_error("htn_mps_create: sub-block of site %d has ld %d < %d rows", i, sb.ld, r.m);
            const cplx* src = data + data_ptr[i] + sb.off;
            for (int c = 0; c < r.n; ++c)
                for (int rr = 0; rr < r.m; ++rr) flat[(size_t)(r.off + rr + (int64_t)c * r.ld)] = src[rr + (int64_t)c * sb.ld];
        }
        e->site_lay[i] = lay;
        e->site_buf[i] = e->zalloc(lay->size, false);
        if (!e->site_buf[i].base) return set_error("htn_mps_create: device allocation failed");
        if (e->be->upload(e->site_buf[i].ptr(), flat.data(), sizeof(cplx) * flat.size())) return 1;
    }
    // boundaries: an open end (no environment blocks: only the implicit identity level), or -- for a window inside a
    // larger system -- the environment of the block beyond that end, in this library's block order
    e->Llay[0] = build_env_layout(sym, 'L', e->bonds[0], mpo->mpo.sites[0].left);
    e->Lbuf[0] = e->zalloc(e->Llay[0]->size, true);
    e->Rlay[nsites] = build_env_layout(sym, 'R', e->bonds[nsites], mpo->mpo.sites[nsites - 1].right);
    e->Rbuf[nsites] = e->zalloc(e->Rlay[nsites]->size, true);
    if (!e->Lbuf[0].base || !e->Rbuf[nsites].base) return set_error("htn_mps_create: device allocation failed");
    if (left_env_host && e->Llay[0]->size && e->be->upload(e->Lbuf[0].ptr(), left_env_host, sizeof(cplx) * e->Llay[0]->size)) return 1;
    if (right_env_host && e->Rlay[nsites]->size &&
        e->be->upload(e->Rbuf[nsites].ptr(), right_env_host, sizeof(cplx) * e->Rlay[nsites]->size))
        return 1;
    if (!left_env_host && e->Llay[0]->size) return set_error("htn_mps_create: the left MPO bond is not a boundary: left_env required");
    if (!right_env_host && e->Rlay[nsites]->size) return set_error("htn_mps_create: the right MPO bond is not a boundary: right_env required");
    for (int i = nsites - 1; i >= 1; --i)
        if (e->right_env(i)) return 1;
    if (e->be->sync()) return 1;
    guard.p = nullptr;
    *out = e;
    return 0;
}
void htn_mps_destroy(htn_mps* mps) {
    if (!mps) return;
    htn_ctx* c = mps->ctx;
    htn_mpo* m = mps->mpo_handle;
    if (mps->be) (void)mps->be->activate();
    delete mps;              // device buffers go back to the backend's pool first ...
    mpo_release(m);          // ... then the references that kept the backend alive
    ctx_release(c);
}

static htn_sweep_opts norm_opts(const htn_sweep_opts* o) {
    htn_sweep_opts d;
    memset(&d, 0, sizeof(d));
    d.krylovdim = 30;
    d.maxrestart = 3;
    d.lanczos_tol = 1e-12;
    d.jacobi_tol = 1e-14;
    d.jacobi_max_sweeps = 40;
    if (!o) return d;
    htn_sweep_opts r = *o;
    if (r.krylovdim <= 0) r.krylovdim = d.krylovdim;
    if (r.lanczos_tol <= 0.0) r.lanczos_tol = d.lanczos_tol;
    if (r.jacobi_tol <= 0.0) r.jacobi_tol = d.jacobi_tol;
    if (r.jacobi_max_sweeps <= 0) r.jacobi_max_sweeps = d.jacobi_max_sweeps;
    if (r.maxrestart < 0) r.maxrestart = 0;
    return r;
}

int htn_bond_update(htn_mps* mps, int32_t i, int32_t direction, int32_t placement, int32_t optimise, const htn_sweep_opts* opts,
                    htn_bond_stats* stats) {
    if (mps->be->activate()) return 1;
    return mps->update_bond(i, direction, placement == 0, optimise != 0, norm_opts(opts), stats);
}
int htn_dmrg2_sweep(htn_mps* mps, const htn_sweep_opts* opts, htn_bond_stats* stats, double* energy) {
    if (mps->be->activate()) return 1;
    return mps->sweep(norm_opts(opts), stats, energy);
}

int64_t htn_mps_theta_size(htn_mps* mps, int32_t i) {
    if (i < 0 || i + 1 >= mps->L) return -1;
    return mps->theta_layout(mps->bonds[i], mps->bonds[i + 2])->size;
}
int htn_mps_get_theta(htn_mps* mps, int32_t i, void* theta_host) {
    if (i < 0 || i + 1 >= mps->L) return set_error("htn_mps_get_theta: bond out of range");
    if (mps->be->activate()) return 1;
    ThetaLayoutP tl = mps->theta_layout(mps->bonds[i], mps->bonds[i + 2]);
    DView t = mps->zalloc(tl->size, false);
    if (!t.base) return set_error("device allocation failed");
    if (mps->theta_into(i, *tl, t.ptr())) return 1;
    return mps->be->download(theta_host, t.ptr(), sizeof(cplx) * tl->size);
}
int htn_heff2_apply(htn_mps* mps, int32_t i, const void* x_host, void* y_host) {
    if (i < 0 || i + 1 >= mps->L) return set_error("htn_heff2_apply: bond out of range");
    if (mps->be->activate()) return 1;
    ThetaLayoutP tl = mps->theta_layout(mps->bonds[i], mps->bonds[i + 2]);
    auto ap = mps->make_apply(i, *tl);
    if (!ap) return 1;
    const int64_t n = tl->size;
    DView x = mps->zalloc(n, false), y = mps->zalloc(n, true), z = mps->zalloc(ap->zsize, false);
    if (!x.base || !y.base || !z.base) return set_error("device allocation failed");
    if (mps->be->upload(x.ptr(), x_host, sizeof(cplx) * n)) return 1;
    if (ap->has_z && mps->gemm(ap->dz, {{BUF_X, x.ptr()}, {BUF_L, mps->Lbuf[i].ptr()}, {BUF_Z, z.ptr()}})) return 1;
    if (mps->gemm(ap->dy, {{BUF_X, x.ptr()}, {BUF_Y, y.ptr()}, {BUF_L, mps->Lbuf[i].ptr()}, {BUF_R, mps->Rbuf[i + 2].ptr()}, {BUF_Z, z.ptr()}}))
        return 1;
    if (mps->ctx->shard && exchange_tramp(y.ptr(), n, mps->ctx)) return set_error("htn_heff2_apply: exchange failed");
    return mps->be->download(y_host, y.ptr(), sizeof(cplx) * n);
}

int32_t htn_mps_nsites(const htn_mps* mps) { return mps->L; }
int32_t htn_mps_bond(const htn_mps* mps, int32_t b, htn_sector* out) {
    if (b < 0 || b > mps->L) return -1;
    const Bond& B = *mps->bonds[b];
    if (out)
        for (size_t k = 0; k < B.secs.size(); ++k) out[k] = {B.secs[k].N, B.secs[k].j, B.dims[k]};
    return (int32_t)B.secs.size();
}
int64_t htn_mps_spectrum(const htn_mps* mps, int32_t b, htn_sector* secs, double* values) {
    auto it = mps->spectra.find(b);
    if (it == mps->spectra.end()) return 0;
    int64_t tot = 0;
    for (size_t k = 0; k < it->second.secs.size(); ++k) {
        const auto& v = it->second.vals[k];
        if (secs) secs[k] = {it->second.secs[k].N, it->second.secs[k].j, (int32_t)v.size()};
        if (values) memcpy(values + tot, v.data(), sizeof(double) * v.size());
        tot += (int64_t)v.size();
    }
    return tot;
}
int64_t htn_mps_site_size(const htn_mps* mps, int32_t i, int32_t* kind) {
    if (i < 0 || i >= mps->L) return -1;
    if (kind) *kind = mps->site_lay[i]->kind;
    return mps->site_lay[i]->size;
}
int32_t htn_mps_get_site(const htn_mps* mps, int32_t i, htn_subblock* subs, void* data_host) {
    if (i < 0 || i >= mps->L) return -1;
    const SiteLayout& lay = *mps->site_lay[i];
    if (subs)
        for (size_t q = 0; q < lay.blocks.size(); ++q) {
            const Key& k = lay.bkeys[q];
            subs[q] = {k[0], k[1], k[2], k[3], k[4], lay.blocks[q].ld, lay.blocks[q].off};
        }
    if (data_host && lay.size && mps->be->activate()) return -1;
    if (data_host && lay.size && mps->be->download(data_host, mps->site_buf[i].ptr(), sizeof(cplx) * lay.size)) return -1;
    return (int32_t)lay.blocks.size();
}
int64_t htn_mps_env_size(const htn_mps* mps, int32_t side, int32_t b) {
    if (b < 0 || b > mps->L) return -1;
    const EnvLayoutP& l = side == 0 ? mps->Llay[b] : mps->Rlay[b];
    return l ? l->size : -1;
}
int htn_mps_get_env(const htn_mps* mps, int32_t side, int32_t b, void* data_host) {
    if (b < 0 || b > mps->L) return set_error("htn_mps_get_env: bond out of range");
    const EnvLayoutP& l = side == 0 ? mps->Llay[b] : mps->Rlay[b];
    if (!l) return set_error("htn_mps_get_env: environment %d of bond %d does not exist yet", side, b);
    if (l->size == 0) return 0;
    if (mps->be->activate()) return 1;
    return mps->be->download(data_host, (side == 0 ? mps->Lbuf[b] : mps->Rbuf[b]).ptr(), sizeof(cplx) * l->size);
}
int32_t htn_mps_env_bond(const htn_mps* mps, int32_t side, int32_t b, htn_sector* out) {
    if (b < 0 || b > mps->L) return -1;
    const EnvLayoutP& l = side == 0 ? mps->Llay[b] : mps->Rlay[b];
    if (!l) return -1;
    const Bond& B = *l->bond;
    if (out)
        for (size_t k = 0; k < B.secs.size(); ++k) out[k] = {B.secs[k].N, B.secs[k].j, B.dims[k]};
    return (int32_t)B.secs.size();
}
int32_t htn_mps_env_blocks(const htn_mps* mps, int32_t side, int32_t b, htn_env_block* out) {
    if (b < 0 || b > mps->L) return -1;
    const EnvLayoutP& l = side == 0 ? mps->Llay[b] : mps->Rlay[b];
    if (!l) return -1;
    if (out)
        for (size_t q = 0; q < l->blocks.size(); ++q) {
            const Key& k = l->bkeys[q];
            out[q] = {k[0], k[1], k[2], k[3], k[4], l->blocks[q].m, l->blocks[q].n, 0, l->blocks[q].off};
        }
    return (int32_t)l->blocks.size();
}
int htn_plan_apply_dump(htn_mps* mps, int32_t i, int32_t stage, int32_t* n_tiles, htn_tile* tiles, int32_t* n_segs, htn_seg* segs,
                        int64_t* z_size, int64_t* flops) {
    if (i < 0 || i + 1 >= mps->L) return set_error("htn_plan_apply_dump: bond out of range");
    ThetaLayoutP tl = mps->theta_layout(mps->bonds[i], mps->bonds[i + 2]);
    ApplyPlan p;
    plan_apply(*mps->mpo, *tl, *mps->Llay[i], *mps->Rlay[i + 2], mps->mpo->sites[i], mps->mpo->sites[i + 1], p);
    const Tasks& t = stage == 0 ? p.tz : p.ty;
    const bool have = stage != 0 || p.has_z;
    if (n_tiles) *n_tiles = have ? t.ntiles : 0;
    if (n_segs) *n_segs = have ? t.nsegs : 0;
    if (have && tiles) memcpy(tiles, t.tiles.data(), sizeof(htn_tile) * t.ntiles);
    if (have && segs) memcpy(segs, t.segs.data(), sizeof(htn_seg) * t.nsegs);
    if (z_size) *z_size = p.zsize;
    if (flops) *flops = have ? t.flops : 0;
    return 0;
}
int32_t htn_balance_tiles(const htn_tile* tiles, int32_t n_tiles, int32_t n_cus, htn_tile* out, int32_t out_cap, int32_t* n_out) {
    Tasks t;
    t.tiles.assign(tiles, tiles + std::max(n_tiles, 0));
    t.ntiles = n_tiles;
    const int ws = balance_tiles(t, n_cus > 0 ? n_cus : 256);
    if (n_out) *n_out = t.ntiles;
    if (out) {
        if (t.ntiles > out_cap) {
            set_error("htn_balance_tiles: %d records do not fit the output array (%d)", t.ntiles, out_cap);
            return -1;
        }
        memcpy(out, t.tiles.data(), sizeof(htn_tile) * (size_t)t.ntiles);
    }
    return ws;
}
int htn_mps_cache_stats(const htn_mps* mps, int64_t* hits, int64_t* misses) {
    if (hits) *hits = mps->hits;
    if (misses) *misses = mps->misses;
    return 0;
}

}  // extern "C"
