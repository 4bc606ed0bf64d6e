// C ABI of libmg_hip (declared in include/mg_hip.h): solver handle, cycle
// driver and the single-operator test entry points.
#include "../../include/mg_hip.h"
#include "mg_internal.h"
#include "mg_kernels.h"

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using mgk::Coef;
using mgk::IRange;
using mgk::View;

namespace {

thread_local char g_err[512] = "";

// Largest node count per axis: several launchers put the plane count in gridDim.z and groups of
// four rows in gridDim.y, both limited to 65535 workgroups.
constexpr int64_t kMaxDim = 65535;

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace

int mgk::set_error(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

namespace {

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (void)hipGetLastError(); /* reported here; not left behind for a later call's check */ \
            return fail(e_ == hipErrorOutOfMemory ? MG_ERR_NOMEM : MG_ERR_HIP, "%s: %s (%s:%d)", \
                        #expr, hipGetErrorString(e_), __FILE__, __LINE__);                   \
        }                                                                                    \
    } while (0)

#define TRY(expr)            \
    do {                     \
        int rc_ = (expr);    \
        if (rc_) return rc_; \
    } while (0)

int select_device(int device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
        return fail(MG_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= count)
        return fail(MG_ERR_INVALID, "device %d out of range (have %d)", device, count);
    HIP_TRY(hipSetDevice(device));
    return MG_OK;
}

// 1/(4^l dx dx) exactly as the reference writes it (cpp/PotentialSolver.cpp:3728-3750)
Coef level_coef(const double dh[3], int level)
{
    double f = 1.0;
    for (int l = 0; l < level; l++) f *= 4.0;
    double c[3];
    for (int a = 0; a < 3; a++) c[a] = (level == 0) ? 1.0 / (dh[a] * dh[a]) : 1.0 / (f * dh[a] * dh[a]);
    return mgk::make_coef(c[0], c[1], c[2], 2 * c[0] + 2 * c[1] + 2 * c[2]);
}

View make_view(double *p, const int64_t d[3])
{
    View v;
    v.p = p;
    for (int a = 0; a < 3; a++) v.d[a] = d[a];
    v.sj = mgk::pitch_for(d[2]);
    v.si = v.sj * d[1];
    v.i0 = 0;
    return v;
}

int alloc_view(View *v, const int64_t d[3])
{
    *v = make_view(nullptr, d);
    const size_t bytes = (size_t)v->elems() * sizeof(double);
    HIP_TRY(hipMalloc((void **)&v->p, bytes));
    HIP_TRY(hipMemset(v->p, 0, bytes));
    return MG_OK;
}

struct Level {
    int64_t n[3];   // level size n_0 >> l: bounds every loop on this level
    Coef c;
    int nu;
    View R;         // dims n
    View eps;       // dims n+1 (l >= 1); unused on l == 0
    View eps_alt;   // ping-pong twin of eps for the out-of-place fused sweep
};

}  // namespace

// A handle created with n_gpus > 1: one slab-partitioned rank per GPU, each driven by a host
// thread of its own that lives as long as the handle (the calls are collective: every rank must be
// inside the call for any of them to finish).
struct MultiGpu {
    std::vector<mg_dist *> ranks;
    mg_local_group *group = nullptr;  // in-process transport (an ordinal repeats in devices[])

    std::vector<std::thread> workers;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    std::function<int(int)> job;
    uint64_t job_id = 0;
    int pending = 0;
    bool quit = false;
    std::vector<int> rc;
    std::vector<std::string> err;

    // Starts one worker per rank; false when the threads cannot be created.
    bool start(int n);
    // Runs f(rank) on every worker and waits for all of them.
    void run(const std::function<int(int)> &f);
    void stop();
    // A rank's call failed: release the others (they may be waiting for its planes).
    void give_up()
    {
        std::vector<mg_dist *> all;
        {
            std::lock_guard<std::mutex> lock(m);
            all = ranks;
        }
        for (mg_dist *d : all)
            if (d) mg_dist_abort(d);
    }
};

struct mg_handle {
    MultiGpu *multi = nullptr;
    mg_config cfg;
    int L;
    Level lv[MG_MAX_LEVELS];
    View phi, b;            // fine potential and rho/EPS_0
    View phi_alt;           // ping-pong twin of phi for the out-of-place fused sweep
    double *ef = nullptr;       // E field, dense [i][j][k][3]; allocated by the first mg_compute_ef
    bool have_ef = false;
    double *partials = nullptr;
    int64_t n_partials = 0;
    double *h_sum = nullptr;    // sum of R_h^2: pinned host memory the reduction kernel writes directly
    double *d_sum = nullptr;    // the same memory as the device sees it
    hipStream_t stream = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    bool have_rho = false, have_phi = false;
    bool fused_residual = true;  // R_0 never stored (default); false with MG_FLAG_KEEP_FINE_RESIDUAL
    bool restrict7 = false;      // MG_FLAG_RESTRICT_7PT: the "B" classes' restriction
    bool correction_pending = false;  // phi -= P(eps_1) of the last iteration not applied yet
    double last_ms = 0;
    // profiling
    bool profiling = false;
    struct Rec { int kind; hipEvent_t a, b; double bytes; int launches; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> ev_pool;
    mg_timers timers;
};

namespace {

int64_t count3(const int64_t d[3]) { return d[0] * d[1] * d[2]; }

struct Prof {
    mg_handle *h;
    int kind;
    double bytes;
    int launches = 1;  // kernels launched inside this scope
    hipEvent_t a = nullptr, b = nullptr;
    Prof(mg_handle *h_, int kind_, double bytes_) : h(h_), kind(kind_), bytes(bytes_)
    {
        if (!h->profiling) return;
        a = take();
        b = take();
        hipEventRecord(a, h->stream);
    }
    ~Prof()
    {
        if (!h->profiling) return;
        hipEventRecord(b, h->stream);
        h->recs.push_back({kind, a, b, bytes, launches});
    }
    hipEvent_t take()
    {
        if (!h->ev_pool.empty()) {
            hipEvent_t e = h->ev_pool.back();
            h->ev_pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        hipEventCreate(&e);
        return e;
    }
};

void drain_profile(mg_handle *h)
{
    if (h->recs.empty()) return;
    hipStreamSynchronize(h->stream);
    for (auto &r : h->recs) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            h->timers.ms[r.kind] += ms;
            h->timers.launches[r.kind] += (uint64_t)r.launches;
            h->timers.bytes[r.kind] += r.bytes;
        }
        h->ev_pool.push_back(r.a);
        h->ev_pool.push_back(r.b);
    }
    h->recs.clear();
}

// Step 6) as its own launch: phi -= P(eps_1).
void flush_correction(mg_handle *h)
{
    if (!h->correction_pending) return;
    Prof p(h, MG_T_PROLONG_CORRECT, 17.0 * (double)count3(h->lv[0].n));
    mgk::prolong_correct(h->stream, h->lv[1].eps, h->phi, h->lv[0].n, IRange{0, h->lv[0].n[0]});
    h->correction_pending = false;
}

// Step 6) of the previous iteration, if still pending, and steps 1)-2b) of this one: correct,
// smooth the fine grid, residual, norm.  The correction phi -= P(eps_1) is folded into the first
// fine sweep where the single-pass sweep kernel runs (one pass over phi less per iteration).
// Leaves sum(R_h^2) in *h->h_sum once the stream has drained (the reduction kernel stores it
// straight into pinned host memory: no copy launch).
void enqueue_smooth_residual(mg_handle *h)
{
    Level &f = h->lv[0];
    const double N = (double)count3(f.n);
    const IRange all{0, f.n[0]};
    if (h->correction_pending && f.nu > 0) {
        Prof p(h, MG_T_SMOOTH_FINE, (32.0 * f.nu + 17.0) * N);
        p.launches = mgk::prolong_smooth_rb(h->stream, h->lv[1].eps, true, &h->phi, &h->phi_alt, h->b, f.n, f.c,
                                            h->cfg.omega, f.nu, all);
        h->correction_pending = false;
    } else {
        flush_correction(h);
        Prof p(h, MG_T_SMOOTH_FINE, 32.0 * N * f.nu);
        p.launches = mgk::smooth_rb(h->stream, &h->phi, &h->phi_alt, h->b, f.n, f.c, h->cfg.omega, f.nu, all);
    }
    int64_t np;
    if (h->fused_residual) {
        // residual, sum of squares and restriction to level 1 in one pass; R_0 stays on chip
        Prof p(h, MG_T_RESIDUAL, 17.0 * N);
        View none = f.R;
        none.p = nullptr;
        mgk::residual_restrict(h->stream, h->phi, h->b, none, h->lv[1].R, f.n, f.c, h->partials, all);
        np = mgk::residual_restrict_partials(f.n, all);
    } else {
        Prof p(h, MG_T_RESIDUAL, 24.0 * N);
        mgk::residual(h->stream, h->phi, h->b, f.R, f.n, f.c, h->partials, all);
        np = mgk::residual_partials(f.n, all);
    }
    {
        Prof p(h, MG_T_NORM, 8.0 * (double)np);
        mgk::sum_partials(h->stream, h->partials, np, h->d_sum);
    }
}

// Steps 3)-5): restrict down, (prolong, smooth) up.  Step 6), phi -= P(eps_1), is left pending:
// the next iteration's first sweep applies it, or flush_correction() when no iteration follows.
void enqueue_coarse_correction(mg_handle *h)
{
    const int L = h->L;
    for (int l = h->fused_residual ? 1 : 0; l < L; l++) {
        Prof p(h, MG_T_RESTRICT, 9.0 * (double)count3(h->lv[l].n));
        if (h->restrict7)
            mgk::restrict7(h->stream, h->lv[l].R, h->lv[l + 1].R, h->lv[l].n, IRange{0, h->lv[l + 1].n[0]});
        else
            mgk::restrict27(h->stream, h->lv[l].R, h->lv[l + 1].R, h->lv[l].n, IRange{0, h->lv[l + 1].n[0]});
    }
    for (int l = L; l >= 1; l--) {
        Level &v = h->lv[l];
        const IRange all{0, v.eps.d[0]};
        const double Nl = (double)count3(v.n);
        if (l < L) {
            // prolongation of eps_{l+1} as the initial guess, then nu_l sweeps
            Prof p(h, MG_T_SMOOTH_COARSE, (32.0 * v.nu + 9.0) * Nl);
            // eps_l is zero outside 0..n_l-2 (nothing ever writes there), so the fused form need not read it
            p.launches = mgk::prolong_smooth_rb(h->stream, h->lv[l + 1].eps, false, &v.eps, &v.eps_alt, v.R, v.n,
                                                v.c, h->cfg.omega, v.nu, all, IRange{0, 0}, true);
        } else {
            Prof p(h, MG_T_SMOOTH_COARSE, 32.0 * Nl * v.nu);
            p.launches = mgk::smooth_rb(h->stream, &v.eps, &v.eps_alt, v.R, v.n, v.c, h->cfg.omega, v.nu, all);
        }
    }
    h->correction_pending = true;
}

double l2_from_sum(const mg_handle *h, double sum)
{
    // sqrt(sum/(ni*nj*nk)) -- reference cpp/PotentialSolver.cpp:3818, with a 64-bit count
    return std::sqrt(sum / (double)count3(h->lv[0].n));
}

}  // namespace

bool MultiGpu::start(int n)
{
    rc.assign(n, 0);
    err.assign(n, std::string());
    try {
        for (int r = 0; r < n; r++)
            workers.emplace_back([this, r] {
                uint64_t seen = 0;
                for (;;) {
                    std::function<int(int)> mine;
                    {
                        std::unique_lock<std::mutex> lock(m);
                        cv_job.wait(lock, [&] { return quit || job_id != seen; });
                        if (quit) return;
                        seen = job_id;
                        mine = job;
                    }
                    const int code = mine(r);
                    if (code) give_up();  // before reporting: the others may be waiting for this rank
                    std::lock_guard<std::mutex> lock(m);
                    rc[r] = code;
                    if (code) err[r] = g_err;  // g_err is per thread
                    if (--pending == 0) cv_done.notify_all();
                }
            });
    } catch (...) {  // std::system_error: no more threads
        stop();
        return false;
    }
    return true;
}

void MultiGpu::run(const std::function<int(int)> &f)
{
    std::unique_lock<std::mutex> lock(m);
    job = f;
    job_id++;
    pending = (int)workers.size();
    for (auto &c : rc) c = 0;
    cv_job.notify_all();
    cv_done.wait(lock, [&] { return pending == 0; });
    job = nullptr;
}

void MultiGpu::stop()
{
    {
        std::lock_guard<std::mutex> lock(m);
        quit = true;
    }
    cv_job.notify_all();
    for (auto &t : workers)
        if (t.joinable()) t.join();
    workers.clear();
}

namespace {

// Runs f(rank) for every rank of a multi-GPU handle, each on the rank's own host thread.
template <typename F>
int on_all_ranks(MultiGpu *m, F f)
{
    m->run(std::function<int(int)>(f));
    for (size_t r = 0; r < m->rc.size(); r++)
        if (m->rc[r]) return fail(m->rc[r], "rank %d: %s", (int)r, m->err[r].c_str());
    return MG_OK;
}

int single_gpu_only(const mg_handle *h, const char *what)
{
    if (h && h->multi) return fail(MG_ERR_STATE, "%s is not available on a multi-GPU handle", what);
    return MG_OK;
}

int create_multi(const mg_config *cfg, mg_handle **out)
{
    const int n = cfg->n_gpus;
    if (n > MG_MAX_GPUS) return fail(MG_ERR_INVALID, "n_gpus %d exceeds %d", n, MG_MAX_GPUS);
    if (cfg->depth < 1) return fail(MG_ERR_INVALID, "a multi-GPU handle needs at least one coarse grid");
    if (cfg->flags & ~MG_FLAG_FAST_SMOOTHER)
        return fail(MG_ERR_INVALID, "a multi-GPU handle supports no MG_FLAG_* option but MG_FLAG_FAST_SMOOTHER");
    bool distinct = true;
    for (int a = 0; a < n; a++)
        for (int b = a + 1; b < n; b++)
            if (cfg->devices[a] == cfg->devices[b]) distinct = false;
    mg_handle *h = new (std::nothrow) mg_handle();
    if (!h) return fail(MG_ERR_NOMEM, "host allocation failed");
    h->cfg = *cfg;
    h->L = cfg->depth;
    h->multi = new (std::nothrow) MultiGpu();
    if (!h->multi) {
        delete h;
        return fail(MG_ERR_NOMEM, "host allocation failed");
    }
    MultiGpu *m = h->multi;
    m->ranks.assign(n, nullptr);
    if (!m->start(n)) {
        delete h->multi;
        delete h;
        return fail(MG_ERR_NOMEM, "could not start %d host threads for the ranks", n);
    }
    mg_dist_config dc;
    memset(&dc, 0, sizeof(dc));
    dc.base = *cfg;
    dc.base.n_gpus = 0;
    dc.world = n;
    dc.coarse_cut = cfg->coarse_cut;
    int rc = MG_OK;
    if (distinct) {
        dc.transport = MG_TRANSPORT_RCCL;
        rc = mg_dist_unique_id(dc.unique_id);
    } else {
        dc.transport = MG_TRANSPORT_LOCAL;
        rc = mg_local_group_create(n, 0, &m->group);
        dc.local_group = m->group;
    }
    if (!rc)
        rc = on_all_ranks(m, [&](int r) {
            mg_dist_config mine = dc;
            mine.rank = r;
            mine.base.device = cfg->devices[r];
            mg_dist *d = nullptr;
            const int code = mg_dist_create(&mine, &d);
            std::lock_guard<std::mutex> lock(m->m);
            m->ranks[r] = d;
            return code;
        });
    if (rc) {
        mg_destroy(h);
        return rc;
    }
    *out = h;
    return MG_OK;
}

// dense global array <-> the ranks' slabs
int multi_upload(mg_handle *h, const double *flat, bool rho)
{
    const size_t plane = (size_t)h->cfg.n[1] * (size_t)h->cfg.n[2];
    return on_all_ranks(h->multi, [&](int r) {
        int64_t sp[2], sr[2];
        int rc = mg_dist_plan(h->multi->ranks[r], nullptr, nullptr, sp, sr);
        if (rc) return rc;
        const int64_t *st = rho ? sr : sp;
        return rho ? mg_dist_set_rho_planes(h->multi->ranks[r], st[0], st[1], flat + (size_t)st[0] * plane)
                   : mg_dist_set_phi_planes(h->multi->ranks[r], st[0], st[1], flat + (size_t)st[0] * plane);
    });
}

int check_ready(mg_handle *h, bool multigrid = true)
{
    if (!h) return fail(MG_ERR_INVALID, "null handle");
    if (multigrid && h->L < 1) return fail(MG_ERR_STATE, "the handle was created with depth 0 (no coarse grids)");
    if (!h->have_rho || !h->have_phi)
        return fail(MG_ERR_STATE, "mg_set_rho and mg_set_phi must be called before solving");
    HIP_TRY(hipSetDevice(h->cfg.device));
    return MG_OK;
}

// dense host array (row length d[2]) <-> pitched device array
int h2d(View dst, const double *flat, hipStream_t s)
{
    const size_t row = (size_t)dst.d[2] * sizeof(double);
    HIP_TRY(hipMemcpy2DAsync(dst.p, (size_t)dst.sj * sizeof(double), flat, row, row,
                             (size_t)(dst.d[0] * dst.d[1]), hipMemcpyHostToDevice, s));
    return MG_OK;
}

int d2h(double *flat, View src, hipStream_t s)
{
    const size_t row = (size_t)src.d[2] * sizeof(double);
    HIP_TRY(hipMemcpy2DAsync(flat, row, src.p, (size_t)src.sj * sizeof(double), row,
                             (size_t)(src.d[0] * src.d[1]), hipMemcpyDeviceToHost, s));
    return MG_OK;
}

// Whole fine-grid fields travel as ONE dense copy (a row-by-row 2-D copy of 4 KB rows reaches
// well under 1 GB/s): when the device rows are padded, the dense image lands in the idle twin of
// phi and a kernel spreads it to the row pitch, dividing rho by EPS_0 on the way.
int upload(mg_handle *h, const double *flat, View dst, bool div_eps0)
{
    const size_t bytes = (size_t)count3(dst.d) * sizeof(double);
    if (dst.sj == dst.d[2]) {
        HIP_TRY(hipMemcpyAsync(dst.p, flat, bytes, hipMemcpyHostToDevice, h->stream));
        if (div_eps0) mgk::div_eps0(h->stream, dst);
    } else {
        HIP_TRY(hipMemcpyAsync(h->phi_alt.p, flat, bytes, hipMemcpyHostToDevice, h->stream));
        mgk::repitch_in(h->stream, dst, h->phi_alt.p, div_eps0);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MG_OK;
}

int download(mg_handle *h, double *flat, View src)
{
    const size_t bytes = (size_t)count3(src.d) * sizeof(double);
    if (src.sj == src.d[2]) {
        HIP_TRY(hipMemcpyAsync(flat, src.p, bytes, hipMemcpyDeviceToHost, h->stream));
    } else {
        mgk::repitch_out(h->stream, h->phi_alt.p, src);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(flat, h->phi_alt.p, bytes, hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MG_OK;
}

}  // namespace

extern "C" {

int mg_abi_version(void) { return MG_ABI_VERSION; }

const char *mg_last_error(void) { return g_err; }

int mg_set_exact_division(int on)
{
    mgk::set_exact_division(on != 0);
    return MG_OK;
}

int mg_device_count(void)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

int mg_default_config_b(mg_config *cfg, int64_t ni, int64_t nj, int64_t nk, const double dh[3], int depth)
{
    int rc = mg_default_config(cfg, ni, nj, nk, dh, depth);
    if (rc) return rc;
    if (depth < 1) return fail(MG_ERR_INVALID, "the B solvers need at least one coarse grid");
    cfg->flags |= MG_FLAG_RESTRICT_7PT;
    if (depth >= 5)  // 1, 2, 3, 4, 5, 50 (cpp/PotentialSolver.cpp:1928-1939); deeper: 5 on the added levels
        for (int l = 1; l < depth; l++) cfg->nu[l] = l + 1 < 5 ? l + 1 : 5;
    return MG_OK;
}

int mg_default_config(mg_config *cfg, int64_t ni, int64_t nj, int64_t nk, const double dh[3], int depth)
{
    if (!cfg || !dh) return fail(MG_ERR_INVALID, "null argument");
    if (depth < 0 || depth >= MG_MAX_LEVELS) return fail(MG_ERR_INVALID, "depth %d out of range 0..%d", depth, MG_MAX_LEVELS - 1);
    memset(cfg, 0, sizeof(*cfg));
    cfg->n[0] = ni;
    cfg->n[1] = nj;
    cfg->n[2] = nk;
    for (int a = 0; a < 3; a++) cfg->dh[a] = dh[a];
    cfg->depth = depth;
    cfg->omega = 1.4;
    // fine sweeps: 3 for depth 1..4, 1 for depth >= 5; 5 on intermediate levels,
    // 50 on the coarsest (reference cpp/PotentialSolver.cpp:2422-2425 ... :3759-3770)
    cfg->nu[0] = depth >= 5 ? 1 : 3;
    for (int l = 1; l < depth; l++) cfg->nu[l] = 5;
    if (depth >= 1) cfg->nu[depth] = 50;
    return MG_OK;
}

int mg_create(const mg_config *cfg, mg_handle **out)
{
    if (!cfg || !out) return fail(MG_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->n_gpus < 0) return fail(MG_ERR_INVALID, "n_gpus %d is negative", cfg->n_gpus);
    if (cfg->n_gpus > 1) return create_multi(cfg, out);
    const int L = cfg->depth;
    if (L < 0 || L >= MG_MAX_LEVELS) return fail(MG_ERR_INVALID, "depth %d out of range 0..%d", L, MG_MAX_LEVELS - 1);
    for (int a = 0; a < 3; a++) {
        if (cfg->n[a] < 3 || cfg->n[a] > kMaxDim) return fail(MG_ERR_INVALID, "n[%d]=%lld out of range 3..%lld", a, (long long)cfg->n[a], (long long)kMaxDim);
        if ((cfg->n[a] >> L) < 3) return fail(MG_ERR_INVALID, "n[%d]=%lld too small for depth %d (coarsest level needs >= 3 nodes)", a, (long long)cfg->n[a], L);
        if (!(cfg->dh[a] > 0)) return fail(MG_ERR_INVALID, "dh[%d] must be positive", a);
    }
    for (int l = 0; l <= L; l++)
        if (cfg->nu[l] < 0 || cfg->nu[l] > 100000) return fail(MG_ERR_INVALID, "nu[%d]=%d out of range", l, cfg->nu[l]);
    if (cfg->flags & ~(MG_FLAG_KEEP_FINE_RESIDUAL | MG_FLAG_RESTRICT_7PT | MG_FLAG_RESERVE_EF | MG_FLAG_FAST_SMOOTHER))
        return fail(MG_ERR_INVALID, "unknown flag bits 0x%x", (unsigned)cfg->flags);
    int rc = select_device(cfg->device);
    if (rc) return rc;

    mg_handle *h = new (std::nothrow) mg_handle();
    if (!h) return fail(MG_ERR_NOMEM, "host allocation failed");
    h->cfg = *cfg;
    h->L = L;
    h->restrict7 = (cfg->flags & MG_FLAG_RESTRICT_7PT) != 0;
    // the fused residual+restriction kernel implements the 27-point weights only
    h->fused_residual = L >= 1 && !(cfg->flags & MG_FLAG_KEEP_FINE_RESIDUAL) && !h->restrict7;
    memset(&h->timers, 0, sizeof(h->timers));
    memset(h->lv, 0, sizeof(h->lv));
    // out of device memory: say how much this configuration needs in all
    auto out_of_memory = [&]() -> int {
        uint64_t need = 0, free_b = 0, total_b = 0;
        size_t f = 0, t = 0;
        if (mg_device_bytes(cfg, &need) != MG_OK) return MG_ERR_NOMEM;
        if (hipMemGetInfo(&f, &t) == hipSuccess) {
            free_b = f;
            total_b = t;
        }
        return fail(MG_ERR_NOMEM, "out of device memory: this configuration needs %.2f GB on device %d (%.2f of %.2f GB were free "
                    "when the allocation failed); mg_device_bytes gives the need of a configuration, n_gpus splits the grid over several GPUs",
                    need / 1e9, cfg->device, free_b / 1e9, total_b / 1e9);
    };
#define CREATE_TRY(expr)                              \
    do {                                              \
        int rc_ = (expr);                             \
        if (rc_) {                                    \
            if (rc_ == MG_ERR_NOMEM) out_of_memory(); \
            mg_destroy(h);                            \
            (void)hipGetLastError(); /* reported; not left behind for the next call's check */ \
            return rc_;                               \
        }                                             \
    } while (0)
    auto hip_try = [](hipError_t e, const char *what) -> int {
        if (e == hipSuccess) return MG_OK;
        (void)hipGetLastError();  // reported here; not left behind for a later call's check
        return fail(e == hipErrorOutOfMemory ? MG_ERR_NOMEM : MG_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    };
    CREATE_TRY(hip_try(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking), "hipStreamCreate"));
    CREATE_TRY(hip_try(hipEventCreate(&h->ev_a), "hipEventCreate"));
    CREATE_TRY(hip_try(hipEventCreate(&h->ev_b), "hipEventCreate"));
    for (int l = 0; l <= L; l++) {
        Level &v = h->lv[l];
        int64_t en[3];
        for (int a = 0; a < 3; a++) {
            v.n[a] = cfg->n[a] >> l;
            en[a] = v.n[a] + 1;
        }
        v.c = level_coef(cfg->dh, l);
        // coarse levels only: their sweeps are bound by instruction issue; on the fine grid, which is
        // bound by HBM, the short form measured slower (fused sweep 712 against 669 us at 513^3)
        if ((cfg->flags & MG_FLAG_FAST_SMOOTHER) && l >= 1) v.c = mgk::with_fast_smoother(v.c, cfg->omega);
        v.nu = cfg->nu[l];
        if (l >= 1 || !h->fused_residual) CREATE_TRY(alloc_view(&v.R, v.n));
        else v.R = make_view(nullptr, v.n);
        if (l >= 1) {
            CREATE_TRY(alloc_view(&v.eps, en));
            CREATE_TRY(alloc_view(&v.eps_alt, en));
        }
    }
    CREATE_TRY(alloc_view(&h->phi, h->lv[0].n));
    CREATE_TRY(alloc_view(&h->phi_alt, h->lv[0].n));
    CREATE_TRY(alloc_view(&h->b, h->lv[0].n));
    {
        const IRange all{0, h->lv[0].n[0]};
        h->n_partials = h->fused_residual ? mgk::residual_restrict_partials(h->lv[0].n, all)
                                          : mgk::residual_partials(h->lv[0].n, all);
        const int64_t gs = mgk::gs_residual_partials(h->lv[0].n, all);
        if (gs > h->n_partials) h->n_partials = gs;
    }
    CREATE_TRY(hip_try(hipMalloc((void **)&h->partials, (size_t)h->n_partials * sizeof(double)), "hipMalloc(partials)"));
    if (cfg->flags & MG_FLAG_RESERVE_EF)
        CREATE_TRY(hip_try(hipMalloc((void **)&h->ef, (size_t)count3(h->lv[0].n) * 3 * sizeof(double)), "hipMalloc(ef)"));
    CREATE_TRY(hip_try(hipHostMalloc((void **)&h->h_sum, sizeof(double), hipHostMallocMapped), "hipHostMalloc"));
    *h->h_sum = 0;
    CREATE_TRY(hip_try(hipHostGetDevicePointer((void **)&h->d_sum, h->h_sum, 0), "hipHostGetDevicePointer"));
#undef CREATE_TRY
    *out = h;
    return MG_OK;
}

int mg_device_bytes(const mg_config *cfg, uint64_t *bytes)
{
    if (!cfg || !bytes) return fail(MG_ERR_INVALID, "null argument");
    const int L = cfg->depth;
    if (L < 0 || L >= MG_MAX_LEVELS) return fail(MG_ERR_INVALID, "depth %d out of range 0..%d", L, MG_MAX_LEVELS - 1);
    for (int a = 0; a < 3; a++)
        if (cfg->n[a] < 3 || cfg->n[a] > kMaxDim || (cfg->n[a] >> L) < 3)
            return fail(MG_ERR_INVALID, "n[%d]=%lld out of range for depth %d", a, (long long)cfg->n[a], L);
    if (cfg->n_gpus > 1) {
        if (cfg->n_gpus > MG_MAX_GPUS) return fail(MG_ERR_INVALID, "n_gpus %d exceeds %d", cfg->n_gpus, MG_MAX_GPUS);
        const long long n[3] = {cfg->n[0], cfg->n[1], cfg->n[2]};
        long long most = 0;
        for (int r = 0; r < cfg->n_gpus; r++) {
            const long long b = mgk::dist_rank_device_bytes(n, L, cfg->n_gpus, cfg->coarse_cut, r, cfg->nu);
            if (b < 0) return MG_ERR_INVALID;  // the plan's message stands
            if (b > most) most = b;
        }
        *bytes = (uint64_t)most;
        return MG_OK;
    }
    // the allocations of mg_create, as arithmetic (keep the two lists alike)
    auto view_bytes = [](const int64_t d[3]) { return (uint64_t)make_view(nullptr, d).elems() * sizeof(double); };
    const bool restrict7 = (cfg->flags & MG_FLAG_RESTRICT_7PT) != 0;
    const bool fused = L >= 1 && !(cfg->flags & MG_FLAG_KEEP_FINE_RESIDUAL) && !restrict7;
    uint64_t sum = 0;
    int64_t n0[3] = {cfg->n[0], cfg->n[1], cfg->n[2]};
    for (int l = 0; l <= L; l++) {
        int64_t n[3], en[3];
        for (int a = 0; a < 3; a++) {
            n[a] = cfg->n[a] >> l;
            en[a] = n[a] + 1;
        }
        if (l >= 1 || !fused) sum += view_bytes(n);
        if (l >= 1) sum += 2 * view_bytes(en);
    }
    sum += 3 * view_bytes(n0);
    const IRange all{0, n0[0]};
    int64_t np = fused ? mgk::residual_restrict_partials(n0, all) : mgk::residual_partials(n0, all);
    const int64_t gs = mgk::gs_residual_partials(n0, all);
    if (gs > np) np = gs;
    sum += (uint64_t)np * sizeof(double);
    if (cfg->flags & MG_FLAG_RESERVE_EF) sum += (uint64_t)count3(n0) * 3 * sizeof(double);
    *bytes = sum;
    return MG_OK;
}

int mg_destroy(mg_handle *h)
{
    if (!h) return MG_OK;
    if (h->multi) {
        // collective like creation: the RCCL communicators are torn down together
        on_all_ranks(h->multi, [&](int r) { return mg_dist_destroy(h->multi->ranks[r]); });
        h->multi->stop();
        mg_local_group_destroy(h->multi->group);
        delete h->multi;
        delete h;
        return MG_OK;
    }
    hipSetDevice(h->cfg.device);
    if (h->stream) hipStreamSynchronize(h->stream);
    for (auto &r : h->recs) { hipEventDestroy(r.a); hipEventDestroy(r.b); }
    for (auto e : h->ev_pool) hipEventDestroy(e);
    for (int l = 0; l < MG_MAX_LEVELS; l++) {
        if (h->lv[l].R.p) hipFree(h->lv[l].R.p);
        if (h->lv[l].eps.p) hipFree(h->lv[l].eps.p);
        if (h->lv[l].eps_alt.p) hipFree(h->lv[l].eps_alt.p);
    }
    if (h->phi.p) hipFree(h->phi.p);
    if (h->phi_alt.p) hipFree(h->phi_alt.p);
    if (h->b.p) hipFree(h->b.p);
    if (h->ef) hipFree(h->ef);
    if (h->partials) hipFree(h->partials);
    if (h->h_sum) hipHostFree(h->h_sum);
    if (h->ev_a) hipEventDestroy(h->ev_a);
    if (h->ev_b) hipEventDestroy(h->ev_b);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return MG_OK;
}

int mg_set_rho(mg_handle *h, const double *rho_flat)
{
    if (!h || !rho_flat) return fail(MG_ERR_INVALID, "null argument");
    if (h->multi) {
        int rc = multi_upload(h, rho_flat, true);
        if (!rc) h->have_rho = true;
        return rc;
    }
    HIP_TRY(hipSetDevice(h->cfg.device));
    int rc = upload(h, rho_flat, h->b, true);
    if (rc) return rc;
    h->have_rho = true;
    return MG_OK;
}

int mg_set_phi(mg_handle *h, const double *phi_flat)
{
    if (!h || !phi_flat) return fail(MG_ERR_INVALID, "null argument");
    if (h->multi) {
        int rc = multi_upload(h, phi_flat, false);
        if (!rc) h->have_phi = true;
        h->have_ef = false;
        return rc;
    }
    HIP_TRY(hipSetDevice(h->cfg.device));
    int rc = upload(h, phi_flat, h->phi, false);
    if (rc) return rc;
    h->have_phi = true;
    h->have_ef = false;
    return MG_OK;
}

int mg_get_phi(mg_handle *h, double *phi_flat)
{
    if (!h || !phi_flat) return fail(MG_ERR_INVALID, "null argument");
    if (!h->have_phi) return fail(MG_ERR_STATE, "no phi on the device yet");
    if (h->multi) {
        const size_t plane = (size_t)h->cfg.n[1] * (size_t)h->cfg.n[2];
        return on_all_ranks(h->multi, [&](int r) {
            int64_t own[2];
            int rc = mg_dist_plan(h->multi->ranks[r], nullptr, own, nullptr, nullptr);
            if (rc) return rc;
            return mg_dist_get_phi_planes(h->multi->ranks[r], own[0], own[1], phi_flat + (size_t)own[0] * plane);
        });
    }
    HIP_TRY(hipSetDevice(h->cfg.device));
    return download(h, phi_flat, h->phi);
}

int mg_solve(mg_handle *h, unsigned max_it, double tol, int *converged, unsigned *cycles, double *L2)
{
    if (h && h->multi) {
        if (!h->have_rho || !h->have_phi) return fail(MG_ERR_STATE, "mg_set_rho and mg_set_phi must be called before solving");
        h->have_ef = false;
        const int n = (int)h->multi->ranks.size();
        std::vector<int> conv(n, 0);
        std::vector<unsigned> cyc(n, 0);
        std::vector<double> l2(n, 0.0);
        int rc = on_all_ranks(h->multi, [&](int r) {
            return mg_dist_solve(h->multi->ranks[r], max_it, tol, &conv[r], &cyc[r], &l2[r]);
        });
        if (rc) return rc;
        if (converged) *converged = conv[0];  // the decision is taken on the all-reduced norm: equal on all ranks
        if (cycles) *cycles = cyc[0];
        if (L2) *L2 = l2[0];
        return MG_OK;
    }
    int rc = check_ready(h);
    if (rc) return rc;
    bool conv = false;
    double l2 = 0;
    unsigned it = 0;
    h->have_ef = false;
    HIP_TRY(hipEventRecord(h->ev_a, h->stream));
    for (it = 0; it < max_it; it++) {
        enqueue_smooth_residual(h);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(h->stream));
        drain_profile(h);  // the stream is idle: fold the event pairs recorded so far into the timers
        l2 = l2_from_sum(h, *h->h_sum);
        if (l2 < tol) {
            conv = true;
            it++;
            break;
        }
        enqueue_coarse_correction(h);
        HIP_TRY(hipGetLastError());
    }
    flush_correction(h);  // max_it exhausted: the last iteration's correction still counts
    HIP_TRY(hipEventRecord(h->ev_b, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
    h->last_ms = ms;
    if (converged) *converged = conv ? 1 : 0;
    if (cycles) *cycles = it;
    if (L2) *L2 = l2;
    return MG_OK;
}

int mg_vcycles(mg_handle *h, unsigned count, double *L2)
{
    if (h && h->multi) {
        if (!h->have_rho || !h->have_phi) return fail(MG_ERR_STATE, "mg_set_rho and mg_set_phi must be called before solving");
        h->have_ef = false;
        std::vector<double> l2(h->multi->ranks.size(), 0.0);
        int rc = on_all_ranks(h->multi, [&](int r) { return mg_dist_vcycles(h->multi->ranks[r], count, &l2[r]); });
        if (rc) return rc;
        if (L2) *L2 = l2[0];
        return MG_OK;
    }
    int rc = check_ready(h);
    if (rc) return rc;
    h->have_ef = false;
    HIP_TRY(hipEventRecord(h->ev_a, h->stream));
    for (unsigned it = 0; it < count; it++) {
        enqueue_smooth_residual(h);
        enqueue_coarse_correction(h);
    }
    flush_correction(h);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev_b, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
    h->last_ms = ms;
    if (L2) *L2 = count ? l2_from_sum(h, *h->h_sum) : 0.0;
    return MG_OK;
}

int mg_vcycle(mg_handle *h, double *L2) { return mg_vcycles(h, 1, L2); }

int mg_gs_solve(mg_handle *h, unsigned max_it, double tol, int *converged, unsigned *iterations, double *L2)
{
    int rc = single_gpu_only(h, "mg_gs_solve");
    if (rc) return rc;
    rc = check_ready(h, false);
    if (rc) return rc;
    Level &f = h->lv[0];
    const double N = (double)count3(f.n);
    const IRange all{0, f.n[0]};
    const int64_t np = mgk::gs_residual_partials(f.n, all);
    bool conv = false;
    double l2 = 0;
    unsigned it = 0;
    h->have_ef = false;
    HIP_TRY(hipEventRecord(h->ev_a, h->stream));
    // The reference sweeps once per iteration and tests the residual when it % 25 == 0
    // (cpp/PotentialSolver.cpp:268).  The sweeps up to the next test are issued as one batch (the
    // sweep kernels then take them two per launch); `it` is the last iteration of the batch.
    for (it = 0; it < max_it; it++) {
        unsigned batch = it % 25 == 0 ? 1 : 25 - it % 25 + 1;  // through the next iteration with it % 25 == 0
        if (batch > max_it - it) batch = max_it - it;
        {
            Prof p(h, MG_T_SMOOTH_FINE, 32.0 * N * batch);
            p.launches = mgk::smooth_rb(h->stream, &h->phi, &h->phi_alt, h->b, f.n, f.c, h->cfg.omega, (int)batch, all);
        }
        it += batch - 1;
        if (it % 25 == 0) {
            {
                Prof p(h, MG_T_RESIDUAL, 16.0 * N);
                mgk::gs_residual_sumsq(h->stream, h->phi, h->b, f.n, f.c, h->partials, all);
            }
            {
                Prof p(h, MG_T_NORM, 8.0 * (double)np);
                mgk::sum_partials(h->stream, h->partials, np, h->d_sum);
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(h->stream));
            drain_profile(h);
            l2 = l2_from_sum(h, *h->h_sum);
            if (l2 < tol) {
                conv = true;
                it++;
                break;
            }
        }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev_b, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
    h->last_ms = ms;
    if (converged) *converged = conv ? 1 : 0;
    if (iterations) *iterations = it;
    if (L2) *L2 = l2;
    return MG_OK;
}

int mg_compute_ef(mg_handle *h)
{
    if (!h) return fail(MG_ERR_INVALID, "null handle");
    if (h->multi) {
        if (!h->have_phi) return fail(MG_ERR_STATE, "no phi on the device yet");
        int rc = on_all_ranks(h->multi, [&](int r) { return mg_dist_compute_ef(h->multi->ranks[r]); });
        if (!rc) h->have_ef = true;
        return rc;
    }
    if (!h->have_phi) return fail(MG_ERR_STATE, "no phi on the device yet");
    HIP_TRY(hipSetDevice(h->cfg.device));
    const int64_t *n = h->lv[0].n;
    if (!h->ef) HIP_TRY(hipMalloc((void **)&h->ef, (size_t)count3(n) * 3 * sizeof(double)));
    flush_correction(h);
    HIP_TRY(hipEventRecord(h->ev_a, h->stream));
    {
        Prof p(h, MG_T_EF, 32.0 * (double)count3(n));
        mgk::compute_ef(h->stream, h->phi, h->ef, 0, n, h->cfg.dh, IRange{0, n[0]});
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev_b, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev_a, h->ev_b));
    h->last_ms = ms;
    h->have_ef = true;
    return MG_OK;
}

int mg_get_ef(mg_handle *h, double *ef_flat)
{
    if (!h || !ef_flat) return fail(MG_ERR_INVALID, "null argument");
    if (!h->have_ef) return fail(MG_ERR_STATE, "mg_compute_ef has not run on the current phi");
    if (h->multi) {
        const size_t plane = (size_t)h->cfg.n[1] * (size_t)h->cfg.n[2] * 3;
        return on_all_ranks(h->multi, [&](int r) {
            int64_t own[2];
            int rc = mg_dist_plan(h->multi->ranks[r], nullptr, own, nullptr, nullptr);
            if (rc) return rc;
            return mg_dist_get_ef_planes(h->multi->ranks[r], own[0], own[1], ef_flat + (size_t)own[0] * plane);
        });
    }
    HIP_TRY(hipSetDevice(h->cfg.device));
    HIP_TRY(hipMemcpyAsync(ef_flat, h->ef, (size_t)count3(h->lv[0].n) * 3 * sizeof(double), hipMemcpyDeviceToHost,
                           h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MG_OK;
}

int mg_host_register(void *ptr, uint64_t bytes)
{
    if (!ptr || !bytes) return fail(MG_ERR_INVALID, "mg_host_register: null pointer or zero size");
    const hipError_t e = hipHostRegister(ptr, (size_t)bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // the caller may carry on with pageable memory: leave no sticky error behind
        return fail(MG_ERR_HIP, "hipHostRegister(%p, %llu): %s", ptr, (unsigned long long)bytes, hipGetErrorString(e));
    }
    return MG_OK;
}

int mg_host_unregister(void *ptr)
{
    if (!ptr) return MG_OK;
    HIP_TRY(hipHostUnregister(ptr));
    return MG_OK;
}

int mg_host_alloc(void **out, uint64_t bytes)
{
    if (!out || !bytes) return fail(MG_ERR_INVALID, "mg_host_alloc: null pointer or zero size");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault));
    return MG_OK;
}

int mg_host_free(void *ptr)
{
    if (!ptr) return MG_OK;
    HIP_TRY(hipHostFree(ptr));
    return MG_OK;
}

int mg_reset_state(mg_handle *h)
{
    if (!h) return fail(MG_ERR_INVALID, "null handle");
    TRY(single_gpu_only(h, "mg_reset_state"));
    HIP_TRY(hipSetDevice(h->cfg.device));
    for (int l = 0; l <= h->L; l++) {
        if (h->lv[l].R.p) HIP_TRY(hipMemsetAsync(h->lv[l].R.p, 0, (size_t)h->lv[l].R.elems() * sizeof(double), h->stream));
        if (l >= 1) HIP_TRY(hipMemsetAsync(h->lv[l].eps.p, 0, (size_t)h->lv[l].eps.elems() * sizeof(double), h->stream));
    }
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MG_OK;
}

int mg_get_level_array(mg_handle *h, int which, int level, double *out_flat, int64_t dims_out[3])
{
    if (!h || !out_flat) return fail(MG_ERR_INVALID, "null argument");
    TRY(single_gpu_only(h, "mg_get_level_array"));
    if (level < 0 || level > h->L) return fail(MG_ERR_INVALID, "level %d out of range", level);
    if (which != 0 && which != 1) return fail(MG_ERR_INVALID, "which must be 0 (R) or 1 (eps)");
    if (which == 1 && level == 0) return fail(MG_ERR_INVALID, "eps_0 is not materialised on the device");
    if (which == 0 && level == 0 && h->fused_residual)
        return fail(MG_ERR_STATE, "R_0 is kept only with MG_FLAG_KEEP_FINE_RESIDUAL");
    HIP_TRY(hipSetDevice(h->cfg.device));
    View src = which == 0 ? h->lv[level].R : h->lv[level].eps;
    int rc = d2h(out_flat, src, h->stream);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (dims_out)
        for (int a = 0; a < 3; a++) dims_out[a] = src.d[a];
    return MG_OK;
}

int mg_set_profiling(mg_handle *h, int on)
{
    if (!h) return fail(MG_ERR_INVALID, "null handle");
    TRY(single_gpu_only(h, "mg_set_profiling"));
    drain_profile(h);
    h->profiling = on != 0;
    return MG_OK;
}

int mg_get_timers(mg_handle *h, mg_timers *out)
{
    if (!h || !out) return fail(MG_ERR_INVALID, "null argument");
    TRY(single_gpu_only(h, "mg_get_timers"));
    drain_profile(h);
    *out = h->timers;
    return MG_OK;
}

int mg_reset_timers(mg_handle *h)
{
    if (!h) return fail(MG_ERR_INVALID, "null handle");
    TRY(single_gpu_only(h, "mg_reset_timers"));
    drain_profile(h);
    memset(&h->timers, 0, sizeof(h->timers));
    return MG_OK;
}

int mg_last_call_ms(mg_handle *h, double *ms)
{
    if (!h || !ms) return fail(MG_ERR_INVALID, "null argument");
    if (h->multi) return mg_dist_stats(h->multi->ranks[0], ms, nullptr);
    *ms = h->last_ms;
    return MG_OK;
}

/* ------------------------------------------------------------------------ */
/* single operators on host arrays                                           */

namespace {

// Device buffers of one test call; freed on scope exit.
struct Scratch {
    std::vector<void *> ptrs;
    ~Scratch()
    {
        for (void *p : ptrs) hipFree(p);
    }
    int view(View *v, const int64_t d[3], const double *host_flat)
    {
        int rc = alloc_view(v, d);
        if (rc) return rc;
        ptrs.push_back(v->p);
        if (host_flat) {
            rc = h2d(*v, host_flat, nullptr);
            if (rc) return rc;
            HIP_TRY(hipDeviceSynchronize());
        }
        return MG_OK;
    }
    int flat(double **p, int64_t count, const double *host)
    {
        HIP_TRY(hipMalloc((void **)p, (size_t)count * sizeof(double)));
        ptrs.push_back(*p);
        if (host) HIP_TRY(hipMemcpy(*p, host, (size_t)count * sizeof(double), hipMemcpyHostToDevice));
        return MG_OK;
    }
    int download(View v, double *host_flat)
    {
        int rc = d2h(host_flat, v, nullptr);
        if (rc) return rc;
        HIP_TRY(hipDeviceSynchronize());
        return MG_OK;
    }
};

int check_dims(const int64_t n[3], const char *what, int64_t lo)
{
    if (!n) return fail(MG_ERR_INVALID, "%s: null dims", what);
    for (int a = 0; a < 3; a++)
        if (n[a] < lo || n[a] > kMaxDim) return fail(MG_ERR_INVALID, "%s: dim %d = %lld out of range", what, a, (long long)n[a]);
    return MG_OK;
}

int reduce_to_host(double *partials, int64_t count, double *out)
{
    double *d_sum = nullptr;
    HIP_TRY(hipMalloc((void **)&d_sum, sizeof(double)));
    mgk::sum_partials(nullptr, partials, count, d_sum);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpy(out, d_sum, sizeof(double), hipMemcpyDeviceToHost);
    hipFree(d_sum);
    HIP_TRY(e);
    return MG_OK;
}

}  // namespace

int mg_op_residual(int device, const int64_t n[3], const double dh[3], const double *phi,
                   const double *rho, double *R_out, double *sumsq)
{
    if (!dh || !phi || !rho || !R_out) return fail(MG_ERR_INVALID, "null argument");
    TRY(check_dims(n, "mg_op_residual", 3));
    TRY(select_device(device));
    Scratch sc;
    View vphi, vb, vR;
    TRY(sc.view(&vphi, n, phi));
    TRY(sc.view(&vb, n, rho));
    mgk::div_eps0(nullptr, vb);  // b = rho/EPS_0
    TRY(sc.view(&vR, n, nullptr));
    double *partials = nullptr;
    const IRange all{0, n[0]};
    const int64_t np = mgk::residual_partials(n, all);
    TRY(sc.flat(&partials, np, nullptr));
    mgk::residual(nullptr, vphi, vb, vR, n, level_coef(dh, 0), partials, all);
    HIP_TRY(hipGetLastError());
    double s = 0;
    TRY(reduce_to_host(partials, np, &s));
    if (sumsq) *sumsq = s;
    return sc.download(vR, R_out);
}

int mg_op_residual_restrict(int device, const int64_t n[3], const double dh[3], const double *phi,
                            const double *rho, double *R_fine_out, double *R_coarse, double *sumsq)
{
    if (!dh || !phi || !rho || !R_coarse) return fail(MG_ERR_INVALID, "null argument");
    TRY(check_dims(n, "mg_op_residual_restrict", 3));
    TRY(select_device(device));
    int64_t nc[3];
    for (int a = 0; a < 3; a++) nc[a] = n[a] >> 1;
    Scratch sc;
    View vphi, vb, vR, vc;
    TRY(sc.view(&vphi, n, phi));
    TRY(sc.view(&vb, n, rho));
    mgk::div_eps0(nullptr, vb);
    if (R_fine_out) TRY(sc.view(&vR, n, nullptr));
    else vR = make_view(nullptr, n);
    TRY(sc.view(&vc, nc, R_coarse));
    double *partials = nullptr;
    const IRange all{0, n[0]};
    const int64_t np = mgk::residual_restrict_partials(n, all);
    TRY(sc.flat(&partials, np, nullptr));
    mgk::residual_restrict(nullptr, vphi, vb, vR, vc, n, level_coef(dh, 0), partials, all);
    HIP_TRY(hipGetLastError());
    double s = 0;
    TRY(reduce_to_host(partials, np, &s));
    if (sumsq) *sumsq = s;
    if (R_fine_out) TRY(sc.download(vR, R_fine_out));
    return sc.download(vc, R_coarse);
}

int mg_op_restrict27(int device, const int64_t n_fine[3], const double *R_fine, double *R_coarse)
{
    if (!R_fine || !R_coarse) return fail(MG_ERR_INVALID, "null argument");
    TRY(check_dims(n_fine, "mg_op_restrict27", 2));
    TRY(select_device(device));
    int64_t nc[3];
    for (int a = 0; a < 3; a++) nc[a] = n_fine[a] >> 1;
    Scratch sc;
    View vf, vc;
    TRY(sc.view(&vf, n_fine, R_fine));
    TRY(sc.view(&vc, nc, R_coarse));
    mgk::restrict27(nullptr, vf, vc, n_fine, IRange{0, nc[0]});
    HIP_TRY(hipGetLastError());
    return sc.download(vc, R_coarse);
}

int mg_op_restrict7(int device, const int64_t n_fine[3], const double *R_fine, double *R_coarse)
{
    if (!R_fine || !R_coarse) return fail(MG_ERR_INVALID, "null argument");
    TRY(check_dims(n_fine, "mg_op_restrict7", 2));
    TRY(select_device(device));
    int64_t nc[3];
    for (int a = 0; a < 3; a++) nc[a] = n_fine[a] >> 1;
    Scratch sc;
    View vf, vc;
    TRY(sc.view(&vf, n_fine, R_fine));
    TRY(sc.view(&vc, nc, R_coarse));
    mgk::restrict7(nullptr, vf, vc, n_fine, IRange{0, nc[0]});
    HIP_TRY(hipGetLastError());
    return sc.download(vc, R_coarse);
}

int mg_op_prolong(int device, const int64_t n[3], const int64_t fine_dims[3], const double *eps_coarse,
                  double *eps_fine)
{
    if (!eps_coarse || !eps_fine) return fail(MG_ERR_INVALID, "null argument");
    TRY(check_dims(n, "mg_op_prolong", 2));
    TRY(check_dims(fine_dims, "mg_op_prolong", 2));
    for (int a = 0; a < 3; a++)
        if (fine_dims[a] < n[a]) return fail(MG_ERR_INVALID, "mg_op_prolong: fine_dims[%d] < n[%d]", a, a);
    TRY(select_device(device));
    int64_t cd[3];
    for (int a = 0; a < 3; a++) cd[a] = (n[a] >> 1) + 1;
    Scratch sc;
    View vc, vf;
    TRY(sc.view(&vc, cd, eps_coarse));
    TRY(sc.view(&vf, fine_dims, eps_fine));
    mgk::prolong(nullptr, vc, vf, n, IRange{0, n[0]});
    HIP_TRY(hipGetLastError());
    return sc.download(vf, eps_fine);
}

int mg_op_correct(int device, int64_t count, double *phi, const double *eps)
{
    if (!phi || !eps || count < 0) return fail(MG_ERR_INVALID, "bad argument");
    TRY(select_device(device));
    if (count == 0) return MG_OK;
    Scratch sc;
    double *dphi = nullptr, *deps = nullptr;
    TRY(sc.flat(&dphi, count, phi));
    TRY(sc.flat(&deps, count, eps));
    mgk::correct(nullptr, dphi, deps, count);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(phi, dphi, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
    return MG_OK;
}

int mg_op_prolong_correct(int device, const int64_t n[3], const double *eps_coarse, double *phi)
{
    if (!eps_coarse || !phi) return fail(MG_ERR_INVALID, "null argument");
    TRY(check_dims(n, "mg_op_prolong_correct", 2));
    TRY(select_device(device));
    int64_t cd[3];
    for (int a = 0; a < 3; a++) cd[a] = (n[a] >> 1) + 1;
    Scratch sc;
    View vc, vphi;
    TRY(sc.view(&vc, cd, eps_coarse));
    TRY(sc.view(&vphi, n, phi));
    mgk::prolong_correct(nullptr, vc, vphi, n, IRange{0, n[0]});
    HIP_TRY(hipGetLastError());
    return sc.download(vphi, phi);
}

int mg_op_smooth_rb(int device, const int64_t n[3], const int64_t u_dims[3], const int64_t rhs_dims[3],
                    const double c[3], double omega, int sweeps, int rhs_is_rho, const double *rhs,
                    double *u)
{
    if (!c || !rhs || !u || sweeps < 0) return fail(MG_ERR_INVALID, "bad argument");
    TRY(check_dims(n, "mg_op_smooth_rb", 3));
    TRY(check_dims(u_dims, "mg_op_smooth_rb", 3));
    TRY(check_dims(rhs_dims, "mg_op_smooth_rb", 3));
    for (int a = 0; a < 3; a++)
        if (u_dims[a] < n[a] || rhs_dims[a] < n[a]) return fail(MG_ERR_INVALID, "mg_op_smooth_rb: array dims smaller than n");
    TRY(select_device(device));
    Scratch sc;
    View vu, valt, vr;
    TRY(sc.view(&vu, u_dims, u));
    TRY(sc.view(&valt, u_dims, nullptr));
    TRY(sc.view(&vr, rhs_dims, rhs));
    if (rhs_is_rho) mgk::div_eps0(nullptr, vr);
    const Coef k = mgk::make_coef(c[0], c[1], c[2], 2 * c[0] + 2 * c[1] + 2 * c[2]);
    mgk::smooth_rb(nullptr, &vu, &valt, vr, n, k, omega, sweeps, IRange{0, u_dims[0]});
    HIP_TRY(hipGetLastError());
    return sc.download(vu, u);
}

int mg_op_compute_ef(int device, const int64_t n[3], const double dh[3], const double *phi, double *ef)
{
    if (!dh || !phi || !ef) return fail(MG_ERR_INVALID, "null argument");
    TRY(check_dims(n, "mg_op_compute_ef", 3));
    for (int a = 0; a < 3; a++)
        if (!(dh[a] > 0)) return fail(MG_ERR_INVALID, "mg_op_compute_ef: dh[%d] must be positive", a);
    TRY(select_device(device));
    Scratch sc;
    View vphi;
    TRY(sc.view(&vphi, n, phi));
    double *def = nullptr;
    const int64_t count = count3(n) * 3;
    TRY(sc.flat(&def, count, nullptr));
    mgk::compute_ef(nullptr, vphi, def, 0, n, dh, IRange{0, n[0]});
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(ef, def, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
    return MG_OK;
}

int mg_op_sumsq(int device, int64_t count, const double *x, double *out)
{
    if (!x || !out || count < 0) return fail(MG_ERR_INVALID, "bad argument");
    TRY(select_device(device));
    if (count == 0) { *out = 0; return MG_OK; }
    Scratch sc;
    double *dx = nullptr, *partials = nullptr;
    TRY(sc.flat(&dx, count, x));
    const int64_t np = mgk::sumsq_partials(count);
    TRY(sc.flat(&partials, np, nullptr));
    mgk::sumsq(nullptr, dx, count, partials);
    HIP_TRY(hipGetLastError());
    return reduce_to_host(partials, np, out);
}

/* ------------------------------------------------------------------------ */
/* device-level operators (slab partition)                                   */

namespace {

int dev_view(const mg_dev_array *a, View *v, const char *what, bool allow_null = false)
{
    if (!a) return fail(MG_ERR_INVALID, "%s: null array descriptor", what);
    if (!a->ptr && !allow_null) return fail(MG_ERR_INVALID, "%s: null device pointer", what);
    if (a->planes < 1 || a->nj < 1 || a->nk < 1 || a->i0 < 0)
        return fail(MG_ERR_INVALID, "%s: bad array shape", what);
    if (((uintptr_t)a->ptr) & 15) return fail(MG_ERR_INVALID, "%s: device pointer not 16-byte aligned", what);
    v->p = (double *)a->ptr;
    v->d[0] = a->planes;
    v->d[1] = a->nj;
    v->d[2] = a->nk;
    v->sj = mgk::pitch_for(a->nk);
    v->si = v->sj * a->nj;
    v->i0 = a->i0;
    return MG_OK;
}

// planes [lo, hi) intersected with [0, gmax) must be stored in v
int need_planes(const View &v, int64_t lo, int64_t hi, int64_t gmax, const char *what)
{
    if (lo < 0) lo = 0;
    if (hi > gmax) hi = gmax;
    if (lo < hi && (lo < v.i0 || hi > v.i_end()))
        return fail(MG_ERR_INVALID, "%s: needs planes [%lld,%lld) but the array holds [%lld,%lld)", what,
                    (long long)lo, (long long)hi, (long long)v.i0, (long long)v.i_end());
    return MG_OK;
}

Coef coef4(const double c[4]) { return mgk::make_coef(c[0], c[1], c[2], c[3]); }

// rows and columns of a stored array must cover what the loops of an n-sized level touch
int need_rows(const View &v, int64_t nj, int64_t nk, const char *what)
{
    if (v.d[1] < nj || v.d[2] < nk)
        return fail(MG_ERR_INVALID, "%s: array has %lld x %lld rows x columns, the level needs %lld x %lld", what,
                    (long long)v.d[1], (long long)v.d[2], (long long)nj, (long long)nk);
    return MG_OK;
}

int check_level(const int64_t n[3], const char *what)
{
    for (int a = 0; a < 3; a++)
        if (n[a] < 2 || n[a] > kMaxDim) return fail(MG_ERR_INVALID, "%s: level size %lld out of range", what, (long long)n[a]);
    return MG_OK;
}

}  // namespace

int64_t mg_dev_pitch(int64_t nk) { return mgk::pitch_for(nk); }

int mg_level_coef(const double dh[3], int level, double out[4])
{
    if (!dh || !out || level < 0 || level >= MG_MAX_LEVELS) return fail(MG_ERR_INVALID, "bad argument");
    const Coef c = level_coef(dh, level);
    out[0] = c.cx;
    out[1] = c.cy;
    out[2] = c.cz;
    out[3] = c.d;
    return MG_OK;
}

int mg_dev_div_eps0(void *stream, const mg_dev_array *v)
{
    View a;
    TRY(dev_view(v, &a, "mg_dev_div_eps0"));
    mgk::div_eps0((hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return MG_OK;
}

int mg_dev_sweep_out_of_place(const mg_dev_array *u)
{
    View vu;
    if (dev_view(u, &vu, "mg_dev_sweep_out_of_place", true)) return 0;
    return mgk::sweep_is_out_of_place(vu) ? 1 : 0;
}

int mg_dev_sweeps_in_pairs(const mg_dev_array *u, const mg_dev_array *rhs, const int64_t n[3], int64_t lo, int64_t hi)
{
    View vu, vr;
    if (!n || dev_view(u, &vu, "mg_dev_sweeps_in_pairs(u)", true) || dev_view(rhs, &vr, "mg_dev_sweeps_in_pairs(rhs)", true))
        return 0;
    View va = vu;
    va.p = vu.p ? vu.p : reinterpret_cast<double *>(sizeof(double));  // "a twin array exists"; never dereferenced
    return mgk::sweeps_in_pairs(vu, va, vr, n, IRange{lo, hi}) ? 1 : 0;
}

int mg_dev_onload_folds(const mg_dev_array *u, int correct)
{
    View vu;
    if (dev_view(u, &vu, "mg_dev_onload_folds", true)) return 0;
    return mgk::onload_folds(vu, correct != 0) ? 1 : 0;
}

namespace {
// second plane range of a sweep call: empty, or as long as the first and above it
int check_second_range(const View &vu, int64_t lo, int64_t hi, int64_t lo2, int64_t hi2, int sweeps, const char *what)
{
    if (lo2 == hi2) return MG_OK;
    if (lo2 > hi2 || hi2 - lo2 != hi - lo || lo2 < hi || hi2 > vu.i_end() || sweeps != 1)
        return fail(MG_ERR_INVALID, "%s: the second plane range must be as long as the first, above it, stored, and sweeps must be 1", what);
    return MG_OK;
}
}  // namespace

int mg_dev_smooth_rb(void *stream, const mg_dev_array *u, const mg_dev_array *alt, const mg_dev_array *rhs,
                     const int64_t n[3], const double c[4], double omega, int sweeps, int64_t lo, int64_t hi,
                     int64_t lo2, int64_t hi2, int *result_in_alt)
{
    if (!n || !c || !result_in_alt || sweeps < 0) return fail(MG_ERR_INVALID, "bad argument");
    View vu, va, vr;
    TRY(dev_view(u, &vu, "mg_dev_smooth_rb(u)"));
    TRY(dev_view(alt, &va, "mg_dev_smooth_rb(alt)"));
    TRY(dev_view(rhs, &vr, "mg_dev_smooth_rb(rhs)"));
    if (va.d[0] != vu.d[0] || va.d[1] != vu.d[1] || va.d[2] != vu.d[2] || va.i0 != vu.i0)
        return fail(MG_ERR_INVALID, "mg_dev_smooth_rb: alt must have the shape of u");
    if (lo < vu.i0 || hi > vu.i_end() || lo > hi) return fail(MG_ERR_INVALID, "mg_dev_smooth_rb: bad plane range");
    TRY(check_level(n, "mg_dev_smooth_rb"));
    TRY(need_rows(vu, n[1], n[2], "mg_dev_smooth_rb(u)"));
    TRY(need_rows(vr, n[1] - 1, n[2] - 1, "mg_dev_smooth_rb(rhs)"));
    TRY(check_second_range(vu, lo, hi, lo2, hi2, sweeps, "mg_dev_smooth_rb"));
    for (int part = 0; part < (lo2 < hi2 ? 2 : 1); part++) {
        const int64_t a = part ? lo2 : lo, b = part ? hi2 : hi;
        const int64_t ulo = a < 1 ? 1 : a, uhi = b > n[0] - 1 ? n[0] - 1 : b;  // planes actually updated
        if (ulo < uhi) {
            // red update of planes ulo-1..uhi reads planes ulo-2..uhi+1 of u; rhs is read on updated planes
            TRY(need_planes(vu, ulo - 2, uhi + 2, n[0], "mg_dev_smooth_rb(u)"));
            TRY(need_planes(vr, ulo - 1, uhi + 1, n[0] - 1, "mg_dev_smooth_rb(rhs)"));
        }
    }
    const double *before = vu.p;
    if (mgk::smooth_rb((hipStream_t)stream, &vu, &va, vr, n, coef4(c), omega, sweeps, IRange{lo, hi},
                       IRange{lo2, hi2}) < 0)
        return fail(MG_ERR_INVALID, "mg_dev_smooth_rb: a second plane range needs the out-of-place sweep kernel");
    HIP_TRY(hipGetLastError());
    *result_in_alt = vu.p != before;
    return MG_OK;
}

int mg_dev_residual_restrict(void *stream, const mg_dev_array *phi, const mg_dev_array *b, const mg_dev_array *Rc,
                             const int64_t n[3], const double c[4], double *partials, int64_t capacity,
                             int64_t lo, int64_t hi, int64_t *count)
{
    if (!n || !c || !partials || !count) return fail(MG_ERR_INVALID, "null argument");
    View vp, vb, vc;
    TRY(dev_view(phi, &vp, "mg_dev_residual_restrict(phi)"));
    TRY(dev_view(b, &vb, "mg_dev_residual_restrict(b)"));
    TRY(dev_view(Rc, &vc, "mg_dev_residual_restrict(Rc)"));
    if ((lo & 1) || lo < 0 || hi > n[0] || lo >= hi) return fail(MG_ERR_INVALID, "mg_dev_residual_restrict: bad plane range");
    TRY(check_level(n, "mg_dev_residual_restrict"));
    if (vp.d[1] != n[1] || vp.d[2] != n[2]) return fail(MG_ERR_INVALID, "mg_dev_residual_restrict: phi rows/columns must equal the level size");
    TRY(need_rows(vb, n[1] - 1, n[2] - 1, "mg_dev_residual_restrict(b)"));
    TRY(need_rows(vc, n[1] >> 1, n[2] >> 1, "mg_dev_residual_restrict(Rc)"));
    if ((hi & 1) && hi != n[0]) return fail(MG_ERR_INVALID, "mg_dev_residual_restrict: hi must be even or the level's last plane");
    TRY(need_planes(vp, lo - 2, hi + 1, n[0], "mg_dev_residual_restrict(phi)"));
    TRY(need_planes(vb, lo - 1, hi, n[0], "mg_dev_residual_restrict(b)"));
    const int64_t clo = lo >> 1, chi = (hi >> 1) < (n[0] >> 1) ? (hi >> 1) : (n[0] >> 1);
    TRY(need_planes(vc, clo, chi, n[0] >> 1, "mg_dev_residual_restrict(Rc)"));
    const IRange r{lo, hi};
    const int64_t np = mgk::residual_restrict_partials(n, r);
    if (np > capacity) return fail(MG_ERR_INVALID, "mg_dev_residual_restrict: partials buffer too small (%lld > %lld)", (long long)np, (long long)capacity);
    View none = vp;
    none.p = nullptr;
    mgk::residual_restrict((hipStream_t)stream, vp, vb, none, vc, n, coef4(c), partials, r);
    HIP_TRY(hipGetLastError());
    *count = np;
    return MG_OK;
}

namespace {

// Workspace of mg_dev_residual_restrict_noalloc: the per-workgroup partial sums.  One copy of the
// array exists on every device the library's code object is loaded on, at its own address; the
// name below is only the host-side handle of the symbol, never a device address.
constexpr int64_t kNoAllocPartials = 1 << 17;  // 1 MiB
static __device__ __align__(256) double scratch[kNoAllocPartials];

constexpr int kMaxScratchDevices = 64;
double *g_scratch_ptr[kMaxScratchDevices];  // device address of `scratch` per device ordinal
std::mutex g_scratch_mutex;

// Device address of the copy of `scratch` on the device the stream belongs to (the current device
// for the null stream), looked up once per device with that device current.  The first lookup on a
// device may make the runtime load the library's code object there, which allocates the object's
// global variables: that is why one warm-up call per device must come before a call that has to
// be allocation free (see the header).
int scratch_address(hipStream_t stream, double **out)
{
    int cur = -1, dev = -1;
    HIP_TRY(hipGetDevice(&cur));
    dev = cur;
    if (stream) {
        hipDevice_t sd = 0;
        HIP_TRY(hipStreamGetDevice(stream, &sd));
        dev = (int)sd;
    }
    if (dev < 0 || dev >= kMaxScratchDevices) return fail(MG_ERR_INVALID, "device ordinal %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        if (dev != cur) HIP_TRY(hipSetDevice(dev));
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (dev != cur) HIP_TRY(hipSetDevice(cur));
        HIP_TRY(e);
        if (!p) return fail(MG_ERR_HIP, "hipGetSymbolAddress returned a null address");
        g_scratch_ptr[dev] = (double *)p;
    }
    *out = g_scratch_ptr[dev];
    return MG_OK;
}

}  // namespace

int mg_dev_residual_restrict_noalloc(void *stream, const mg_dev_array *phi, const mg_dev_array *b,
                                     const mg_dev_array *Rc, const int64_t n[3], const double c[4], int64_t lo,
                                     int64_t hi, double *sumsq)
{
    if (!n || !sumsq) return fail(MG_ERR_INVALID, "null argument");
    TRY(check_level(n, "mg_dev_residual_restrict_noalloc"));
    if (lo < hi) {
        const int64_t np = mgk::residual_restrict_partials(n, IRange{lo, hi});
        if (np > kNoAllocPartials)
            return fail(MG_ERR_INVALID,
                        "mg_dev_residual_restrict_noalloc: %lld workgroup partial sums exceed the built-in workspace of "
                        "%lld; split the plane range or use mg_dev_residual_restrict with a caller-owned buffer",
                        (long long)np, (long long)kNoAllocPartials);
    }
    double *ws = nullptr;
    TRY(scratch_address((hipStream_t)stream, &ws));
    int64_t count = 0;
    // every partial the sum reads is written by the launch before it: nothing needs clearing
    TRY(mg_dev_residual_restrict(stream, phi, b, Rc, n, c, ws, kNoAllocPartials, lo, hi, &count));
    return mg_dev_sum_partials(stream, ws, count, sumsq);
}

int mg_dev_sum_partials(void *stream, const double *partials, int64_t count, double *out)
{
    if (!partials || !out || count < 0) return fail(MG_ERR_INVALID, "bad argument");
    mgk::sum_partials((hipStream_t)stream, partials, count, out);
    HIP_TRY(hipGetLastError());
    return MG_OK;
}

int mg_dev_restrict27(void *stream, const mg_dev_array *Rf, const mg_dev_array *Rc, const int64_t nf[3],
                      int64_t lo, int64_t hi)
{
    if (!nf) return fail(MG_ERR_INVALID, "null argument");
    View vf, vc;
    TRY(dev_view(Rf, &vf, "mg_dev_restrict27(Rf)"));
    TRY(dev_view(Rc, &vc, "mg_dev_restrict27(Rc)"));
    if (lo < 0 || hi > (nf[0] >> 1) || lo > hi) return fail(MG_ERR_INVALID, "mg_dev_restrict27: bad plane range");
    TRY(check_level(nf, "mg_dev_restrict27"));
    TRY(need_rows(vf, nf[1], nf[2], "mg_dev_restrict27(Rf)"));
    if (vc.d[1] != (nf[1] >> 1) || vc.d[2] != (nf[2] >> 1))
        return fail(MG_ERR_INVALID, "mg_dev_restrict27: Rc rows/columns must be the fine level's size halved");
    if (lo < hi) {
        TRY(need_planes(vc, lo, hi, nf[0] >> 1, "mg_dev_restrict27(Rc)"));
        TRY(need_planes(vf, lo == 0 ? 0 : 2 * lo - 1, 2 * hi, nf[0], "mg_dev_restrict27(Rf)"));
    }
    mgk::restrict27((hipStream_t)stream, vf, vc, nf, IRange{lo, hi});
    HIP_TRY(hipGetLastError());
    return MG_OK;
}

int mg_dev_prolong(void *stream, const mg_dev_array *ec, const mg_dev_array *out, const int64_t n[3], int64_t lo,
                   int64_t hi, int correct)
{
    if (!n) return fail(MG_ERR_INVALID, "null argument");
    View vc, vo;
    TRY(dev_view(ec, &vc, "mg_dev_prolong(ec)"));
    TRY(dev_view(out, &vo, "mg_dev_prolong(out)"));
    if (lo < 0 || lo > hi) return fail(MG_ERR_INVALID, "mg_dev_prolong: bad plane range");
    TRY(check_level(n, "mg_dev_prolong"));
    TRY(need_rows(vo, n[1] - 1, n[2] - 1, "mg_dev_prolong(out)"));
    TRY(need_rows(vc, (n[1] >> 1) + 1, (n[2] >> 1) + 1, "mg_dev_prolong(ec)"));
    const int64_t whi = hi > n[0] - 1 ? n[0] - 1 : hi;  // planes 0..n-2 are written
    if (lo < whi) {
        TRY(need_planes(vo, lo, whi, n[0], "mg_dev_prolong(out)"));
        TRY(need_planes(vc, lo >> 1, ((whi - 1) >> 1) + 2, (n[0] >> 1) + 1, "mg_dev_prolong(ec)"));
    }
    if (correct) mgk::prolong_correct((hipStream_t)stream, vc, vo, n, IRange{lo, hi});
    else mgk::prolong((hipStream_t)stream, vc, vo, n, IRange{lo, hi});
    HIP_TRY(hipGetLastError());
    return MG_OK;
}

int mg_dev_prolong_smooth_rb(void *stream, const mg_dev_array *ec, int correct, const mg_dev_array *u,
                             const mg_dev_array *alt, const mg_dev_array *rhs, const int64_t n[3],
                             const double c[4], double omega, int sweeps, int64_t lo, int64_t hi,
                             int64_t lo2, int64_t hi2, int *result_in_alt)
{
    if (!n || !c || !result_in_alt || sweeps < 1) return fail(MG_ERR_INVALID, "bad argument");
    View vc, vu, va, vr;
    TRY(dev_view(ec, &vc, "mg_dev_prolong_smooth_rb(ec)"));
    TRY(dev_view(u, &vu, "mg_dev_prolong_smooth_rb(u)"));
    TRY(dev_view(alt, &va, "mg_dev_prolong_smooth_rb(alt)"));
    TRY(dev_view(rhs, &vr, "mg_dev_prolong_smooth_rb(rhs)"));
    if (va.d[0] != vu.d[0] || va.d[1] != vu.d[1] || va.d[2] != vu.d[2] || va.i0 != vu.i0)
        return fail(MG_ERR_INVALID, "mg_dev_prolong_smooth_rb: alt must have the shape of u");
    if (lo < vu.i0 || hi > vu.i_end() || lo > hi) return fail(MG_ERR_INVALID, "mg_dev_prolong_smooth_rb: bad plane range");
    TRY(check_level(n, "mg_dev_prolong_smooth_rb"));
    TRY(need_rows(vu, n[1], n[2], "mg_dev_prolong_smooth_rb(u)"));
    TRY(need_rows(vr, n[1] - 1, n[2] - 1, "mg_dev_prolong_smooth_rb(rhs)"));
    if (vc.d[1] < (n[1] >> 1) + 1 || vc.d[2] < (n[2] >> 1) + 1)
        return fail(MG_ERR_INVALID, "mg_dev_prolong_smooth_rb: coarse array smaller than (n>>1)+1");
    TRY(check_second_range(vu, lo, hi, lo2, hi2, sweeps, "mg_dev_prolong_smooth_rb"));
    for (int part = 0; part < (lo2 < hi2 ? 2 : 1); part++) {
        const int64_t a = part ? lo2 : lo, b = part ? hi2 : hi;
        const int64_t ulo = a < 1 ? 1 : a, uhi = b > n[0] - 1 ? n[0] - 1 : b;
        if (ulo < uhi) {
            TRY(need_planes(vu, ulo - 2, uhi + 2, n[0], "mg_dev_prolong_smooth_rb(u)"));
            TRY(need_planes(vr, ulo - 1, uhi + 1, n[0] - 1, "mg_dev_prolong_smooth_rb(rhs)"));
        }
        // fine planes the operation is applied to, and the coarse planes they interpolate from
        int64_t flo = a - 2 < vu.i0 ? vu.i0 : a - 2, fhi = b + 2 > vu.i_end() ? vu.i_end() : b + 2;
        if (fhi > n[0] - 1) fhi = n[0] - 1;
        if (flo < fhi) TRY(need_planes(vc, flo >> 1, ((fhi - 1) >> 1) + 2, (n[0] >> 1) + 1, "mg_dev_prolong_smooth_rb(ec)"));
    }
    const double *before = vu.p;
    if (mgk::prolong_smooth_rb((hipStream_t)stream, vc, correct != 0, &vu, &va, vr, n, coef4(c), omega, sweeps,
                               IRange{lo, hi}, IRange{lo2, hi2}) < 0)
        return fail(MG_ERR_INVALID, "mg_dev_prolong_smooth_rb: a second plane range needs the out-of-place sweep kernel");
    HIP_TRY(hipGetLastError());
    *result_in_alt = vu.p != before;
    return MG_OK;
}

int mg_dev_compute_ef(void *stream, const mg_dev_array *phi, void *ef, int64_t ef_i0, const int64_t n[3],
                      const double dh[3], int64_t lo, int64_t hi)
{
    if (!n || !dh || !ef) return fail(MG_ERR_INVALID, "null argument");
    View vp;
    TRY(dev_view(phi, &vp, "mg_dev_compute_ef(phi)"));
    if (n[0] < 3 || n[1] < 3 || n[2] < 3 || vp.d[1] != n[1] || vp.d[2] != n[2])
        return fail(MG_ERR_INVALID, "mg_dev_compute_ef: phi rows/columns must match n (>= 3 per axis)");
    if (lo < 0 || hi > n[0] || lo > hi || ef_i0 > lo) return fail(MG_ERR_INVALID, "mg_dev_compute_ef: bad plane range");
    if (lo < hi) {
        TRY(need_planes(vp, lo - 1, hi + 1, n[0], "mg_dev_compute_ef(phi)"));
        if (lo == 0) TRY(need_planes(vp, 0, 3, n[0], "mg_dev_compute_ef(phi)"));
        if (hi == n[0]) TRY(need_planes(vp, n[0] - 3, n[0], n[0], "mg_dev_compute_ef(phi)"));
    }
    mgk::compute_ef((hipStream_t)stream, vp, (double *)ef, ef_i0, n, dh, IRange{lo, hi});
    HIP_TRY(hipGetLastError());
    return MG_OK;
}

}  // extern "C"
