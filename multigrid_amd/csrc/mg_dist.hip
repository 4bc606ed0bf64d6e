// Slab-partitioned multigrid solve: one rank per GPU, driven from C++ inside libmg_hip
// (mg_dist_* in include/mg_hip.h).
//
// The reference is a single-threaded CPU program (cpp/PotentialSolver.cpp:3756-4216); nothing
// here has a counterpart in it.  What is distributed is the cycle the single-GPU handle runs,
// operator by operator, with fields bit-identical to the undivided red-black cycle (the residual
// norm, summed per rank and then over the ranks, agrees to rounding):
//
//   * the i axis (slowest; planes are contiguous, so a halo is one message) of every level
//     l < cut is split into slabs whose fine-plane boundaries are multiples of 2^cut, so the
//     coarse slabs nest and the fused residual+restriction always starts on an even plane;
//   * a rank stores its planes plus halo planes: 2 for swept fields (phi, eps_l), 1 for
//     right-hand sides (b, R_l).  A red-black sweep recomputes the red update of the first halo
//     plane on either side, so ONE exchange of two planes per side per sweep replaces an
//     exchange after every colour;
//   * the sweep kernel is out of place, so a sweep is issued as (1) the two planes next to each
//     neighbour, (2) start of the exchange of exactly those planes on the communication stream,
//     (3) the interior -- which reads no halo plane of the previous sweep -- and the next reader
//     of the halos waits for the transfer.  The fine residual+restriction is split the same way;
//   * deep halos on the coarse distributed levels: where every rank finds them on its two
//     neighbours, a rank receives ONCE per cycle as many planes of R_l as all nu_l sweeps of level l
//     reach (2 per sweep), forms eps_l on its slab widened by that much and sweeps ranges that
//     shrink by two planes per side and sweep -- no exchange between the sweeps, no separate
//     launches for the planes next to a neighbour, two sweeps per launch where the kernels offer
//     it.  The planes computed twice (here and on the neighbour) get the same bits on both;
//   * the sum of squared residuals is all-reduced once per cycle, beside the kernels: nothing on the
//     kernel stream waits for it but the next writer of the sum and the host's read of it;
//   * R of level `cut` is all-gathered and EVERY rank solves level `cut` and everything below it on
//     whole arrays of its own with the single-GPU kernels.  Those levels are serial work of the
//     cycle whoever runs them (everything after them waits for eps_cut); run on all ranks at once
//     they cost what they cost on one, no rank idles and then receives, and eps_cut needs no
//     transfer at all: each rank prolongs from its own copy.
//
// Transports: RCCL (ncclSend/ncclRecv/ncclAllReduce on a stream of its own, ordered against the
// kernel stream with events; librccl is loaded on first use, so single-GPU users never need
// it), and an in-process one for several ranks (host threads) that share one GPU or one
// address space: device-to-device copies between the ranks' slabs behind host barriers.  The
// in-process transport can hold every started exchange back until its slab is next needed --
// the latest moment the schedule allows -- which is how the tests pin the schedule's
// dependencies without a multi-GPU node.
#include "../../include/mg_hip.h"
#include "mg_internal.h"
#include "mg_kernels.h"

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only; the library is bound with dlopen

#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

using mgk::Coef;
using mgk::IRange;
using mgk::View;

#define FAIL(code, ...) return mgk::set_error(code, __VA_ARGS__)
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (void)hipGetLastError(); /* reported here; not left behind for a later call's check */ \
            return mgk::set_error(e_ == hipErrorOutOfMemory ? MG_ERR_NOMEM : MG_ERR_HIP, "%s: %s (%s:%d)", #expr, \
                                  hipGetErrorString(e_), __FILE__, __LINE__);                      \
        }                                                                                          \
    } while (0)
#define TRY(expr)            \
    do {                     \
        int rc_ = (expr);    \
        if (rc_) return rc_; \
    } while (0)

namespace {

// ---------------------------------------------------------------------------- RCCL binding

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;  // optional: without it a failed group cannot be released
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

// librccl, loaded once.  A process that has PyTorch-ROCm loaded gets PyTorch's copy (same
// soname), so both talk to one library.
const Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
#define BIND(f) r.f = reinterpret_cast<decltype(r.f)>(dlsym(r.lib, "nccl" #f))
        BIND(GetUniqueId);
        BIND(CommInitRank);
        BIND(CommDestroy);
        BIND(CommAbort);
        BIND(Send);
        BIND(Recv);
        BIND(AllReduce);
        BIND(GroupStart);
        BIND(GroupEnd);
        BIND(GetErrorString);
#undef BIND
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Send || !r.Recv || !r.AllReduce ||
            !r.GroupStart || !r.GroupEnd || !r.GetErrorString) {
            dlclose(r.lib);
            r.lib = nullptr;
        }
    });
    return r.lib ? &r : nullptr;
}

#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess)                                                                           \
            return mgk::set_error(MG_ERR_COMM, "%s: %s (%s:%d)", #expr, rccl()->GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// ---------------------------------------------------------------------------- plan

constexpr int H_U = 2, H_RHS = 1;  // halo planes of swept fields / of right-hand sides

inline IRange widen(IRange r, int64_t by, int64_t top)
{
    return IRange{r.lo - by < 0 ? 0 : r.lo - by, r.hi + by > top ? top : r.hi + by};
}

// Who owns which planes of which level.  Pure arithmetic.
struct Plan {
    int64_t n0[3];
    int depth, world, cut;
    std::vector<int64_t> bounds;  // fine-plane boundaries: rank r owns [bounds[r], bounds[r+1])

    int init(const int64_t n[3], int depth_, int world_, int cut_)
    {
        for (int a = 0; a < 3; a++) n0[a] = n[a];
        depth = depth_;
        world = world_;
        if (depth < 1) FAIL(MG_ERR_INVALID, "a partitioned solve needs at least one coarse grid");
        int64_t nmin = n0[0] < n0[1] ? n0[0] : n0[1];
        if (n0[2] < nmin) nmin = n0[2];
        if ((nmin >> depth) < 3) FAIL(MG_ERR_INVALID, "grid too small for depth %d", depth);
        if (cut_ <= 0) {  // first level small enough to live on one GPU, at least 1
            cut_ = depth;
            for (int l = 1; l <= depth; l++)
                if ((n0[0] >> l) <= 128) {
                    cut_ = l;
                    break;
                }
        }
        if (cut_ < 1 || cut_ > depth) FAIL(MG_ERR_INVALID, "coarse_cut %d outside 1..depth", cut_);
        cut = cut_;
        const int64_t align = (int64_t)1 << cut;
        const int64_t units = (n0[0] - 1) / align;  // slabs are whole units of 2^cut fine planes
        const int64_t even = units / world;
        if (even < 1)
            FAIL(MG_ERR_INVALID, "%lld planes over %d ranks leave fewer than 2^%d fine planes per rank; lower coarse_cut or the rank count",
                 (long long)n0[0], world, cut);
        bounds.assign(world + 1, 0);
        bounds[world] = n0[0];
        if (world == 1) return MG_OK;
        // Whole units as evenly as they go, the spare ones to the lower ranks (the last rank also
        // takes the planes beyond the last whole unit).  No rank is given less for running the
        // levels below the cut: every rank runs them, at the same time.
        const int64_t spare = units % world;
        for (int r = 1; r < world; r++) bounds[r] = bounds[r - 1] + (even + (r - 1 < spare ? 1 : 0)) * align;
        return MG_OK;
    }
    void level_n(int l, int64_t out[3]) const
    {
        for (int a = 0; a < 3; a++) out[a] = n0[a] >> l;
    }
    // global plane count of an array of level l: eps arrays of l >= 1 have one extra node
    int64_t planes(int l, bool eps) const { return (n0[0] >> l) + ((eps && l >= 1) ? 1 : 0); }
    void dims(int l, bool eps, int64_t out[3]) const
    {
        level_n(l, out);
        if (eps && l >= 1)
            for (int a = 0; a < 3; a++) out[a] += 1;
    }
    IRange own(int l, bool eps, int rank) const
    {
        return IRange{bounds[rank] >> l, rank == world - 1 ? planes(l, eps) : bounds[rank + 1] >> l};
    }
    IRange stored(int l, bool eps, int rank, int halo) const
    {
        IRange o = own(l, eps, rank);
        const int64_t top = planes(l, eps);
        return IRange{o.lo - halo < 0 ? 0 : o.lo - halo, o.hi + halo > top ? top : o.hi + halo};
    }

    // Deep halos.  Planes of eps_l (levels 1..cut-1) a rank must hold after the level's last sweep
    // (`last`: what the level above forms its own planes from) and forms by prolongation before the
    // first (`first`): a sweep needs its input two planes further out on either side than it
    // produces, so sweep s of nu produces `last` widened by 2*(nu-s) and the prolongation before the
    // first sweep covers `last` widened by 2*nu.  The level below is needed on the planes that
    // prolongation reads: fine planes [lo, hi) read coarse planes lo/2 .. hi/2.
    void deep_ranges(int rank, const int *nu, IRange *last, IRange *first) const
    {
        IRange need = stored(0, false, rank, H_U);  // fine planes the correction phi -= P(eps_1) reaches
        for (int l = 1; l < cut; l++) {
            const int64_t top = planes(l, true);
            last[l] = IRange{need.lo >> 1, (need.hi >> 1) + 1 > top ? top : (need.hi >> 1) + 1};
            first[l] = widen(last[l], 2 * (int64_t)nu[l], top);
            need = first[l];
        }
    }
    // the planes of R_l a rank smooths against in that scheme: those of eps_l, as far as R_l has them
    IRange deep_rhs(int l, IRange first) const
    {
        const int64_t top = planes(l, false);
        return IRange{first.lo, first.hi > top ? top : first.hi};
    }
    // Deep halos are used when every rank finds the planes beyond its own on its two neighbours (one
    // message per side and level).
    bool deep_fits(const int *nu) const
    {
        if (world < 2 || cut < 2) return false;
        for (int l = 1; l < cut; l++)
            if (nu[l] > 4096) return false;
        for (int r = 0; r < world; r++) {
            IRange last[MG_MAX_LEVELS], first[MG_MAX_LEVELS];
            deep_ranges(r, nu, last, first);
            for (int l = 1; l < cut; l++) {
                const int64_t lo = r > 0 ? own(l, true, r - 1).lo : 0;
                const int64_t hi = r + 1 < world ? own(l, true, r + 1).hi : planes(l, true);
                if (first[l].lo < lo || first[l].hi > hi) return false;
            }
        }
        return true;
    }
};

// ---------------------------------------------------------------------------- storage

// Planes exchanged with the slab neighbours around an owned range [lo, hi): received into
// [lo - rlo, lo) and [hi, hi + rhi), sent from [lo, lo + slo) to the rank below and from
// [hi - shi, hi) to the rank above.  What a rank sends is what its neighbour receives; the sides
// without a neighbour are ignored.
struct Halo {
    int rlo, slo, shi, rhi;
    static Halo both(int w) { return Halo{w, w, w, w}; }
};

// The one exchange of R_l per cycle on deep halos: what this rank receives is what its ranges reach
// beyond the owned planes; what it sends is what its neighbours' ranges reach into them.
Halo deep_halo(const Plan &P, int rank, int l, const int *nu)
{
    IRange last[MG_MAX_LEVELS], first[MG_MAX_LEVELS], below[MG_MAX_LEVELS], above[MG_MAX_LEVELS];
    P.deep_ranges(rank, nu, last, first);
    const IRange o = P.own(l, false, rank), f = P.deep_rhs(l, first[l]);
    Halo h{(int)(o.lo - f.lo), 0, 0, (int)(f.hi > o.hi ? f.hi - o.hi : 0)};
    if (rank > 0) {
        P.deep_ranges(rank - 1, nu, last, below);
        const IRange fb = P.deep_rhs(l, below[l]);
        h.slo = (int)(fb.hi > o.lo ? fb.hi - o.lo : 0);
    }
    if (rank + 1 < P.world) {
        P.deep_ranges(rank + 1, nu, last, above);
        h.shi = (int)(o.hi - P.deep_rhs(l, above[l]).lo);
    }
    return h;
}

// Planes [v.i0, v.i0 + v.d[0]) of a level's array on this rank's GPU.
struct Slab {
    View v{};
    // a halo exchange into this slab that has been started and not yet waited for
    int tag = 0;  // which array of which level this is (16 * level + an array number): the same on every rank
    bool pending = false;
    hipEvent_t done = nullptr;  // RCCL transport: recorded on the communication stream
    IRange held_own{0, 0};      // in-process transport holding the exchange back: what to do at the wait
    Halo held{0, 0, 0, 0};

    double *plane(int64_t i) const { return v.p + (i - v.i0) * v.si; }
    size_t plane_bytes() const { return (size_t)v.si * sizeof(double); }
};

// ---------------------------------------------------------------------------- transports

// One point-to-point transfer of whole planes: global planes [lo, hi) of a level's array, to or
// from the rank `peer`.
struct Msg {
    int peer;
    bool send;
    int64_t lo, hi;
};

// The transfers of one halo exchange `h` around the owned planes `own` (a side with nothing to move
// is left out -- on both ranks, since one rank's send count is its neighbour's receive count).
int halo_msgs(int rank, int world, IRange own, Halo h, Msg out[4])
{
    int n = 0;
    if (rank + 1 < world) {
        if (h.shi > 0) out[n++] = Msg{rank + 1, true, own.hi - h.shi, own.hi};
        if (h.rhi > 0) out[n++] = Msg{rank + 1, false, own.hi, own.hi + h.rhi};
    }
    if (rank > 0) {
        if (h.slo > 0) out[n++] = Msg{rank - 1, true, own.lo, own.lo + h.slo};
        if (h.rlo > 0) out[n++] = Msg{rank - 1, false, own.lo - h.rlo, own.lo};
    }
    return n;
}

// The transfers of an all-gather of a level's owned planes: this rank's share to every other rank,
// every other rank's share from it (grouped send/recv pairs rather than ncclAllGather: the ranks'
// shares differ in length).
void gather_msgs(const Plan &p, int l, bool eps, int rank, std::vector<Msg> &out)
{
    const IRange me = p.own(l, eps, rank);
    out.clear();
    for (int r = 0; r < p.world; r++) {
        if (r == rank) continue;
        const IRange o = p.own(l, eps, r);
        out.push_back(Msg{r, true, me.lo, me.hi});
        out.push_back(Msg{r, false, o.lo, o.hi});
    }
}

// Record of a rank's communication calls, five numbers each: {what, peer, array tag, lo, hi}.
enum { LOG_HALO = 0, LOG_SEND = 1, LOG_RECV = 2, LOG_ALLREDUCE = 3, LOG_GATHER = 4 };

struct Transport {
    int rank = 0, world = 1;
    uint64_t bytes_sent = 0;
    // kept by the transport that moves nothing: what the rank would have sent and received
    virtual const std::vector<int64_t> *comm_log() const { return nullptr; }
    // Set by abort(): a rank of the group has failed.  Every later operation returns MG_ERR_COMM at
    // once instead of queueing transfers no peer will ever match.
    std::atomic<bool> aborted{false};
    virtual ~Transport() {}
    // Callable from any thread, also while the rank's own thread is inside a call: gives up the
    // group's pending and future communication so that nobody waits for a rank that is gone.
    virtual void abort() { aborted.store(true); }
    int alive() const
    {
        if (aborted.load()) return mgk::set_error(MG_ERR_COMM, "the group was given up after a rank failed");
        return MG_OK;
    }
    // Exchange the halo planes `h` of `s` around the owned range `own` with the slab neighbours;
    // ordered after everything enqueued on `compute` so far.  finish() makes `compute` wait for it.
    virtual int start(Slab &s, IRange own, Halo h, hipStream_t compute) = 0;
    int start(Slab &s, IRange own, int width, hipStream_t compute) { return start(s, own, Halo::both(width), compute); }
    virtual int finish(Slab &s, hipStream_t compute) = 0;
    // *dev_scalar = sum over ranks, same bits on every rank; ordered after everything enqueued on
    // `compute` so far.  `compute` itself does not wait: sum_ready() makes it, and must precede the
    // next write and every read of the scalar.
    virtual int allreduce(double *dev_scalar, hipStream_t compute) = 0;
    virtual int sum_ready(hipStream_t compute) = 0;
    // owned planes of level-l array `mine` of every rank -> every rank's whole array `full`
    virtual int allgather(const Plan &p, int l, bool eps, const Slab &mine, Slab &full, hipStream_t compute) = 0;
};

int copy_planes(Slab &dst, const View &src, IRange r, hipStream_t s)
{
    if (r.hi <= r.lo) return MG_OK;
    const double *from = src.p + (r.lo - src.i0) * src.si;
    HIP_TRY(hipMemcpyAsync(dst.plane(r.lo), from, (size_t)(r.hi - r.lo) * dst.plane_bytes(), hipMemcpyDefault, s));
    return MG_OK;
}

// ---- RCCL

struct RcclTransport : Transport {
    const Rccl *api = nullptr;
    ncclComm_t comm = nullptr;
    hipStream_t cs = nullptr;         // every RCCL call of this rank goes to this stream, in issue order
    hipEvent_t to_cs = nullptr, from_cs = nullptr;

    ~RcclTransport() override
    {
        if (cs && !aborted.load()) hipStreamSynchronize(cs);
        if (comm && api && !comm_gone.load()) api->CommDestroy(comm);
        if (to_cs) hipEventDestroy(to_cs);
        if (from_cs) hipEventDestroy(from_cs);
        if (sum_done) hipEventDestroy(sum_done);
        if (cs) hipStreamDestroy(cs);
    }
    std::atomic<bool> comm_gone{false};  // ncclCommAbort has run: it frees the communicator itself
    // ncclCommAbort ends the communicator's kernels, including those of transfers whose peer never
    // arrives, so streams waiting on them drain and the host calls blocked on those streams return.
    void abort() override
    {
        if (aborted.exchange(true)) return;
        if (comm && api && api->CommAbort && !comm_gone.exchange(true)) api->CommAbort(comm);
    }
    int init(int rank_, int world_, const uint8_t id_bytes[MG_UNIQUE_ID_BYTES])
    {
        rank = rank_;
        world = world_;
        api = rccl();
        if (!api) FAIL(MG_ERR_COMM, "librccl could not be loaded (or lacks the send/recv entry points)");
        static_assert(sizeof(ncclUniqueId) == MG_UNIQUE_ID_BYTES, "unique id size");
        ncclUniqueId id;
        memcpy(&id, id_bytes, sizeof(id));
        // Highest priority: a transfer is started while the interior launch of the same sweep is
        // queued right behind it on the kernel stream, and that launch fills every CU for its whole
        // run; RCCL's few workgroups must be placed first, not after it.
        int prio_least = 0, prio_greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        HIP_TRY(hipStreamCreateWithPriority(&cs, hipStreamNonBlocking, prio_greatest));
        HIP_TRY(hipEventCreateWithFlags(&to_cs, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&from_cs, hipEventDisableTiming));
        NCCL_TRY(api->CommInitRank(&comm, world, id, rank));
        // A world of one never exchanges anything; what it can do is prove that the library was
        // bound correctly and that its calls run on the communication stream the way start(),
        // allreduce() and allgather() issue them.
        if (world == 1) TRY(loopback_check());
        return MG_OK;
    }
    // Grouped send + receive of a block to this rank itself, then an all-reduce of one value, on the
    // communication stream ordered after and before a compute stream with the two events.
    int loopback_check()
    {
        constexpr int kCount = 1024;
        std::vector<double> host(2 * kCount, -1.0);
        for (int k = 0; k < kCount; k++) host[k] = 1.0 + 0.5 * k;
        double *buf = nullptr;
        HIP_TRY(hipMalloc((void **)&buf, 2 * kCount * sizeof(double)));
        hipStream_t compute = nullptr;
        hipError_t e = hipStreamCreateWithFlags(&compute, hipStreamNonBlocking);
        int rc = MG_OK;
        auto body = [&]() -> int {
            HIP_TRY(e);
            HIP_TRY(hipMemcpyAsync(buf, host.data(), 2 * kCount * sizeof(double), hipMemcpyHostToDevice, compute));
            TRY(order_after_compute(compute));
            NCCL_TRY(api->GroupStart());
            NCCL_TRY(api->Send(buf, kCount, ncclDouble, rank, comm, cs));
            NCCL_TRY(api->Recv(buf + kCount, kCount, ncclDouble, rank, comm, cs));
            NCCL_TRY(api->GroupEnd());
            NCCL_TRY(api->AllReduce(buf, buf, 1, ncclDouble, ncclSum, comm, cs));
            TRY(compute_waits(compute));
            HIP_TRY(hipMemcpyAsync(host.data(), buf, 2 * kCount * sizeof(double), hipMemcpyDeviceToHost, compute));
            HIP_TRY(hipStreamSynchronize(compute));
            for (int k = 0; k < kCount; k++)
                if (host[kCount + k] != 1.0 + 0.5 * k || host[k] != 1.0 + 0.5 * k)
                    FAIL(MG_ERR_COMM, "RCCL loopback check: element %d came back as %g / %g", k, host[k], host[kCount + k]);
            return MG_OK;
        };
        rc = body();
        if (compute) hipStreamDestroy(compute);
        hipFree(buf);
        return rc;
    }
    int order_after_compute(hipStream_t compute)
    {
        HIP_TRY(hipEventRecord(to_cs, compute));
        HIP_TRY(hipStreamWaitEvent(cs, to_cs, 0));
        return MG_OK;
    }
    int compute_waits(hipStream_t compute)
    {
        HIP_TRY(hipEventRecord(from_cs, cs));
        HIP_TRY(hipStreamWaitEvent(compute, from_cs, 0));
        return MG_OK;
    }
    int start(Slab &s, IRange own, Halo h, hipStream_t compute) override
    {
        TRY(alive());
        TRY(finish(s, compute));
        if (world == 1) return MG_OK;
        if (!s.done) HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        TRY(order_after_compute(compute));
        const size_t plane = (size_t)s.v.si;
        Msg msgs[4];
        const int n_msgs = halo_msgs(rank, world, own, h, msgs);
        NCCL_TRY(api->GroupStart());
        for (int k = 0; k < n_msgs; k++) {
            const Msg &m = msgs[k];
            const size_t count = (size_t)(m.hi - m.lo) * plane;
            if (m.send) {
                NCCL_TRY(api->Send(s.plane(m.lo), count, ncclDouble, m.peer, comm, cs));
                bytes_sent += count * 8;
            } else {
                NCCL_TRY(api->Recv(s.plane(m.lo), count, ncclDouble, m.peer, comm, cs));
            }
        }
        NCCL_TRY(api->GroupEnd());
        HIP_TRY(hipEventRecord(s.done, cs));
        s.pending = true;
        return MG_OK;
    }
    int finish(Slab &s, hipStream_t compute) override
    {
        TRY(alive());
        if (!s.pending) return MG_OK;
        HIP_TRY(hipStreamWaitEvent(compute, s.done, 0));
        s.pending = false;
        return MG_OK;
    }
    hipEvent_t sum_done = nullptr;
    bool sum_pending = false;
    int allreduce(double *x, hipStream_t compute) override
    {
        TRY(alive());
        if (world == 1) return MG_OK;
        if (!sum_done) HIP_TRY(hipEventCreateWithFlags(&sum_done, hipEventDisableTiming));
        TRY(order_after_compute(compute));
        NCCL_TRY(api->AllReduce(x, x, 1, ncclDouble, ncclSum, comm, cs));
        HIP_TRY(hipEventRecord(sum_done, cs));
        sum_pending = true;
        return MG_OK;
    }
    int sum_ready(hipStream_t compute) override
    {
        if (!sum_pending) return MG_OK;
        HIP_TRY(hipStreamWaitEvent(compute, sum_done, 0));
        sum_pending = false;
        return MG_OK;
    }
    int allgather(const Plan &p, int l, bool eps, const Slab &mine, Slab &full, hipStream_t compute) override
    {
        TRY(alive());
        const IRange me = p.own(l, eps, rank);
        TRY(copy_planes(full, mine.v, me, compute));
        if (world == 1) return MG_OK;
        TRY(order_after_compute(compute));
        std::vector<Msg> msgs;
        gather_msgs(p, l, eps, rank, msgs);
        NCCL_TRY(api->GroupStart());
        for (const Msg &m : msgs) {
            if (m.send) {
                NCCL_TRY(api->Send(mine.plane(m.lo), (size_t)(m.hi - m.lo) * mine.v.si, ncclDouble, m.peer, comm, cs));
                bytes_sent += (uint64_t)(m.hi - m.lo) * mine.v.si * 8;
            } else {
                NCCL_TRY(api->Recv(full.plane(m.lo), (size_t)(m.hi - m.lo) * full.v.si, ncclDouble, m.peer, comm, cs));
            }
        }
        NCCL_TRY(api->GroupEnd());
        return compute_waits(compute);
    }
};

}  // namespace

// ---- in-process: ranks are host threads of one process

// Meeting point of the ranks of one in-process group.
struct mg_local_group {
    int world = 1;
    bool hold_back = false;  // complete every started exchange only at its wait
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;  // a rank failed: everybody still waiting is released with an error
    std::vector<const View *> post;
    std::vector<double> value;

    // Returns false when the group was broken (a rank left with an error).
    bool barrier()
    {
        std::unique_lock<std::mutex> lock(m);
        if (broken) return false;
        const uint64_t gen = generation;
        if (++arrived == world) {
            arrived = 0;
            generation++;
            cv.notify_all();
            return true;
        }
        cv.wait(lock, [&] { return generation != gen || broken; });
        return !broken;
    }
    void abandon()
    {
        std::lock_guard<std::mutex> lock(m);
        broken = true;
        cv.notify_all();
    }
};

namespace {

struct LocalTransport : Transport {
    mg_local_group *g = nullptr;

    void abort() override
    {
        aborted.store(true);
        if (g) g->abandon();
    }

    int meet()
    {
        if (!g->barrier()) FAIL(MG_ERR_COMM, "another rank of the in-process group failed");
        return MG_OK;
    }
    // on any failure release the ranks waiting at a barrier instead of leaving them hung
    int guard(int rc)
    {
        if (rc) g->abandon();
        return rc;
    }
    int exchange_now(Slab &s, IRange own, Halo h, hipStream_t compute)
    {
        HIP_TRY(hipStreamSynchronize(compute));
        g->post[rank] = &s.v;
        TRY(meet());
        if (rank + 1 < world) {
            TRY(copy_planes(s, *g->post[rank + 1], IRange{own.hi, own.hi + h.rhi}, compute));
            bytes_sent += (uint64_t)h.shi * s.v.si * 8;
        }
        if (rank > 0) {
            TRY(copy_planes(s, *g->post[rank - 1], IRange{own.lo - h.rlo, own.lo}, compute));
            bytes_sent += (uint64_t)h.slo * s.v.si * 8;
        }
        HIP_TRY(hipStreamSynchronize(compute));
        return meet();  // nobody overwrites what a neighbour is still copying
    }
    int start(Slab &s, IRange own, Halo h, hipStream_t compute) override
    {
        TRY(finish(s, compute));
        if (world == 1) return MG_OK;
        if (g->hold_back) {
            s.pending = true;
            s.held_own = own;
            s.held = h;
            return MG_OK;
        }
        return guard(exchange_now(s, own, h, compute));
    }
    int finish(Slab &s, hipStream_t compute) override
    {
        if (!s.pending) return MG_OK;
        s.pending = false;
        return guard(exchange_now(s, s.held_own, s.held, compute));
    }
    int allreduce_now(double *x, hipStream_t compute)
    {
        double mine = 0;
        HIP_TRY(hipMemcpyAsync(&mine, x, sizeof(double), hipMemcpyDeviceToHost, compute));
        HIP_TRY(hipStreamSynchronize(compute));
        g->value[rank] = mine;
        TRY(meet());
        double sum = 0;
        for (int r = 0; r < world; r++) sum += g->value[r];  // rank order: the same bits everywhere
        TRY(meet());
        HIP_TRY(hipMemcpyAsync(x, &sum, sizeof(double), hipMemcpyHostToDevice, compute));
        HIP_TRY(hipStreamSynchronize(compute));
        return MG_OK;
    }
    int allreduce(double *x, hipStream_t compute) override
    {
        if (world == 1) return MG_OK;
        return guard(allreduce_now(x, compute));
    }
    int sum_ready(hipStream_t) override { return MG_OK; }
    int allgather_now(const Plan &p, int l, bool eps, const Slab &mine, Slab &full, hipStream_t compute)
    {
        HIP_TRY(hipStreamSynchronize(compute));
        g->post[rank] = &mine.v;
        TRY(meet());
        for (int r = 0; r < world; r++) {
            TRY(copy_planes(full, *g->post[r], p.own(l, eps, r), compute));
            if (r != rank) bytes_sent += (uint64_t)(p.own(l, eps, rank).hi - p.own(l, eps, rank).lo) * mine.v.si * 8;
        }
        HIP_TRY(hipStreamSynchronize(compute));
        return meet();
    }
    int allgather(const Plan &p, int l, bool eps, const Slab &mine, Slab &full, hipStream_t compute) override
    {
        if (world == 1) return copy_planes(full, mine.v, p.own(l, eps, 0), compute);
        return guard(allgather_now(p, l, eps, mine, full, compute));
    }
};

// ---- no transfers: ONE rank of a larger world, alone on its GPU, for timing that rank's share
//
// Every launch, its plane ranges and the order of the streams' dependencies are those of the rank
// in the full world; halo planes keep whatever they hold and the sums stay local, so the numbers
// a solve produces mean nothing.
//
// It keeps a record of the calls the RCCL transport would have made in its place -- built from the
// same message lists (halo_msgs, gather_msgs) -- so that the records of all ranks of a world, taken
// one rank at a time on one GPU, can be checked against each other: every send has its receive on
// the peer, in the same call, for the same planes of the same array.
struct SoloTransport : Transport {
    static constexpr size_t kMaxLog = 5 * 100000;
    std::vector<int64_t> log;
    const std::vector<int64_t> *comm_log() const override { return &log; }
    void note(int what, int peer, int tag, int64_t lo, int64_t hi)
    {
        if (log.size() + 5 > kMaxLog) return;
        const int64_t e[5] = {what, peer, tag, lo, hi};
        log.insert(log.end(), e, e + 5);
    }
    int start(Slab &s, IRange own, Halo h, hipStream_t) override
    {
        if (world == 1) return MG_OK;
        Msg msgs[4];
        const int n_msgs = halo_msgs(rank, world, own, h, msgs);
        note(LOG_HALO, -1, s.tag, own.lo, own.hi);
        for (int k = 0; k < n_msgs; k++) note(msgs[k].send ? LOG_SEND : LOG_RECV, msgs[k].peer, s.tag, msgs[k].lo, msgs[k].hi);
        return MG_OK;
    }
    int finish(Slab &, hipStream_t) override { return MG_OK; }
    int allreduce(double *, hipStream_t) override
    {
        if (world > 1) note(LOG_ALLREDUCE, -1, -1, 0, 1);
        return MG_OK;
    }
    int sum_ready(hipStream_t) override { return MG_OK; }
    int allgather(const Plan &p, int l, bool eps, const Slab &mine, Slab &full, hipStream_t compute) override
    {
        if (world > 1) {
            std::vector<Msg> msgs;
            gather_msgs(p, l, eps, rank, msgs);
            const IRange me = p.own(l, eps, rank);
            note(LOG_GATHER, -1, mine.tag, me.lo, me.hi);
            for (const Msg &m : msgs) note(m.send ? LOG_SEND : LOG_RECV, m.peer, m.send ? mine.tag : full.tag, m.lo, m.hi);
        }
        return copy_planes(full, mine.v, p.own(l, eps, rank), compute);
    }
};

}  // namespace

// ---------------------------------------------------------------------------- one rank

struct mg_dist {
    Plan plan;
    int rank = 0, world = 1, L = 0, cut = 0, device = 0;
    bool overlap = true;
    // deep halos on levels 1..cut-1 (see Plan::deep_ranges): planes of eps_l held after the last
    // sweep / formed before the first, and the planes of R_l moved once per cycle for them
    bool deep = false;
    IRange dlast[MG_MAX_LEVELS], dfirst[MG_MAX_LEVELS];
    Halo dhalo[MG_MAX_LEVELS];
    int nu[MG_MAX_LEVELS] = {};
    Coef c[MG_MAX_LEVELS];
    double omega = 1.4;
    hipStream_t stream = nullptr;
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    Transport *tr = nullptr;
    Slab phi, phi_alt, b;
    Slab R[MG_MAX_LEVELS], eps[MG_MAX_LEVELS], eps_alt[MG_MAX_LEVELS];     // slabs: R of levels 1..cut, eps of 1..cut-1
    Slab cR[MG_MAX_LEVELS], ceps[MG_MAX_LEVELS], ceps_alt[MG_MAX_LEVELS];  // whole levels cut..L, on every rank
    std::vector<void *> blocks;  // every device allocation of this rank
    double *partials = nullptr;
    int64_t n_partials = 0;
    double *d_sum = nullptr;  // this rank's, then the global, sum of squared residuals
    double *h_sum = nullptr;  // pinned
    double n_nodes = 0;
    double dh[3] = {0, 0, 0};
    double *ef = nullptr;  // E on the owned planes, dense [plane][nj][nk][3]; allocated by the first mg_dist_compute_ef
    bool have_ef = false;
    bool correction_pending = false, have_rho = false, have_phi = false;
    double last_ms = 0;

    int alloc(Slab *s, int l, bool eps_kind, IRange st)
    {
        int64_t d[3];
        plan.dims(l, eps_kind, d);
        View v;
        v.d[0] = st.hi - st.lo;
        v.d[1] = d[1];
        v.d[2] = d[2];
        v.sj = mgk::pitch_for(d[2]);
        v.si = v.sj * d[1];
        v.i0 = st.lo;
        const size_t bytes = (size_t)v.d[0] * v.si * sizeof(double);
        HIP_TRY(hipMalloc((void **)&v.p, bytes));
        blocks.push_back(v.p);
        HIP_TRY(hipMemsetAsync(v.p, 0, bytes, stream));
        s->v = v;
        return MG_OK;
    }
    IRange own(int l, bool eps_kind) const { return plan.own(l, eps_kind, rank); }
    // eps of level l as the level above prolongs from it: this rank's slab, or for the first whole
    // level the rank's own copy of the whole array
    Slab &eps_of(int l) { return l >= cut ? ceps[l] : eps[l]; }

    // plane ranges of an owned slab: the two planes next to each existing neighbour (what a halo
    // exchange of width 2 sends) and the interior between them
    void split(IRange o, IRange edges[2], int *n_edges, IRange *inner) const
    {
        const int64_t a = rank == 0 ? o.lo : o.lo + H_U, z = rank == world - 1 ? o.hi : o.hi - H_U;
        *n_edges = 0;
        if (a > o.lo) edges[(*n_edges)++] = IRange{o.lo, a};
        if (z < o.hi) edges[(*n_edges)++] = IRange{z, o.hi};
        *inner = IRange{a, z};
    }

    // One launch group of a sweep: planes r (and r2) of *alt = sweep(*u), or in place.
    // Returns 1 when the result landed in alt, 0 when in place, < 0 on error.
    int run_sweep(Slab &u, Slab &alt, const Slab &rhs, const Slab *fold, bool correct, const int64_t n[3], int l,
                  IRange r, IRange r2)
    {
        View vu = u.v, va = alt.v;
        const double *before = vu.p;
        int rc;
        // (the slabs of eps_l are zero outside 0..n_l-2 like the whole arrays: see kLoadProlongZero)
        if (fold) rc = mgk::prolong_smooth_rb(stream, fold->v, correct, &vu, &va, rhs.v, n, c[l], omega, 1, r, r2, true);
        else rc = mgk::smooth_rb(stream, &vu, &va, rhs.v, n, c[l], omega, 1, r, r2);
        if (rc < 0) return mgk::set_error(MG_ERR_STATE, "a two-range sweep needs the out-of-place kernel");
        return vu.p != before ? 1 : 0;
    }

    // nu[l] red-black sweeps of a distributed level, one halo exchange after each.  With `ec`
    // (eps of level l+1, halos current) the first sweep is preceded by the prolongation from it:
    // u := P(ec), or u -= P(ec) when `correct` -- folded into that sweep's loads where possible.
    int sweeps(int l, Slab *ec, bool correct)
    {
        int64_t n[3];
        plan.level_n(l, n);
        for (int sweep = 0; sweep < nu[l]; sweep++) {
            Slab &u = l == 0 ? phi : eps[l];
            Slab &alt = l == 0 ? phi_alt : eps_alt[l];
            const Slab &rhs = l == 0 ? b : R[l];
            const IRange o = own(l, l != 0);
            const Slab *fold = nullptr;
            if (sweep == 0 && ec) {
                TRY(tr->finish(*ec, stream));
                if (alt.v.p && mgk::onload_folds(u.v, correct)) {
                    fold = ec;  // the operation rides on this sweep's loads; u stays as it is
                } else {
                    // applied in place once, on every stored plane this sweep reads
                    TRY(tr->finish(u, stream));
                    IRange pr{o.lo - H_U < u.v.i0 ? u.v.i0 : o.lo - H_U, o.hi + H_U > u.v.i_end() ? u.v.i_end() : o.hi + H_U};
                    if (correct) mgk::prolong_correct(stream, ec->v, u.v, n, pr);
                    else mgk::prolong(stream, ec->v, u.v, n, pr);
                }
            }
            IRange edges[2], inner;
            int n_edges;
            split(o, edges, &n_edges, &inner);
            const bool oop = alt.v.p && mgk::sweep_is_out_of_place(u.v);
            const bool ovl = world > 1 && overlap && oop && inner.hi - inner.lo >= 2 * H_U;
            TRY(tr->finish(u, stream));
            if (ovl && l == 0 && sweep == nu[l] - 1) {
                // The last fine sweep is followed by the residual pass, whose interior part reads no
                // halo plane: the sweep runs as ONE launch and its exchange under that interior part
                // (an edge launch re-reads four lead-in planes per edge).
                const int rc = run_sweep(u, alt, rhs, fold, correct, n, l, o, IRange{0, 0});
                if (rc < 0) return rc;
                if (rc != 1) FAIL(MG_ERR_STATE, "out-of-place sweep expected");
                TRY(tr->start(alt, o, H_U, stream));
                std::swap(u, alt);
                continue;
            }
            int swapped;
            if (ovl) {
                // edge planes need the halos of the previous sweep; the interior does not
                int rc = run_sweep(u, alt, rhs, fold, correct, n, l, edges[0], n_edges == 2 ? edges[1] : IRange{0, 0});
                if (rc < 0) return rc;
                if (rc != 1) FAIL(MG_ERR_STATE, "out-of-place sweep expected");
                TRY(tr->start(alt, o, H_U, stream));
                rc = run_sweep(u, alt, rhs, fold, correct, n, l, inner, IRange{0, 0});
                if (rc < 0) return rc;
                if (rc != 1) FAIL(MG_ERR_STATE, "out-of-place sweep expected");
                swapped = 1;
            } else {
                swapped = run_sweep(u, alt, rhs, fold, correct, n, l, o, IRange{0, 0});
                if (swapped < 0) return swapped;
            }
            if (swapped) std::swap(u, alt);
            if (!ovl) {
                TRY(tr->start(u, o, H_U, stream));
                TRY(tr->finish(u, stream));
            }
        }
        HIP_TRY(hipGetLastError());
        return MG_OK;
    }

    // The nu[l] sweeps of a distributed level on deep halos: eps_l = P(eps_{l+1}) on dfirst[l], then
    // sweep s on dlast[l] widened by 2*(nu-s) planes per side.  No exchange: the planes beyond the
    // owned ones are computed here as the neighbour computes them, from R_l planes received once.
    int sweeps_deep(int l, Slab &below)
    {
        int64_t n[3];
        plan.level_n(l, n);
        const int64_t top = plan.planes(l, true);
        Slab &u = eps[l], &alt = eps_alt[l];
        const Slab &rhs = R[l];
        TRY(tr->finish(below, stream));
        TRY(tr->finish(R[l], stream));
        const auto out = [&](int s) { return widen(dlast[l], 2 * (int64_t)(nu[l] - s), top); };
        int s = 1;
        if (nu[l] > 0 && alt.v.p && mgk::onload_folds(u.v, false)) {
            const int rc = run_sweep(u, alt, rhs, &below, false, n, l, out(1), IRange{0, 0});
            if (rc < 0) return rc;
            if (rc != 1) FAIL(MG_ERR_STATE, "out-of-place sweep expected");
            std::swap(u, alt);
            s = 2;
        } else {
            mgk::prolong(stream, below.v, u.v, n, IRange{dfirst[l].lo, dfirst[l].hi > n[0] ? n[0] : dfirst[l].hi});
        }
        while (s <= nu[l]) {
            View vu = u.v, va = alt.v;
            const double *before = vu.p;
            if (s + 1 <= nu[l] && mgk::sweeps_in_pairs(vu, va, rhs.v, n, out(s + 1))) {
                mgk::smooth_rb(stream, &vu, &va, rhs.v, n, c[l], omega, 2, out(s + 1));
                s += 2;
            } else {
                mgk::smooth_rb(stream, &vu, &va, rhs.v, n, c[l], omega, 1, out(s));
                s += 1;
            }
            if (vu.p != before) std::swap(u, alt);
        }
        HIP_TRY(hipGetLastError());
        return MG_OK;
    }

    // phi -= P(eps_1) of the last iteration, if it still waits for the next fine sweep
    int flush_correction()
    {
        if (!correction_pending) return MG_OK;
        // on every stored plane, halos included: eps_1's halos are current
        TRY(tr->finish(eps_of(1), stream));
        TRY(tr->finish(phi, stream));
        mgk::prolong_correct(stream, eps_of(1).v, phi.v, plan.n0, plan.stored(0, false, rank, H_U));
        HIP_TRY(hipGetLastError());
        correction_pending = false;
        return MG_OK;
    }

    // Steps 1)-2b) of the reference iteration (+ the restriction to level 1, which shares the
    // pass).  Leaves the global sum of squared residuals in *d_sum.
    int smooth_and_residual()
    {
        if (correction_pending && nu[0] > 0) {
            TRY(sweeps(0, &eps_of(1), true));  // step 6) of the previous iteration rides along
            correction_pending = false;
        } else {
            TRY(flush_correction());
            TRY(sweeps(0, nullptr, false));
        }
        const IRange o = own(0, false);
        IRange edges[2], inner, parts[2], parts2[2] = {IRange{0, 0}, IRange{0, 0}};
        int n_edges, n_parts = 0, wait_at = 0;
        split(o, edges, &n_edges, &inner);
        // slab boundaries are multiples of 2^cut, so every range below starts on an even plane
        if (world > 1 && overlap && inner.hi - inner.lo >= 2 * H_U) {
            // the interior needs no halo plane: it runs while the last sweep's exchange completes;
            // the planes next to both neighbours follow in ONE launch
            parts[n_parts++] = inner;
            if (n_edges > 0) {
                parts[n_parts] = edges[0];
                if (n_edges == 2) parts2[n_parts] = edges[1];
                n_parts++;
            }
            wait_at = 1;
        } else {
            parts[n_parts++] = o;
        }
        int64_t used = 0;
        View none = phi.v;
        none.p = nullptr;
        for (int idx = 0; idx < n_parts; idx++) {
            if (idx == wait_at) TRY(tr->finish(phi, stream));
            const int64_t np = mgk::residual_restrict_partials(plan.n0, parts[idx], parts2[idx]);
            if (used + np > n_partials) FAIL(MG_ERR_STATE, "partial-sum buffer too small");
            mgk::residual_restrict(stream, phi.v, b.v, none, R[1].v, plan.n0, c[0], partials + used, parts[idx],
                                   parts2[idx]);
            used += np;
        }
        TRY(tr->sum_ready(stream));  // the previous cycle's all-reduce has read the scalar
        mgk::sum_partials(stream, partials, used, d_sum);
        HIP_TRY(hipGetLastError());
        TRY(tr->finish(phi, stream));
        return tr->allreduce(d_sum, stream);
    }

    // Levels cut..L on whole arrays (every rank its own): restrict down, smooth the coarsest,
    // (prolong, smooth) back up to level `cut`.
    //
    // Of eps_cut a rank only ever reads the planes under its own slab of level cut-1, and a sweep
    // needs its input two planes further out on either side than it produces (red on the next
    // plane, black on the plane itself).  So sweep s of nu runs on those planes widened by
    // 2*(nu-s), the prolongation before the first sweep on the range widened by 2*nu, and the
    // level below is needed on the planes that prolongation reads, and so on down.  A level is
    // cut down this way when that leaves at most half of it (1025^3 over 8 ranks: 40 of the 129
    // planes of level 3 at the start, 24 at the end); every value computed is the one the whole
    // sweep would give, since nothing outside a range is read later.
    int coarse_levels()
    {
        int64_t n[3];
        for (int l = cut; l < L; l++) {
            plan.level_n(l, n);
            mgk::restrict27(stream, cR[l].v, cR[l + 1].v, n, IRange{0, plan.planes(l + 1, false)});
        }
        IRange last[MG_MAX_LEVELS], first[MG_MAX_LEVELS];  // planes of eps_l after its last sweep / before its first
        // planes of level cut-1 formed from eps_cut
        IRange need = deep ? dfirst[cut - 1] : plan.stored(cut - 1, cut > 1, rank, H_U);
        for (int l = cut; l <= L; l++) {
            const int64_t top = plan.planes(l, true);
            // the prolongation onto fine planes [lo, hi) reads coarse planes lo/2 .. hi/2
            last[l] = IRange{need.lo >> 1, (need.hi >> 1) + 1 > top ? top : (need.hi >> 1) + 1};
            first[l] = widen(last[l], 2 * (int64_t)nu[l], top);
            if (l == L || world == 1 || 2 * (first[l].hi - first[l].lo) > top) last[l] = first[l] = IRange{0, top};
            need = first[l];
        }
        for (int l = L; l >= cut; l--) {
            plan.level_n(l, n);
            const int64_t top = plan.planes(l, true);
            if (l < L) mgk::prolong(stream, ceps[l + 1].v, ceps[l].v, n, IRange{first[l].lo, first[l].hi > n[0] ? n[0] : first[l].hi});
            View vu = ceps[l].v, va = ceps_alt[l].v;
            const double *before = vu.p;
            if (first[l].lo == 0 && first[l].hi == top) {
                mgk::smooth_rb(stream, &vu, &va, cR[l].v, n, c[l], omega, nu[l], IRange{0, top});
            } else {
                for (int s = 1; s <= nu[l]; s++)
                    mgk::smooth_rb(stream, &vu, &va, cR[l].v, n, c[l], omega, 1, widen(last[l], 2 * (int64_t)(nu[l] - s), top));
            }
            if (vu.p != before) std::swap(ceps[l], ceps_alt[l]);
        }
        HIP_TRY(hipGetLastError());
        return MG_OK;
    }

    // Steps 3)-5): restrict down to the cut level, all-gather it, solve the whole levels (every
    // rank), then (prolong, smooth) up.  Step 6), the correction of phi, is left pending for the
    // next iteration's first sweep.
    int coarse_correction()
    {
        int64_t n[3];
        for (int l = 1; l < cut; l++) {
            // R_l's halo plane below feeds the restriction; both feed the sweeps' red halo planes
            plan.level_n(l, n);
            const IRange oc = own(l + 1, false);
            // (with deep halos the plane above the slab comes later, with the others the sweeps reach)
            TRY(tr->start(R[l], own(l, false), deep ? Halo{H_RHS, 0, H_RHS, 0} : Halo::both(H_RHS), stream));
            if (world > 1 && overlap && rank > 0 && oc.hi - oc.lo >= 2) {
                // only the first coarse plane reads the halo plane: the others go first, under the exchange
                mgk::restrict27(stream, R[l].v, R[l + 1].v, n, IRange{oc.lo + 1, oc.hi});
                TRY(tr->finish(R[l], stream));
                mgk::restrict27(stream, R[l].v, R[l + 1].v, n, IRange{oc.lo, oc.lo + 1});
            } else {
                TRY(tr->finish(R[l], stream));
                mgk::restrict27(stream, R[l].v, R[l + 1].v, n, oc);
            }
        }
        HIP_TRY(hipGetLastError());
        TRY(tr->allgather(plan, cut, false, R[cut], cR[cut], stream));
        if (deep) {
            // the planes of R_l the sweeps of level l reach beyond the owned ones: one message per
            // side and level, queued behind the all-gather (which the whole levels wait for) and in
            // the order they are needed; they travel while the whole levels are solved
            for (int l = cut - 1; l >= 1; l--) TRY(tr->start(R[l], own(l, false), dhalo[l], stream));
            TRY(coarse_levels());
            for (int l = cut - 1; l >= 1; l--) TRY(sweeps_deep(l, eps_of(l + 1)));
            correction_pending = true;
            return MG_OK;
        }
        TRY(coarse_levels());
        for (int l = cut - 1; l >= 1; l--) {
            Slab &below = eps_of(l + 1);
            if (nu[l] > 0) {
                // eps_l = P(eps_{l+1}) as the first sweep's input, on halo planes too (eps_{l+1}'s
                // halos are current, or it is a whole array), so the prolongation needs no exchange
                TRY(sweeps(l, &below, false));
            } else {
                TRY(tr->finish(below, stream));
                plan.level_n(l, n);
                mgk::prolong(stream, below.v, eps[l].v, n, plan.stored(l, true, rank, H_U));
                HIP_TRY(hipGetLastError());
            }
        }
        correction_pending = true;
        return MG_OK;
    }

    int read_sum(double *out)
    {
        TRY(tr->sum_ready(stream));
        HIP_TRY(hipMemcpyAsync(h_sum, d_sum, sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        *out = *h_sum;
        return MG_OK;
    }
    double l2_of(double sum) const { return std::sqrt(sum / n_nodes); }
};

namespace {

int check_rank(mg_dist *d, bool need_fields)
{
    if (!d) FAIL(MG_ERR_INVALID, "null rank handle");
    if (need_fields && (!d->have_rho || !d->have_phi))
        FAIL(MG_ERR_STATE, "rho and phi planes must be set before solving");
    HIP_TRY(hipSetDevice(d->device));
    return MG_OK;
}

// dense planes [lo,hi) (rows of nk doubles) <-> the slab, staged through the idle twin of phi
int upload_planes(mg_dist *d, Slab &dst, int64_t lo, int64_t hi, const double *dense, bool div_eps0)
{
    if (!dense || lo < dst.v.i0 || hi > dst.v.i_end() || lo >= hi)
        FAIL(MG_ERR_INVALID, "planes [%lld,%lld) outside the stored range [%lld,%lld)", (long long)lo, (long long)hi,
             (long long)dst.v.i0, (long long)dst.v.i_end());
    const int64_t cap = d->phi_alt.v.d[0];  // planes the staging area holds
    for (int64_t a = lo; a < hi; a += cap) {
        const int64_t z = a + cap < hi ? a + cap : hi;
        const size_t bytes = (size_t)(z - a) * dst.v.d[1] * dst.v.d[2] * sizeof(double);
        View part = dst.v;
        part.p = dst.plane(a);
        part.i0 = a;
        part.d[0] = z - a;
        const double *src = dense + (size_t)(a - lo) * dst.v.d[1] * dst.v.d[2];
        if (part.sj == part.d[2]) {
            HIP_TRY(hipMemcpyAsync(part.p, src, bytes, hipMemcpyHostToDevice, d->stream));
            if (div_eps0) mgk::div_eps0(d->stream, part);
        } else {
            HIP_TRY(hipMemcpyAsync(d->phi_alt.v.p, src, bytes, hipMemcpyHostToDevice, d->stream));
            mgk::repitch_in(d->stream, part, d->phi_alt.v.p, div_eps0);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(d->stream));
    }
    return MG_OK;
}

}  // namespace

// The allocations of mg_dist_create for one rank, as arithmetic (keep the two lists alike).
long long mgk::dist_rank_device_bytes(const long long n_[3], int depth, int world, int coarse_cut, int rank,
                                      const int *nu)
{
    const int64_t n[3] = {n_[0], n_[1], n_[2]};
    Plan P;
    if (P.init(n, depth, world, coarse_cut)) return -1;
    IRange dlast[MG_MAX_LEVELS], dfirst[MG_MAX_LEVELS];
    const bool deep = P.deep_fits(nu);
    if (deep) P.deep_ranges(rank, nu, dlast, dfirst);
    auto bytes = [&](int l, bool eps_kind, IRange st) -> long long {
        int64_t d[3];
        P.dims(l, eps_kind, d);
        return (long long)(st.hi - st.lo) * d[1] * mgk::pitch_for(d[2]) * (long long)sizeof(double);
    };
    long long sum = 2 * bytes(0, false, P.stored(0, false, rank, H_U)) + bytes(0, false, P.stored(0, false, rank, H_RHS));
    for (int l = 1; l <= P.cut; l++) {
        const bool wide = deep && l < P.cut;
        sum += bytes(l, false, wide ? P.deep_rhs(l, dfirst[l]) : P.stored(l, false, rank, H_RHS));
        if (l < P.cut) sum += 2 * bytes(l, true, wide ? dfirst[l] : P.stored(l, true, rank, H_U));
    }
    for (int l = P.cut; l <= depth; l++)
        sum += bytes(l, false, IRange{0, P.planes(l, false)}) + 2 * bytes(l, true, IRange{0, P.planes(l, true)});
    const IRange o = P.own(0, false, rank);
    sum += (mgk::residual_restrict_partials(n, o) + 2 * mgk::residual_restrict_partials(n, IRange{o.lo, o.lo + 2}) + 1) *
           (long long)sizeof(double);  // an upper bound of the partial sums of the split residual pass
    return sum;
}

extern "C" {

int mg_dist_unique_id(uint8_t out[MG_UNIQUE_ID_BYTES])
{
    if (!out) FAIL(MG_ERR_INVALID, "null argument");
    const Rccl *api = rccl();
    if (!api) FAIL(MG_ERR_COMM, "librccl could not be loaded");
    ncclUniqueId id;
    NCCL_TRY(api->GetUniqueId(&id));
    memcpy(out, &id, MG_UNIQUE_ID_BYTES);
    return MG_OK;
}

int mg_local_group_create(int world, int hold_back, mg_local_group **out)
{
    if (!out || world < 1 || world > MG_MAX_GPUS) FAIL(MG_ERR_INVALID, "bad argument");
    mg_local_group *g = new (std::nothrow) mg_local_group();
    if (!g) FAIL(MG_ERR_NOMEM, "host allocation failed");
    g->world = world;
    g->hold_back = hold_back != 0;
    g->post.assign(world, nullptr);
    g->value.assign(world, 0.0);
    *out = g;
    return MG_OK;
}

int mg_local_group_destroy(mg_local_group *g)
{
    delete g;
    return MG_OK;
}

int mg_dist_destroy(mg_dist *d)
{
    if (!d) return MG_OK;
    hipSetDevice(d->device);
    if (d->stream) hipStreamSynchronize(d->stream);
    delete d->tr;  // the RCCL transport drains its stream first
    Slab *all[] = {&d->phi, &d->phi_alt, &d->b};
    for (Slab *s : all)
        if (s->done) hipEventDestroy(s->done);
    for (int l = 0; l < MG_MAX_LEVELS; l++)
        for (Slab *s : {&d->R[l], &d->eps[l], &d->eps_alt[l]})
            if (s->done) hipEventDestroy(s->done);
    for (void *p : d->blocks) hipFree(p);
    if (d->h_sum) hipHostFree(d->h_sum);
    if (d->ev_a) hipEventDestroy(d->ev_a);
    if (d->ev_b) hipEventDestroy(d->ev_b);
    if (d->stream) hipStreamDestroy(d->stream);
    delete d;
    return MG_OK;
}

int mg_dist_create(const mg_dist_config *cfg, mg_dist **out)
{
    if (!cfg || !out) FAIL(MG_ERR_INVALID, "null argument");
    *out = nullptr;
    const mg_config &base = cfg->base;
    if (cfg->world < 1 || cfg->world > MG_MAX_GPUS || cfg->rank < 0 || cfg->rank >= cfg->world)
        FAIL(MG_ERR_INVALID, "rank %d / world %d out of range", cfg->rank, cfg->world);
    if (base.depth < 1 || base.depth >= MG_MAX_LEVELS) FAIL(MG_ERR_INVALID, "depth %d out of range 1..%d", base.depth, MG_MAX_LEVELS - 1);
    if (base.flags & ~MG_FLAG_FAST_SMOOTHER)
        FAIL(MG_ERR_INVALID, "a partitioned solve supports no MG_FLAG_* option but MG_FLAG_FAST_SMOOTHER");
    for (int a = 0; a < 3; a++) {
        if (base.n[a] < 3 || base.n[a] > 65535) FAIL(MG_ERR_INVALID, "n[%d]=%lld out of range", a, (long long)base.n[a]);
        if (!(base.dh[a] > 0)) FAIL(MG_ERR_INVALID, "dh[%d] must be positive", a);
    }
    for (int l = 0; l <= base.depth; l++)
        if (base.nu[l] < 0 || base.nu[l] > 100000) FAIL(MG_ERR_INVALID, "nu[%d]=%d out of range", l, base.nu[l]);
    if (cfg->transport == MG_TRANSPORT_LOCAL && cfg->world > 1 && (!cfg->local_group || cfg->local_group->world != cfg->world))
        FAIL(MG_ERR_INVALID, "the in-process transport needs an mg_local_group of the same world size");
    if (cfg->transport != MG_TRANSPORT_LOCAL && cfg->transport != MG_TRANSPORT_RCCL && cfg->transport != MG_TRANSPORT_NONE)
        FAIL(MG_ERR_INVALID, "unknown transport %d", cfg->transport);
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) FAIL(MG_ERR_NO_DEVICE, "no HIP device visible");
    if (base.device < 0 || base.device >= count) FAIL(MG_ERR_INVALID, "device %d out of range (have %d)", base.device, count);
    HIP_TRY(hipSetDevice(base.device));

    mg_dist *d = new (std::nothrow) mg_dist();
    if (!d) FAIL(MG_ERR_NOMEM, "host allocation failed");
#define CREATE_TRY(expr)        \
    do {                        \
        int rc_ = (expr);       \
        if (rc_) {              \
            mg_dist_destroy(d); \
            return rc_;         \
        }                       \
    } while (0)
    auto hip_ok = [](hipError_t e, const char *what) -> int {
        if (e == hipSuccess) return MG_OK;
        (void)hipGetLastError();  // reported here; not left behind for a later call's check
        return mgk::set_error(e == hipErrorOutOfMemory ? MG_ERR_NOMEM : MG_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    };
    d->rank = cfg->rank;
    d->world = cfg->world;
    d->device = base.device;
    d->L = base.depth;
    d->omega = base.omega;
    d->overlap = (cfg->no_overlap & 1) == 0;
    CREATE_TRY(d->plan.init(base.n, base.depth, cfg->world, cfg->coarse_cut));
    d->cut = d->plan.cut;
    d->deep = (cfg->no_overlap & 2) == 0 && d->plan.deep_fits(base.nu);
    if (d->deep) {
        const Plan &P = d->plan;
        P.deep_ranges(cfg->rank, base.nu, d->dlast, d->dfirst);
        for (int l = 1; l < d->cut; l++) d->dhalo[l] = deep_halo(P, cfg->rank, l, base.nu);
    }
    d->n_nodes = (double)base.n[0] * (double)base.n[1] * (double)base.n[2];
    for (int a = 0; a < 3; a++) d->dh[a] = base.dh[a];
    for (int l = 0; l <= d->L; l++) {
        d->nu[l] = base.nu[l];
        double k[4];
        CREATE_TRY(mg_level_coef(base.dh, l, k));
        d->c[l] = mgk::make_coef(k[0], k[1], k[2], k[3]);
        if ((base.flags & MG_FLAG_FAST_SMOOTHER) && l >= 1) d->c[l] = mgk::with_fast_smoother(d->c[l], base.omega);
    }
    CREATE_TRY(hip_ok(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking), "hipStreamCreate"));
    CREATE_TRY(hip_ok(hipEventCreate(&d->ev_a), "hipEventCreate"));
    CREATE_TRY(hip_ok(hipEventCreate(&d->ev_b), "hipEventCreate"));
    // the communicator first: its creation is the one step that waits for the other ranks, and a
    // rank that failed later (out of memory, say) must not leave them waiting here
    if (cfg->transport == MG_TRANSPORT_RCCL) {  // a world of one still creates its communicator (and checks it)
        RcclTransport *t = new (std::nothrow) RcclTransport();
        if (!t) {
            mg_dist_destroy(d);
            FAIL(MG_ERR_NOMEM, "host allocation failed");
        }
        d->tr = t;
        CREATE_TRY(t->init(cfg->rank, cfg->world, cfg->unique_id));
    } else if (cfg->transport == MG_TRANSPORT_NONE) {
        SoloTransport *t = new (std::nothrow) SoloTransport();
        if (!t) {
            mg_dist_destroy(d);
            FAIL(MG_ERR_NOMEM, "host allocation failed");
        }
        t->rank = cfg->rank;
        t->world = cfg->world;
        d->tr = t;
    } else {
        LocalTransport *t = new (std::nothrow) LocalTransport();
        if (!t) {
            mg_dist_destroy(d);
            FAIL(MG_ERR_NOMEM, "host allocation failed");
        }
        t->rank = cfg->rank;
        t->world = cfg->world;
        t->g = cfg->local_group;
        d->tr = t;
    }
    const Plan &P = d->plan;
    const int r = d->rank;
    CREATE_TRY(d->alloc(&d->phi, 0, false, P.stored(0, false, r, H_U)));
    CREATE_TRY(d->alloc(&d->phi_alt, 0, false, P.stored(0, false, r, H_U)));
    CREATE_TRY(d->alloc(&d->b, 0, false, P.stored(0, false, r, H_RHS)));
    for (int l = 1; l <= d->cut; l++) {
        const bool wide = d->deep && l < d->cut;
        CREATE_TRY(d->alloc(&d->R[l], l, false, wide ? P.deep_rhs(l, d->dfirst[l]) : P.stored(l, false, r, H_RHS)));
        if (l < d->cut) {
            CREATE_TRY(d->alloc(&d->eps[l], l, true, wide ? d->dfirst[l] : P.stored(l, true, r, H_U)));
            CREATE_TRY(d->alloc(&d->eps_alt[l], l, true, wide ? d->dfirst[l] : P.stored(l, true, r, H_U)));
        }
    }
    for (int l = d->cut; l <= d->L; l++) {
        CREATE_TRY(d->alloc(&d->cR[l], l, false, IRange{0, P.planes(l, false)}));
        CREATE_TRY(d->alloc(&d->ceps[l], l, true, IRange{0, P.planes(l, true)}));
        CREATE_TRY(d->alloc(&d->ceps_alt[l], l, true, IRange{0, P.planes(l, true)}));
    }
    // array tags: 16 * level + {0 phi, 1 its out-of-place twin, 2 right-hand side, 3 R, 4 eps, 5 its twin,
    // 6 whole R, 7 whole eps, 8 its twin}
    d->phi.tag = 0;
    d->phi_alt.tag = 1;
    d->b.tag = 2;
    for (int l = 1; l <= d->L; l++) {
        d->R[l].tag = 16 * l + 3;
        d->eps[l].tag = 16 * l + 4;
        d->eps_alt[l].tag = 16 * l + 5;
        d->cR[l].tag = 16 * l + 6;
        d->ceps[l].tag = 16 * l + 7;
        d->ceps_alt[l].tag = 16 * l + 8;
    }
    {
        // partial sums of the fine residual: the owned range, or interior + edges
        const IRange o = d->own(0, false);
        IRange edges[2], inner;
        int n_edges;
        d->split(o, edges, &n_edges, &inner);
        int64_t need = mgk::residual_restrict_partials(P.n0, o);
        int64_t parts = inner.hi > inner.lo ? mgk::residual_restrict_partials(P.n0, inner) : 0;
        if (n_edges > 0) parts += mgk::residual_restrict_partials(P.n0, edges[0], n_edges == 2 ? edges[1] : IRange{0, 0});
        d->n_partials = need > parts ? need : parts;
        CREATE_TRY(hip_ok(hipMalloc((void **)&d->partials, (size_t)(d->n_partials + 1) * sizeof(double)), "hipMalloc(partials)"));
        d->blocks.push_back(d->partials);
        d->d_sum = d->partials + d->n_partials;
    }
    CREATE_TRY(hip_ok(hipHostMalloc((void **)&d->h_sum, sizeof(double), hipHostMallocDefault), "hipHostMalloc"));
    *d->h_sum = 0;
    CREATE_TRY(hip_ok(hipStreamSynchronize(d->stream), "hipStreamSynchronize"));
#undef CREATE_TRY
    *out = d;
    return MG_OK;
}

int mg_dist_plan(mg_dist *d, int32_t *cut, int64_t own[2], int64_t stored_phi[2], int64_t stored_rho[2])
{
    if (!d) FAIL(MG_ERR_INVALID, "null rank handle");
    if (cut) *cut = d->cut;
    const IRange o = d->own(0, false);
    if (own) { own[0] = o.lo; own[1] = o.hi; }
    if (stored_phi) { stored_phi[0] = d->phi.v.i0; stored_phi[1] = d->phi.v.i_end(); }
    if (stored_rho) { stored_rho[0] = d->b.v.i0; stored_rho[1] = d->b.v.i_end(); }
    return MG_OK;
}

int mg_dist_deep_halos(mg_dist *d, int32_t *deep)
{
    if (!d || !deep) FAIL(MG_ERR_INVALID, "null argument");
    *deep = d->deep ? 1 : 0;
    return MG_OK;
}

int mg_dist_comm_log(mg_dist *d, int64_t *entries, int64_t capacity, int64_t *count)
{
    if (!d || !count || (capacity > 0 && !entries)) FAIL(MG_ERR_INVALID, "null argument");
    const std::vector<int64_t> *log = d->tr ? d->tr->comm_log() : nullptr;
    const int64_t have = log ? (int64_t)(log->size() / 5) : 0;
    *count = have;
    const int64_t take = have < capacity ? have : capacity;
    if (take > 0) memcpy(entries, log->data(), (size_t)take * 5 * sizeof(int64_t));
    return MG_OK;
}

int mg_dist_exchange_plan(const mg_dist_config *cfg, int32_t level, int32_t *deep, int32_t *cut, int64_t own[2],
                          int64_t first[2], int64_t last[2], int32_t halo[4])
{
    if (!cfg) FAIL(MG_ERR_INVALID, "null configuration");
    // the checks of mg_dist_create that the plan depends on
    if (cfg->world < 1 || cfg->world > MG_MAX_GPUS || cfg->rank < 0 || cfg->rank >= cfg->world)
        FAIL(MG_ERR_INVALID, "rank %d / world %d out of range", cfg->rank, cfg->world);
    if (cfg->base.depth < 1 || cfg->base.depth >= MG_MAX_LEVELS) FAIL(MG_ERR_INVALID, "depth %d out of range 1..%d", cfg->base.depth, MG_MAX_LEVELS - 1);
    for (int a = 0; a < 3; a++)
        if (cfg->base.n[a] < 3 || cfg->base.n[a] > 65535) FAIL(MG_ERR_INVALID, "n[%d]=%lld out of range", a, (long long)cfg->base.n[a]);
    for (int l = 0; l <= cfg->base.depth; l++)
        if (cfg->base.nu[l] < 0 || cfg->base.nu[l] > 100000) FAIL(MG_ERR_INVALID, "nu[%d]=%d out of range", l, cfg->base.nu[l]);
    Plan P;
    TRY(P.init(cfg->base.n, cfg->base.depth, cfg->world, cfg->coarse_cut));
    const bool is_deep = (cfg->no_overlap & 2) == 0 && P.deep_fits(cfg->base.nu);
    if (deep) *deep = is_deep ? 1 : 0;
    if (cut) *cut = P.cut;
    if (level < 1 || level > P.cut) FAIL(MG_ERR_INVALID, "level %d outside the partitioned levels 1..%d", level, P.cut);
    const IRange o = P.own(level, false, cfg->rank);
    if (own) { own[0] = o.lo; own[1] = o.hi; }
    IRange f{0, 0}, l{0, 0};
    Halo h{0, 0, 0, 0};
    if (is_deep && level < P.cut) {
        IRange lasts[MG_MAX_LEVELS], firsts[MG_MAX_LEVELS];
        P.deep_ranges(cfg->rank, cfg->base.nu, lasts, firsts);
        f = firsts[level];
        l = lasts[level];
        h = deep_halo(P, cfg->rank, level, cfg->base.nu);
    }
    if (first) { first[0] = f.lo; first[1] = f.hi; }
    if (last) { last[0] = l.lo; last[1] = l.hi; }
    if (halo) { halo[0] = h.rlo; halo[1] = h.slo; halo[2] = h.shi; halo[3] = h.rhi; }
    return MG_OK;
}

int mg_dist_set_rho_planes(mg_dist *d, int64_t lo, int64_t hi, const double *dense)
{
    TRY(check_rank(d, false));
    TRY(upload_planes(d, d->b, lo, hi, dense, true));
    d->have_rho = true;
    return MG_OK;
}

int mg_dist_set_phi_planes(mg_dist *d, int64_t lo, int64_t hi, const double *dense)
{
    TRY(check_rank(d, false));
    if (d->phi.pending) FAIL(MG_ERR_STATE, "a halo exchange of phi is still in flight");
    TRY(upload_planes(d, d->phi, lo, hi, dense, false));
    d->have_phi = true;
    d->have_ef = false;
    d->correction_pending = false;
    return MG_OK;
}

int mg_dist_get_phi_planes(mg_dist *d, int64_t lo, int64_t hi, double *dense)
{
    TRY(check_rank(d, false));
    if (!dense || !d->have_phi) FAIL(MG_ERR_STATE, "no phi on the device yet");
    const IRange o = d->own(0, false);
    if (lo < o.lo || hi > o.hi || lo >= hi) FAIL(MG_ERR_INVALID, "planes [%lld,%lld) outside the owned range [%lld,%lld)", (long long)lo, (long long)hi, (long long)o.lo, (long long)o.hi);
    if (d->correction_pending) FAIL(MG_ERR_STATE, "a coarse-grid correction is pending: finish the solve call first");
    TRY(d->tr->finish(d->phi, d->stream));
    const View &v = d->phi.v;
    const int64_t cap = d->phi_alt.v.d[0];
    for (int64_t a = lo; a < hi; a += cap) {
        const int64_t z = a + cap < hi ? a + cap : hi;
        const size_t bytes = (size_t)(z - a) * v.d[1] * v.d[2] * sizeof(double);
        View part = v;
        part.p = d->phi.plane(a);
        part.i0 = a;
        part.d[0] = z - a;
        double *dst = dense + (size_t)(a - lo) * v.d[1] * v.d[2];
        if (part.sj == part.d[2]) {
            HIP_TRY(hipMemcpyAsync(dst, part.p, bytes, hipMemcpyDeviceToHost, d->stream));
        } else {
            mgk::repitch_out(d->stream, d->phi_alt.v.p, part);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(dst, d->phi_alt.v.p, bytes, hipMemcpyDeviceToHost, d->stream));
        }
        HIP_TRY(hipStreamSynchronize(d->stream));
    }
    return MG_OK;
}

namespace {
// A rank that fails inside a collective call gives its communication up: its later calls fail at
// once, and (in-process transport) the ranks waiting for it are released.  Ranks of an RCCL group
// in OTHER processes are not reached by this -- their launcher ends them when this process exits
// with the error; ranks that are threads of this process are released by mg_dist_abort on each
// (mg_create with n_gpus does so).
int ended(mg_dist *d, int rc)
{
    if (rc && d->tr) d->tr->abort();
    return rc;
}
int vcycles_impl(mg_dist *d, unsigned count, double *L2);
int solve_impl(mg_dist *d, unsigned max_it, double tol, int *converged, unsigned *cycles, double *L2);
}  // namespace

int mg_dist_vcycles(mg_dist *d, unsigned count, double *L2)
{
    TRY(check_rank(d, true));
    return ended(d, vcycles_impl(d, count, L2));
}

int mg_dist_solve(mg_dist *d, unsigned max_it, double tol, int *converged, unsigned *cycles, double *L2)
{
    TRY(check_rank(d, true));
    return ended(d, solve_impl(d, max_it, tol, converged, cycles, L2));
}

namespace {

int vcycles_impl(mg_dist *d, unsigned count, double *L2)
{
    d->have_ef = false;
    HIP_TRY(hipEventRecord(d->ev_a, d->stream));
    for (unsigned it = 0; it < count; it++) {
        TRY(d->smooth_and_residual());
        TRY(d->coarse_correction());
    }
    TRY(d->flush_correction());
    HIP_TRY(hipEventRecord(d->ev_b, d->stream));
    double sum = 0;
    TRY(d->read_sum(&sum));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, d->ev_a, d->ev_b));
    d->last_ms = ms;
    if (L2) *L2 = count ? d->l2_of(sum) : 0.0;
    return MG_OK;
}

int solve_impl(mg_dist *d, unsigned max_it, double tol, int *converged, unsigned *cycles, double *L2)
{
    d->have_ef = false;
    bool conv = false;
    double l2 = 0;
    unsigned it = 0;
    HIP_TRY(hipEventRecord(d->ev_a, d->stream));
    for (it = 0; it < max_it; it++) {
        TRY(d->smooth_and_residual());
        double sum = 0;
        TRY(d->read_sum(&sum));
        l2 = d->l2_of(sum);  // the all-reduced sum: every rank takes the same decision
        if (l2 < tol) {
            conv = true;
            it++;
            break;
        }
        TRY(d->coarse_correction());
    }
    TRY(d->flush_correction());
    HIP_TRY(hipEventRecord(d->ev_b, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, d->ev_a, d->ev_b));
    d->last_ms = ms;
    if (converged) *converged = conv ? 1 : 0;
    if (cycles) *cycles = it;
    if (L2) *L2 = l2;
    return MG_OK;
}

}  // namespace

int mg_dist_compute_ef(mg_dist *d)
{
    TRY(check_rank(d, false));
    if (!d->have_phi) FAIL(MG_ERR_STATE, "no phi on the device yet");
    const IRange o = d->own(0, false);
    const int64_t *n = d->plan.n0;
    // the central differences read one halo plane per side; the one-sided formulas on the two outer
    // faces of the grid read planes 0..2 / n-3..n-1, which the outer ranks' stored planes must cover
    if ((o.lo == 0 && d->phi.v.i_end() < 3) || (o.hi == n[0] && d->phi.v.i0 > n[0] - 3))
        FAIL(MG_ERR_STATE, "an outer rank stores fewer than three planes: the field on the grid's faces needs three");
    if (!d->ef) {
        HIP_TRY(hipMalloc((void **)&d->ef, (size_t)(o.hi - o.lo) * n[1] * n[2] * 3 * sizeof(double)));
        d->blocks.push_back(d->ef);
    }
    TRY(d->flush_correction());
    // phi's halo planes must be those of the neighbours' current potential
    TRY(d->tr->start(d->phi, o, H_U, d->stream));
    TRY(d->tr->finish(d->phi, d->stream));
    mgk::compute_ef(d->stream, d->phi.v, d->ef, o.lo, n, d->dh, o);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(d->stream));
    d->have_ef = true;
    return MG_OK;
}

int mg_dist_get_ef_planes(mg_dist *d, int64_t lo, int64_t hi, double *dense)
{
    TRY(check_rank(d, false));
    if (!dense) FAIL(MG_ERR_INVALID, "null argument");
    if (!d->have_ef) FAIL(MG_ERR_STATE, "mg_dist_compute_ef has not run on the current phi");
    const IRange o = d->own(0, false);
    if (lo < o.lo || hi > o.hi || lo >= hi) FAIL(MG_ERR_INVALID, "planes [%lld,%lld) outside the owned range [%lld,%lld)", (long long)lo, (long long)hi, (long long)o.lo, (long long)o.hi);
    const size_t plane = (size_t)d->plan.n0[1] * d->plan.n0[2] * 3;
    HIP_TRY(hipMemcpyAsync(dense, d->ef + (size_t)(lo - o.lo) * plane, (size_t)(hi - lo) * plane * sizeof(double),
                           hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return MG_OK;
}

int mg_dist_abort(mg_dist *d)
{
    if (!d) FAIL(MG_ERR_INVALID, "null rank handle");
    if (d->tr) d->tr->abort();
    return MG_OK;
}

int mg_dist_sync(mg_dist *d)
{
    TRY(check_rank(d, false));
    HIP_TRY(hipStreamSynchronize(d->stream));
    return MG_OK;
}

int mg_dist_stats(mg_dist *d, double *last_ms, uint64_t *bytes_sent)
{
    if (!d) FAIL(MG_ERR_INVALID, "null rank handle");
    if (last_ms) *last_ms = d->last_ms;
    if (bytes_sent) *bytes_sent = d->tr ? d->tr->bytes_sent : 0;
    return MG_OK;
}

}  // extern "C"
