// fit.hip -- the two drop-in fit entry points (level 1 of include/cmfrec_hip.h).
//
// Host-side restatement of the driver logic of the reference's fit_collective_implicit_als
// (/root/reference/src/collective.c:9375-10207) and fit_collective_explicit_als (:7263-9370) for
// the supported option set: input validation, log transform, global mean (common.c:3423-3648), side-info column centering
// (common.c:4911-4997), start values; X := (X - mean)*alpha, COO -> CSR + CSC (stable, helpers.c:1375-1491)
// and the bias initialisation (common.c:4410-4909) run on the device (coo_device.hpp); the ALS loop
// order C -> D -> B -> A, which runs on the device-resident session (session.hip).  Every
// temporary is allocated and freed here; outputs are caller-allocated (cmfrec.h.in:240-241).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <cstdlib>
#include <functional>
#include <vector>
#include <dlfcn.h>
#include <set>
#include <string>
#include <hip/hip_runtime.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>          // types and prototypes only: the library is bound at run time, when a fit spans several devices
#define CMF_HAVE_RCCL 1
#else
#define CMF_HAVE_RCCL 0         // a ROCm install without the RCCL headers: the shards exchange by peer copies only
#endif

#include "../../include/cmfrec_hip.h"
#include "rng_host.hpp"
#include "exchange_plan.hpp"

namespace {

// ---- SIGINT between half-steps (helpers.c:1493-1501, collective.c:9520-9531) -----------------
volatile sig_atomic_t g_stop = 0;
void on_sigint(int) { g_stop = 1; }
struct SigGuard {
    void (*old)(int) = nullptr;
    bool armed = false;
    explicit SigGuard(bool arm) : armed(arm)
    {
        // the flag is cleared for every fit, armed or not: the loops test it unconditionally, and a Ctrl-C handled by an
        // earlier fit must not end the next one at iteration 0 (the reference resets should_stop_procedure in its cleanup)
        g_stop = 0;
        if (arm) old = signal(SIGINT, on_sigint);
    }
    ~SigGuard() { if (armed) signal(SIGINT, old); }
};

#ifdef CMFREC_HIP_FLOAT
const real_t EPS_T = FLT_EPSILON;
#else
const real_t EPS_T = DBL_EPSILON;
#endif

// CMFREC_HIP_TIMING=1: wall-clock of the host phases of a fit on stderr
struct PhaseTimer {
    bool on = getenv("CMFREC_HIP_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        if (!on) return;
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[cmfrec_hip timing] %-28s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};

int fail(bool verbose, const char *msg)
{
    if (verbose) fprintf(stderr, "%s\n", msg);
    return 2;
}

// Dense side information with missing values (NaN).  The reference centres the present entries of each column
// (center_by_cols, common.c:4938-4997) and then uses, for every row / column, only what is present -- the same
// arithmetic as its sparse-side-information route run on the centred values (collective.c:1566-1653: the C^T C block
// and the right-hand side over the present attributes, lambda scaled by their count; optimizeA Case 2 for C / D).  So
// such a matrix is handed to the sparse route as COO triplets of its centred present entries.
struct DenseNanSide {
    std::vector<int_t> row, col;
    std::vector<real_t> val;
    std::vector<int_t> na_col;                  // missing values per attribute
    int_t n_rows = 0;
    // What the reference's dense C / D update (optimizeA Cases 1-2 on the transposed matrix, common.c:2793-3116) does differently
    // from the sparse one, per attribute.  An attribute that misses fewer than 2 kc values is solved from the precomputed Gramian
    // minus the missing rows (factors_closed_form :759-790, ahead of the CG branch at :884): in closed form whatever use_cg says
    // (mask 1), under scale_lam with the n_rows x lam of a complete attribute (:3031-3032 / :2832).  When at least 75 % of the
    // attributes are complete (helpers.c:151-250) those share one factorisation (mask 1) and the ones that miss 2 kc values or
    // more are redone by CG from zero with kc steps (mask 2; Case 1's fix-up loop, :2953-2985).  Mask 0: the solver asked for.
    void rules(int_t kc, bool scale_lam, std::vector<unsigned char> &mask, std::vector<real_t> &mult) const
    {
        const int_t p = (int_t)na_col.size();
        int_t with_na = 0;
        bool any_few = false;
        for (int_t c = 0; c < p; c++) { with_na += (na_col[c] > 0); any_few = any_few || (na_col[c] > 0 && na_col[c] < 2 * kc); }
        const bool near = (p - with_na) >= (int_t)((real_t)0.75 * (real_t)p);
        mask.resize((size_t)p);
        for (int_t c = 0; c < p; c++) mask[c] = (na_col[c] < 2 * kc) ? 1 : (near ? 2 : 0);
        mult.clear();
        if (scale_lam && any_few) {
            mult.resize((size_t)p);
            for (int_t c = 0; c < p; c++)
                mult[c] = (na_col[c] < 2 * kc) ? (real_t)n_rows : (na_col[c] < n_rows ? (real_t)(n_rows - na_col[c]) : (real_t)1);
        }
    }
    bool convert(const real_t *M, int_t rows, int_t cols, real_t *means)
    {
        bool any = false;
        for (size_t e = 0; e < (size_t)rows * cols && !any; e++) any = std::isnan(M[e]);
        if (!any) return false;
        std::vector<double> sum((size_t)cols, 0.0);
        std::vector<size_t> cnt((size_t)cols, 0);
        for (int_t r = 0; r < rows; r++)
            for (int_t c = 0; c < cols; c++) {
                const real_t v = M[(size_t)r * cols + c];
                if (!std::isnan(v)) { sum[c] += v; cnt[c]++; }
            }
        for (int_t c = 0; means && c < cols; c++) means[c] = (real_t)(sum[c] / (double)cnt[c]);
        n_rows = rows; na_col.resize((size_t)cols);
        for (int_t c = 0; c < cols; c++) na_col[c] = rows - (int_t)cnt[c];
        for (int_t r = 0; r < rows; r++)
            for (int_t c = 0; c < cols; c++) {
                const real_t v = M[(size_t)r * cols + c];
                if (std::isnan(v)) continue;
                row.push_back(r); col.push_back(c); val.push_back(means ? v - means[c] : v);
            }
        return true;
    }
};

// NA_as_zero_U / NA_as_zero_I: a sparse side-information matrix whose absent entries are ZEROS is the dense matrix holding those
// zeros, on every row of X (rows of X beyond the last row of the triplets are zero rows: the column means divide by all of
// them).  The reference reaches the same numbers by other operations -- optimizeA Case 3 for C / D with the column means as a
// rank-one correction (collective.c:8354-8386, common.c:3116-3205), the full w C^T C block plus a constant -w C^T colmeans on
// every right-hand side (collective.c:1277-1457, :5790-5800, :5823-5836; block CG: :2292-2298) -- and agrees with the dense
// route on the zero-filled matrix to 1e-15, closed form and CG, both models (tests/test_oracle_vs_ref.py).  So that is what the
// fit runs: the triplets are scattered into a [rows of X, cols] matrix here and take the dense path, GEMMs and all.  Cost: rows x
// cols numbers of host and device memory instead of the triplets.  More rows of side information than X has: refused (the
// reference treats the rows beyond X differently from its dense branch, nothing pins them).
struct ZeroFilledSide {
    std::vector<real_t> dense;
    // Size limit of the zero-filled matrix: it exists three times (here, as the centred copy, on the device), so it is refused
    // -- a controlled "not implemented at this size", not a bad_alloc / hipErrorOutOfMemory late in the fit -- beyond
    // CMFREC_HIP_ZEROFILL_MAX_GB (default 8 GB per copy; realistic sparse side information of 1e6..1e7 rows x 1e3..1e4 columns
    // is 8e9..8e11 bytes and belongs on the sparse kernels with the -w C^T colmeans constant, which this route does not build).
    static double max_bytes()
    {
        const char *e = getenv("CMFREC_HIP_ZEROFILL_MAX_GB");
        const double gb = (e != nullptr && atof(e) > 0) ? atof(e) : 8.0;
        return gb * 1e9;
    }
    // 0 ok, 1 triplets missing / out of range, 2 more rows than X, 3 larger than the limit above.  Triplets that repeat a
    // position are SUMMED (the reference's COO -> CSR keeps both and its sums add both: the same numbers).
    int build(int_t rows_x, int_t rows_side, int_t cols, const int_t *r, const int_t *c, const real_t *v, size_t nnz)
    {
        if (rows_side > rows_x) return 2;
        if (!r || !c || !v || cols <= 0) return 1;
        if ((double)rows_x * (double)cols * (double)sizeof(real_t) > max_bytes()) return 3;
        dense.assign((size_t)rows_x * (size_t)cols, (real_t)0);
        for (size_t e = 0; e < nnz; e++) {
            if (r[e] < 0 || r[e] >= rows_side || c[e] < 0 || c[e] >= cols) return 1;
            dense[(size_t)r[e] * (size_t)cols + (size_t)c[e]] += v[e];
        }
        return 0;
    }
};
// ... with one exception the reference makes: a row with neither an entry of X nor an entry of the side information is not solved
// but set to zero, bias included (collective_closed_form_block, collective.c:1262-1271; _implicit: :1876-1884).  The list of
// those rows, for cmfrec_hip_session_set_zero_rows.
static std::vector<int_t> rows_without_data(int_t rows, const int_t *ix_x, size_t nnz, const int_t *ix_side, size_t nnz_side)
{
    std::vector<char> has((size_t)rows, 0);
    for (size_t e = 0; e < nnz; e++) if (ix_x[e] >= 0 && ix_x[e] < rows) has[(size_t)ix_x[e]] = 1;
    for (size_t e = 0; e < nnz_side; e++) if (ix_side[e] >= 0 && ix_side[e] < rows) has[(size_t)ix_side[e]] = 1;
    std::vector<int_t> out;
    for (int_t r = 0; r < rows; r++) if (!has[(size_t)r]) out.push_back(r);
    return out;
}
// precomputedCtUbias of the reference's epilogue (collective.c:9244-9252, :10115-10123): -w C^T colmeans
static void fill_CtUbias(real_t *out, const real_t *C, const real_t *colmeans, int_t p, int_t kc, real_t w)
{
    if (!out || !C || !colmeans) return;
    for (int_t f = 0; f < kc; f++) {
        double acc = 0;
        for (int_t j = 0; j < p; j++) acc += (double)C[(size_t)j * kc + f] * (double)colmeans[j];
        out[f] = (real_t)(-(double)w * acc);
    }
}

// column means of sparse side information are reported like the reference does (common.c:4976-4990); the fit itself runs on
// the values as given (the centred copy center_by_cols makes is not the one coo_to_csr_and_csc reads, collective.c:6541-6558)
void sparse_colmeans(const int_t *col, const real_t *val, size_t nnz, int_t cols, real_t *means)
{
    if (!means) return;
    std::vector<double> sum((size_t)cols, 0.0);
    std::vector<size_t> cnt((size_t)cols, 0);
    for (size_t e = 0; e < nnz; e++) { sum[col[e]] += val[e]; cnt[col[e]]++; }
    for (int_t c = 0; c < cols; c++) means[c] = (real_t)(sum[c] / (double)cnt[c]);
}

// chol_A / chol_B: that half-step is a closed-form solve whatever use_cg says (dense X without weights whose rows / columns are
// all or nearly all complete: optimizeA Case 1, common.c:2787-2993)
int run_loop(cmfrec_hip_session *s, const cmfrec_hip_model &mdl, int niter, bool finalize_chol, bool verbose, bool implicit_feats = false,
             bool chol_A = false, bool chol_B = false)
{
    for (int it = 0; it < niter; it++) {
        if (g_stop) return 3;
        int chol = (finalize_chol && mdl.use_cg && it == niter - 1) ? 1 : 0;
        int rc;
        if (mdl.p > 0) { if (verbose) { printf("Updating C..."); fflush(stdout); }
            if ((rc = cmfrec_hip_session_update(s, 'C', chol))) return rc; if (verbose) printf(" done\n"); }
        if (g_stop) return 3;
        if (mdl.q > 0) { if (verbose) { printf("Updating D..."); fflush(stdout); }
            if ((rc = cmfrec_hip_session_update(s, 'D', chol))) return rc; if (verbose) printf(" done\n"); }
        if (g_stop) return 3;
        if (implicit_feats) {                                             // collective.c:8448-8534
            if (verbose) { printf("Updating Bi..."); fflush(stdout); }
            if ((rc = cmfrec_hip_session_update(s, 'b', 1))) return rc;
            if (verbose) { printf(" done\nUpdating Ai..."); fflush(stdout); }
            if ((rc = cmfrec_hip_session_update(s, 'a', 1))) return rc;
            if (verbose) printf(" done\n");
            if (g_stop) return 3;
        }
        if (verbose) { printf("Updating B..."); fflush(stdout); }
        if ((rc = cmfrec_hip_session_update(s, 'B', (chol || chol_B) ? 1 : 0))) return rc;
        if ((rc = cmfrec_hip_session_after_gather(s, 'B'))) return rc;
        if (verbose) { cmfrec_hip_session_sync(s); printf(" done\n"); }
        if (g_stop) return 3;
        if (verbose) { printf("Updating A..."); fflush(stdout); }
        if ((rc = cmfrec_hip_session_update(s, 'A', (chol || chol_A) ? 1 : 0))) return rc;
        if ((rc = cmfrec_hip_session_after_gather(s, 'A'))) return rc;
        if (verbose) { cmfrec_hip_session_sync(s); printf(" done\n\tCompleted ALS iteration %2d\n\n", it + 1); fflush(stdout); }
    }
    return 0;
}

// Dense X (Xfull [m, n] row-major, NaN = missing; weight, if given, dense too) -> the COO of its present entries in row-major
// order, which is the order every running mean of the reference's dense branches takes them in (common.c:3463-3490,
// :4152-4180), plus what optimizeA's case selection needs (helpers.c:151-250): a half-step is "full" when no row (column) has
// a missing entry and "near dense" when at least 75 % of them have none.
struct DenseX {
    std::vector<int_t> row, col;
    std::vector<real_t> val, w;
    bool full = false, near_row = false, near_col = false;
    std::vector<int_t> na_row, na_col;          // missing entries per row / column
    void convert(const real_t *Xfull, const real_t *weight, int_t m, int_t n)
    {
        na_row.assign((size_t)m, 0); na_col.assign((size_t)n, 0);
        size_t present = 0;
        for (int_t r = 0; r < m; r++)
            for (int_t c = 0; c < n; c++) {
                const bool na = std::isnan(Xfull[(size_t)r * n + c]);
                na_row[r] += na; na_col[c] += na; present += !na;
            }
        row.reserve(present); col.reserve(present); val.reserve(present);
        if (weight) w.reserve(present);
        for (int_t r = 0; r < m; r++)
            for (int_t c = 0; c < n; c++) {
                const real_t x = Xfull[(size_t)r * n + c];
                if (std::isnan(x)) continue;
                row.push_back(r); col.push_back(c); val.push_back(x);
                if (weight) w.push_back(weight[(size_t)r * n + c]);
            }
        full = (present == (size_t)m * (size_t)n);
        if (!full) {
            int_t with_na = 0;
            for (int_t r = 0; r < m; r++) with_na += (na_row[r] > 0);
            near_row = (m - with_na) >= (int)(0.75 * (double)m);
            with_na = 0;
            for (int_t c = 0; c < n; c++) with_na += (na_col[c] > 0);
            near_col = (n - with_na) >= (int_t)((real_t)0.75 * (real_t)n);
        }
    }
};

// ---- several GPUs of one node behind the unchanged C signature ----------------------------------
// CMFREC_HIP_DEVICES="0,1,2,3" (HIP device ordinals, comma separated; an ordinal may repeat, which shards one device --
// that is how the path is tested on a single-GPU box).  One entry: that device instead of the current one.  Several:
// users are cut into equal contiguous blocks and items into nnz-balanced ones, every device owns one block of each, holds
// full replicas of A and B, updates its own rows, and the updated rows travel to the peers over xGMI
// (hipMemcpyPeerAsync, ordered by events; no host synchronisation inside the loop) -- the all-gather of distributed.py
// without torch.  Rows are independent given the opposing matrix and every device computes B^T B from identical
// replicas, so the factors are bit-for-bit those of the single-device fit as long as every row takes the same kernel path on
// the shard as on the whole matrix.  That holds for rows of at most 1024 entries (512 in double precision); for split rows the
// choice between the streaming and the Gramian path, and the slice length, follow the SHARD's statistics (device.hpp,
// prefer_gram / slice_len), so their sums may be taken in another order: equal to rounding, not to the bit
// (tests/test_gpu_multidevice.py has both cases).
std::vector<int> devices_from_env()
{
    std::vector<int> out;
    const char *e = getenv("CMFREC_HIP_DEVICES");
    if (e == nullptr) return out;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return out;
    const char *q = e;
    while (*q) {
        char *end = nullptr;
        const long v = strtol(q, &end, 10);
        if (end == q) break;
        if (v >= 0 && v < count) out.push_back((int)v);
        q = end;
        while (*q == ',' || *q == ' ') q++;
    }
    return out;
}

// RCCL for the exchange between the devices of one fit ("ncclAllGather ... or direct placement").  Bound with
// dlopen when the first multi-device fit asks for it -- a single-device caller never maps the library -- through the prototypes
// of <rccl/rccl.h> (decltype: nothing is declared by hand).
#if !CMF_HAVE_RCCL
// no RCCL headers on this install: enough of the interface for the code below to compile; load() never succeeds, so none of these
// is ever called and every multi-device fit exchanges by peer copies
typedef void *ncclComm_t;
enum ncclResult_t { ncclSuccess = 0, ncclSystemError = 2 };
enum ncclDataType_t { ncclFloat = 7, ncclDouble = 8 };
enum ncclRedOp_t { ncclSum = 0 };
ncclResult_t ncclCommInitAll(ncclComm_t *, int, const int *);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
ncclResult_t ncclSend(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclRecv(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclAllReduce(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t);
const char *ncclGetErrorString(ncclResult_t);
#endif
struct RcclApi {
    void *handle = nullptr;
    bool failed = false;        // a previous load found the library unusable: do not dlopen it again for every fit
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load()
    {
        if (handle) return true;
        if (failed || !CMF_HAVE_RCCL) return false;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) { failed = true; return false; }
#define CMF_SYM(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(handle, #sym)); if (!field) { dlclose(handle); handle = nullptr; failed = true; return false; }
        CMF_SYM(CommInitAll, ncclCommInitAll) CMF_SYM(CommDestroy, ncclCommDestroy) CMF_SYM(GroupStart, ncclGroupStart)
        CMF_SYM(GroupEnd, ncclGroupEnd) CMF_SYM(Send, ncclSend) CMF_SYM(Recv, ncclRecv) CMF_SYM(AllReduce, ncclAllReduce)
        CMF_SYM(GetErrorString, ncclGetErrorString)
#undef CMF_SYM
        return true;
    }
};
RcclApi &rccl() { static RcclApi api; return api; }
#ifdef CMFREC_HIP_FLOAT
const ncclDataType_t NCCL_REAL = ncclFloat;
#else
const ncclDataType_t NCCL_REAL = ncclDouble;
#endif

// acc[i] += x[i]: the partial sums of the C / D update when the shards exchange by copies (fixed order: shard 0, 1, 2 ...)
__global__ void add_into_kernel(real_t *__restrict__ acc, const real_t *__restrict__ x, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] += x[i];
}

// The shards of one fit and what travels between them.  Two transports:
//   RCCL   (distinct devices; the default there): the updated row blocks go by DIRECT PLACEMENT -- one ncclGroup of
//          ncclSend / ncclRecv per half-step, every device sends its block to each peer and receives the peers' blocks straight
//          into its replica, all pairs at once (on an xGMI node every pair of GPUs has a link of its own: one hop on D - 1 links
//          instead of a ring's D - 1 steps on one); the partial sums of the C / D update by ncclAllReduce.  Every call is
//          enqueued on the owning session's stream: kernels -> exchange -> next kernels are ordered by the stream, the host
//          thread never waits inside the loop.
//   copies (an ordinal repeats, i.e. several shards on one device -- how the path runs on a single-GPU box -- or
//          CMFREC_HIP_EXCHANGE=copy): hipMemcpyPeerAsync ordered by events; the C / D partial sums are added on the first
//          shard's device in shard order and copied back.
struct MultiDev {
    std::vector<cmfrec_hip_session *> sess;
    std::vector<int> dev;
    std::vector<int> rb, cb;               // block boundaries of users / items (D + 1 each)
    std::vector<hipEvent_t> ev;            // per device: "my block has reached every peer"
    std::vector<ncclComm_t> comm;          // RCCL transport: one communicator per shard (empty: copies)
    real_t *stage = nullptr, *stage2 = nullptr;   // copies transport: partial sums on the first shard's device
    size_t stage_n = 0;
    std::string error;
    ~MultiDev()
    {
        for (auto e : ev) if (e) (void)hipEventDestroy(e);
        if (!comm.empty()) for (auto c : comm) if (c) (void)rccl().CommDestroy(c);
        if (stage) { (void)hipSetDevice(dev.empty() ? 0 : dev[0]); (void)hipFree(stage); (void)hipFree(stage2); }
        for (auto s : sess) if (s) cmfrec_hip_session_destroy(s);
    }
    bool use_rccl() const { return !comm.empty(); }
    // CMFREC_HIP_EXCHANGE = rccl | copy (default: rccl where the ordinals are distinct and librccl can be bound)
    int init_transport(bool verbose)
    {
        const char *e = getenv("CMFREC_HIP_EXCHANGE");
        const bool want_copy = e != nullptr && strcmp(e, "copy") == 0, want_rccl = e != nullptr && strcmp(e, "rccl") == 0;
        const bool distinct = std::set<int>(dev.begin(), dev.end()).size() == dev.size();
        if (want_copy || !distinct) {
            if (want_rccl) { error = "cmfrec_hip: CMFREC_HIP_EXCHANGE=rccl needs distinct device ordinals in CMFREC_HIP_DEVICES"; return 2; }
            return 0;
        }
        if (!rccl().load()) {
            if (want_rccl) { error = "cmfrec_hip: librccl could not be loaded"; return 4; }
            if (verbose) printf("cmfrec_hip: librccl not found, the shards exchange by peer copies\n");
            return 0;
        }
        comm.assign(dev.size(), nullptr);
        const ncclResult_t r = rccl().CommInitAll(comm.data(), (int)dev.size(), dev.data());
        if (r != ncclSuccess) {
            comm.clear();
            error = std::string("cmfrec_hip: ncclCommInitAll failed: ") + rccl().GetErrorString(r);
            if (want_rccl) return 4;
            if (verbose) printf("%s; the shards exchange by peer copies\n", error.c_str());
            error.clear();
        }
        return 0;
    }
    // block d of matrix `which` (updated on device d) reaches every other replica; every stream then holds all blocks
    int exchange(int which)
    {
        const int D = (int)sess.size();
        const std::vector<int> &bb = (which == 'A') ? rb : cb;
        if (use_rccl()) {
            if (D == 1) return 0;
            std::vector<real_t *> base(D); std::vector<size_t> ld(D); std::vector<hipStream_t> st(D);
            for (int d = 0; d < D; d++) {
                size_t rows = 0;
                base[d] = (real_t *)cmfrec_hip_session_device_ptr(sess[d], which, &rows, &ld[d]);
                st[d] = (hipStream_t)cmfrec_hip_session_stream(sess[d]);
            }
            ncclResult_t r = rccl().GroupStart();
            // the schedule is a pure function of the block boundaries (exchange_plan.hpp; tests/test_exchange_plan.py)
            for (const cmfhip::ExchangeOp &op : cmfhip::direct_placement_plan(bb)) {
                if (r != ncclSuccess) break;
                const int d = op.dev;
                real_t *ptr = base[d] + (size_t)op.first_row * ld[d];
                const size_t cnt = (size_t)op.rows * ld[d];
                r = op.send ? rccl().Send(ptr, cnt, NCCL_REAL, op.peer, comm[d], st[d]) : rccl().Recv(ptr, cnt, NCCL_REAL, op.peer, comm[d], st[d]);
            }
            const ncclResult_t r2 = rccl().GroupEnd();
            if (r != ncclSuccess || r2 != ncclSuccess) {
                error = std::string("cmfrec_hip: RCCL exchange failed: ") + rccl().GetErrorString(r != ncclSuccess ? r : r2);
                return 4;
            }
            return 0;
        }
        for (int d = 0; d < D; d++) {
            size_t rows = 0, ld = 0;
            real_t *src = (real_t *)cmfrec_hip_session_device_ptr(sess[d], which, &rows, &ld);
            hipStream_t st = (hipStream_t)cmfrec_hip_session_stream(sess[d]);
            if (hipSetDevice(dev[d]) != hipSuccess) return 4;
            const size_t off = (size_t)bb[d] * ld, bytes = (size_t)(bb[d + 1] - bb[d]) * ld * sizeof(real_t);
            for (int e = 0; e < D && bytes; e++) {
                if (e == d) continue;
                size_t r2 = 0, l2 = 0;
                real_t *dst = (real_t *)cmfrec_hip_session_device_ptr(sess[e], which, &r2, &l2);
                if (hipMemcpyPeerAsync(dst + off, dev[e], src + off, dev[d], bytes, st) != hipSuccess) return 4;
            }
            if (hipEventRecord(ev[d], st) != hipSuccess) return 4;
        }
        for (int e = 0; e < D; e++) {
            hipStream_t st = (hipStream_t)cmfrec_hip_session_stream(sess[e]);
            if (hipSetDevice(dev[e]) != hipSuccess) return 4;
            for (int d = 0; d < D; d++)
                if (d != e && hipStreamWaitEvent(st, ev[d], 0) != hipSuccess) return 4;
        }
        return 0;
    }
    // the shards' partial sums [F_loc^T F_loc | U_loc^T F_loc] of a C / D update (cmfrec_hip_session_sideinfo_partial) become
    // their total on every shard -- the same numbers everywhere, so that every replica solves the same small system
    int allreduce_partials(size_t count)
    {
        const int D = (int)sess.size();
        if (use_rccl()) {
            ncclResult_t r = rccl().GroupStart();
            for (int d = 0; d < D && r == ncclSuccess; d++) {
                size_t rows = 0, ld = 0;
                real_t *part = (real_t *)cmfrec_hip_session_device_ptr(sess[d], 'P', &rows, &ld);
                r = rccl().AllReduce(part, part, count, NCCL_REAL, ncclSum, comm[d], (hipStream_t)cmfrec_hip_session_stream(sess[d]));
            }
            const ncclResult_t r2 = rccl().GroupEnd();
            if (r != ncclSuccess || r2 != ncclSuccess) {
                error = std::string("cmfrec_hip: RCCL all-reduce failed: ") + rccl().GetErrorString(r != ncclSuccess ? r : r2);
                return 4;
            }
            return 0;
        }
        if (D == 1) return 0;
        // copies: everything through the first shard's stream, between two host synchronisations (the test transport)
        for (int d = 0; d < D; d++) { const int rc = cmfrec_hip_session_sync(sess[d]); if (rc) return rc; }
        if (hipSetDevice(dev[0]) != hipSuccess) return 4;
        if (stage_n < count) {
            if (stage) { (void)hipFree(stage); (void)hipFree(stage2); stage = stage2 = nullptr; }
            if (hipMalloc((void **)&stage, count * sizeof(real_t)) != hipSuccess || hipMalloc((void **)&stage2, count * sizeof(real_t)) != hipSuccess) return 1;
            stage_n = count;
        }
        hipStream_t st0 = (hipStream_t)cmfrec_hip_session_stream(sess[0]);
        for (int d = 0; d < D; d++) {
            size_t rows = 0, ld = 0;
            const real_t *part = (const real_t *)cmfrec_hip_session_device_ptr(sess[d], 'P', &rows, &ld);
            if (hipMemcpyPeerAsync(d == 0 ? stage : stage2, dev[0], part, dev[d], count * sizeof(real_t), st0) != hipSuccess) return 4;
            if (d > 0) hipLaunchKernelGGL(add_into_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st0, stage, stage2, count);
        }
        for (int d = 0; d < D; d++) {
            size_t rows = 0, ld = 0;
            real_t *part = (real_t *)cmfrec_hip_session_device_ptr(sess[d], 'P', &rows, &ld);
            if (hipMemcpyPeerAsync(part, dev[d], stage, dev[0], count * sizeof(real_t), st0) != hipSuccess) return 4;
        }
        if (hipStreamSynchronize(st0) != hipSuccess) return 4;
        return 0;
    }
};

// One session per entry of `devs`, each owning a block of users and a block of items and holding the block's entries of X
// ((x - subtract) * alpha): users in equal contiguous blocks, items in nnz-balanced contiguous ones.
// `configure` is called on every new session once its shards are built.
int build_shards(MultiDev &md, const std::vector<int> &devs, const cmfrec_hip_model &mdl, int_t m, int_t n, const int_t *ixA,
                 const int_t *ixB, const real_t *X, size_t nnz, real_t subtract, real_t alpha,
                 const std::function<int(cmfrec_hip_session *, int)> &configure)
{
    const int D = (int)devs.size();
    md.dev = devs;
    {
        const int rc = md.init_transport(false);
        if (rc) { fprintf(stderr, "%s\n", md.error.c_str()); return rc; }
    }
    md.rb.assign(D + 1, 0); md.cb.assign(D + 1, 0);
    const int step = (m + D - 1) / D;
    for (int d = 0; d <= D; d++) md.rb[d] = std::min(d * step, (int)m);
    {
        std::vector<size_t> cnt((size_t)n + 1, 0);
        for (size_t e = 0; e < nnz; e++) cnt[(size_t)ixB[e] + 1]++;
        for (int j = 0; j < n; j++) cnt[j + 1] += cnt[j];
        int j = 0;
        for (int d = 1; d < D; d++) {
            const double target = (double)nnz * d / D;
            while (j < n && (double)cnt[j] < target) j++;
            md.cb[d] = std::max(md.cb[d - 1], j);
        }
        md.cb[D] = n;
    }
    for (int d = 0; d < D; d++)
        for (int e = 0; e < D; e++)
            if (devs[d] != devs[e] && hipSetDevice(devs[d]) == hipSuccess) {
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devs[d], devs[e]) == hipSuccess && can) {
                    const hipError_t pe = hipDeviceEnablePeerAccess(devs[e], 0);
                    if (pe != hipSuccess) (void)hipGetLastError();            // already enabled
                }
            }
    md.sess.assign(D, nullptr); md.ev.assign(D, nullptr);
    std::vector<int_t> key, oth; std::vector<real_t> val;
    for (int d = 0; d < D; d++) {
        cmfrec_hip_model md_d = mdl;
        md_d.row_begin = md.rb[d]; md_d.row_end = md.rb[d + 1]; md_d.col_begin = md.cb[d]; md_d.col_end = md.cb[d + 1];
        md.sess[d] = cmfrec_hip_session_create(&md_d, devs[d]);
        if (!md.sess[d]) { const int ec = cmfrec_hip_last_error_code(); return ec ? ec : 1; }
        if (hipEventCreateWithFlags(&md.ev[d], hipEventDisableTiming) != hipSuccess) return 4;
        // the block's entries in COO order (the order the stable device sort keeps inside a row), staged through device buffers
        for (int side = 0; side < 2; side++) {
            const int lo = side == 0 ? md.rb[d] : md.cb[d], hi = side == 0 ? md.rb[d + 1] : md.cb[d + 1];
            const int_t *kk = side == 0 ? ixA : ixB, *oo = side == 0 ? ixB : ixA;
            key.clear(); oth.clear(); val.clear();
            for (size_t e = 0; e < nnz; e++)
                if (kk[e] >= lo && kk[e] < hi) { key.push_back(kk[e] - lo); oth.push_back(oo[e]); val.push_back(X[e]); }
            int_t *dk = nullptr, *d_o = nullptr; real_t *dv = nullptr;
            const size_t cntE = key.size();
            if (hipSetDevice(devs[d]) != hipSuccess) return 4;
            if (hipMalloc((void **)&dk, std::max<size_t>(cntE, 1) * sizeof(int_t)) != hipSuccess) return 1;
            if (hipMalloc((void **)&d_o, std::max<size_t>(cntE, 1) * sizeof(int_t)) != hipSuccess) { (void)hipFree(dk); return 1; }
            if (hipMalloc((void **)&dv, std::max<size_t>(cntE, 1) * sizeof(real_t)) != hipSuccess) { (void)hipFree(dk); (void)hipFree(d_o); return 1; }
            (void)hipMemcpy(dk, key.data(), cntE * sizeof(int_t), hipMemcpyHostToDevice);
            (void)hipMemcpy(d_o, oth.data(), cntE * sizeof(int_t), hipMemcpyHostToDevice);
            (void)hipMemcpy(dv, val.data(), cntE * sizeof(real_t), hipMemcpyHostToDevice);
            const int rc = cmfrec_hip_session_set_X_coo_device(md.sess[d], side == 0 ? 'r' : 'c', dk, d_o, dv, cntE, subtract, alpha);
            (void)hipFree(dk); (void)hipFree(d_o); (void)hipFree(dv);
            if (rc) return rc;
        }
        const int rc = configure(md.sess[d], d);
        if (rc) return rc;
    }
    return 0;
}

// C ('C') or D ('D') update of the collective model over the shards (optimizeA Case 1, common.c:2793-2991, with U / I cut into the
// same row blocks as A / B): every shard adds up  F_loc^T F_loc  and  U_loc^T F_loc  over ITS rows, the sums are made global
// (ncclAllReduce, or added in shard order by the copies transport), and every shard solves the same small system -- C / D stay
// identical replicas without ever being exchanged.
int multi_sideinfo_step(MultiDev &md, const cmfrec_hip_model &mdl, int which)
{
    const int D = (int)md.sess.size();
    const size_t pp = (size_t)(which == 'C' ? mdl.p : mdl.q), kc = (size_t)((which == 'C' ? mdl.k_user : mdl.k_item) + mdl.k);
    for (int d = 0; d < D; d++) { const int rc = cmfrec_hip_session_sideinfo_partial(md.sess[d], which); if (rc) return rc; }
    const int rc = md.allreduce_partials(kc * kc + pp * kc);
    if (rc) return rc;
    for (int d = 0; d < D; d++) { const int rc2 = cmfrec_hip_session_sideinfo_finish(md.sess[d], which); if (rc2) return rc2; }
    return 0;
}

// The ALS loop over the shards, both models: C, D (dense side information), then B, then A (collective.c:8334-8898, :9855-10022) --
// every device updates its block, the updated rows (a bias column rides in them, collective.c:8538-8543) reach the peers
// (MultiDev::exchange), and every replica splits the opposing bias off again (cmfrec_hip_session_after_gather).
int multi_loop(MultiDev &md, const cmfrec_hip_model &mdl, int niter, bool finalize_chol, bool verbose)
{
    const int D = (int)md.sess.size();
    if (verbose) {
        printf("Starting ALS optimization routine (%d device shards, exchange by %s)\n\n", D, md.use_rccl() ? "RCCL" : "peer copies");
        fflush(stdout);
    }
    for (int it = 0; it < niter; it++) {
        if (g_stop) return 3;
        const int chol = (finalize_chol && mdl.use_cg && it == niter - 1) ? 1 : 0;
        if (mdl.p > 0) { const int rc = multi_sideinfo_step(md, mdl, 'C'); if (rc) return rc; }
        if (mdl.q > 0) { const int rc = multi_sideinfo_step(md, mdl, 'D'); if (rc) return rc; }
        for (int pass = 0; pass < 2; pass++) {
            const int which = pass == 0 ? 'B' : 'A';
            for (int d = 0; d < D; d++) { const int rc = cmfrec_hip_session_update(md.sess[d], which, chol); if (rc) return rc; }
            int rc = md.exchange(which);
            if (rc == 4 && !md.error.empty()) fprintf(stderr, "%s\n", md.error.c_str());
            for (int d = 0; d < D && !rc; d++) rc = cmfrec_hip_session_after_gather(md.sess[d], which);
            if (rc) return rc;
            if (g_stop) return 3;
        }
        if (verbose) { for (int d = 0; d < D; d++) cmfrec_hip_session_sync(md.sess[d]); printf("\tCompleted ALS iteration %2d\n\n", it + 1); fflush(stdout); }
    }
    for (int d = 0; d < D; d++) { const int rc = cmfrec_hip_session_sync(md.sess[d]); if (rc) return rc; }
    return 0;
}

// the rows of the (centred) dense side information that belong to shard d: U rows [rb[d], min(rb[d+1], m_u)), I likewise
int set_sideinfo_shard(const MultiDev &md, cmfrec_hip_session *sd, int d, const real_t *Uc, int_t m_u, int_t p, const real_t *Ic, int_t n_i, int_t q)
{
    const real_t *Ul = (Uc && p > 0 && md.rb[d] < m_u) ? Uc + (size_t)md.rb[d] * p : nullptr;
    const real_t *Il = (Ic && q > 0 && md.cb[d] < n_i) ? Ic + (size_t)md.cb[d] * q : nullptr;
    if ((Uc && p > 0) || (Ic && q > 0)) return cmfrec_hip_session_set_sideinfo_local(sd, Ul, Il);
    return 0;
}

// several entries in CMFREC_HIP_DEVICES, or one entry together with CMFREC_HIP_SHARDED=1 (the sharded driver on a single
// shard: what a one-GPU box can run of the RCCL transport -- communicator, all-reduce -- tests/test_gpu_multidevice.py)
bool sharded_fit_wanted(const std::vector<int> &devs)
{
    if (devs.size() > 1) return true;
    const char *e = getenv("CMFREC_HIP_SHARDED");
    return devs.size() == 1 && e != nullptr && e[0] == '1';
}

}  // namespace

extern "C" {

/* the exchange schedule of MultiDev::exchange as data (include/cmfrec_hip.h); host-only */
int cmfrec_hip_exchange_plan(int D, const int *bb, int *out, int cap)
{
    if (D < 1 || bb == nullptr) return -1;
    const std::vector<cmfhip::ExchangeOp> ops = cmfhip::direct_placement_plan(std::vector<int>(bb, bb + D + 1));
    for (size_t i = 0; i < ops.size() && (int)i < cap && out != nullptr; i++) {
        out[5 * i] = ops[i].dev; out[5 * i + 1] = ops[i].peer; out[5 * i + 2] = ops[i].send;
        out[5 * i + 3] = ops[i].first_row; out[5 * i + 4] = ops[i].rows;
    }
    return (int)ops.size();
}

/* Start values exactly as the reference's random_parallel (helpers.c:927-1043) draws them. */
int cmfrec_hip_random_parallel(real_t *A, size_t sizeA, real_t *B, size_t sizeB, int_t seed, bool normal)
{
    cmfrng::random_parallel<real_t>(A, sizeA, B, sizeB, seed, normal);
    return CMF_ZIGGURAT_TABLES_EXACT;
}

int_t fit_collective_implicit_als(
    real_t *A, real_t *B, real_t *C, real_t *D, bool reset_values, int_t seed,
    real_t *U_colmeans, real_t *I_colmeans, int_t m, int_t n, int_t k,
    int_t ixA[], int_t ixB[], real_t *X, size_t nnz,
    real_t lam, real_t *lam_unique, real_t l1_lam, real_t *l1_lam_unique,
    real_t *U, int_t m_u, int_t p, real_t *II, int_t n_i, int_t q,
    int_t U_row[], int_t U_col[], real_t *U_sp, size_t nnz_U,
    int_t I_row[], int_t I_col[], real_t *I_sp, size_t nnz_I,
    bool NA_as_zero_U, bool NA_as_zero_I, int_t k_main, int_t k_user, int_t k_item,
    real_t w_main, real_t w_user, real_t w_item, real_t *w_main_multiplier,
    real_t alpha, bool adjust_weight, bool apply_log_transf, int_t niter, int nthreads,
    bool verbose, bool handle_interrupt, bool use_cg, int_t max_cg_steps, bool precondition_cg,
    bool finalize_chol, bool nonneg, int_t max_cd_steps, bool nonneg_C, bool nonneg_D,
    bool precompute_for_predictions, real_t *precomputedBtB, real_t *precomputedBeTBe,
    real_t *precomputedBeTBeChol, real_t *precomputedCtUbias)
{

    (void)nthreads;
    // collective.c:9406-9435
    if (k_user && U == nullptr && nnz_U == 0) return fail(verbose, "Cannot pass 'k_user' without U data.");
    if (k_item && II == nullptr && nnz_I == 0) return fail(verbose, "Cannot pass 'k_item' without I data.");
    if (k_main && nnz == 0) return fail(verbose, "Cannot pass 'k_main' without X data.");
    if (nnz == 0) return fail(verbose, "cmfrec_hip: the implicit model needs at least one entry of X.");
    // sparse side information whose absent entries are zeros -> the dense route on the zero-filled matrix (ZeroFilledSide)
    ZeroFilledSide zfU, zfI;
    const bool naz_U = NA_as_zero_U && U == nullptr && nnz_U > 0, naz_I = NA_as_zero_I && II == nullptr && nnz_I > 0;
    std::vector<int_t> zero_rows_A, zero_rows_B;
    if (naz_U) {
        zero_rows_A = rows_without_data(m, ixA, nnz, U_row, nnz_U);
        const int e = zfU.build(m, m_u, p, U_row, U_col, U_sp, nnz_U);
        if (e) return fail(verbose, e == 2 ? "cmfrec_hip: NA_as_zero_U with more rows of U than X is not implemented."
                                    : e == 3 ? "cmfrec_hip: NA_as_zero_U runs on the zero-filled dense matrix (rows of X x p), which is larger than "
                                               "CMFREC_HIP_ZEROFILL_MAX_GB (default 8) here: not implemented at this size."
                                             : "cmfrec_hip: U index out of range.");
        U = zfU.dense.data(); m_u = m; nnz_U = 0; U_row = U_col = nullptr; U_sp = nullptr;
    }
    if (naz_I) {
        zero_rows_B = rows_without_data(n, ixB, nnz, I_row, nnz_I);
        const int e = zfI.build(n, n_i, q, I_row, I_col, I_sp, nnz_I);
        if (e) return fail(verbose, e == 2 ? "cmfrec_hip: NA_as_zero_I with more rows of I than X has columns is not implemented."
                                    : e == 3 ? "cmfrec_hip: NA_as_zero_I runs on the zero-filled dense matrix (columns of X x q), which is larger than "
                                               "CMFREC_HIP_ZEROFILL_MAX_GB (default 8) here: not implemented at this size."
                                             : "cmfrec_hip: I index out of range.");
        II = zfI.dense.data(); n_i = n; nnz_I = 0; I_row = I_col = nullptr; I_sp = nullptr;
    }
    // dense side information with NaN -> the sparse route on its centred present entries
    DenseNanSide nanU, nanI;
    const bool hadU = (U != nullptr);
    bool nan_side = false;
    if (U && m_u > 0 && p > 0 && nanU.convert(U, m_u, p, U_colmeans)) {
        U = nullptr; U_row = nanU.row.data(); U_col = nanU.col.data(); U_sp = nanU.val.data(); nnz_U = nanU.val.size(); nan_side = true;
    } else if (U == nullptr && nnz_U > 0 && U_row && U_col && U_sp) {
        bool ok = true;
        for (size_t e = 0; e < nnz_U && ok; e++) ok = (U_col[e] >= 0 && U_col[e] < p);
        if (ok) sparse_colmeans(U_col, U_sp, nnz_U, p, U_colmeans);
    }
    if (II && n_i > 0 && q > 0 && nanI.convert(II, n_i, q, I_colmeans)) {
        II = nullptr; I_row = nanI.row.data(); I_col = nanI.col.data(); I_sp = nanI.val.data(); nnz_I = nanI.val.size(); nan_side = true;
    } else if (II == nullptr && nnz_I > 0 && I_row && I_col && I_sp) {
        bool ok = true;
        for (size_t e = 0; e < nnz_I && ok; e++) ok = (I_col[e] >= 0 && I_col[e] < q);
        if (ok) sparse_colmeans(I_col, I_sp, nnz_I, q, I_colmeans);
    }
    if (nan_side && nnz_U == 0 && nanU.row.empty() && hadU && U == nullptr)
        return fail(verbose, "cmfrec_hip: U has no present entries.");
    // side information: dense or sparse COO (missing = absent).  Sparse: rows within X.
    const bool spU = (U == nullptr && nnz_U > 0), spI = (II == nullptr && nnz_I > 0);
    if ((spU && (m_u > m || !U_row || !U_col || !U_sp)) || (spI && (n_i > n || !I_row || !I_col || !I_sp)))
        return fail(verbose, "cmfrec_hip: sparse side information must be COO triplets with rows inside X.");
    if (U == nullptr && !spU) { m_u = 0; p = 0; }
    if (II == nullptr && !spI) { n_i = 0; q = 0; }
    for (size_t e = 0; spU && e < nnz_U; e++)
        if (U_row[e] < 0 || U_row[e] >= m_u || U_col[e] < 0 || U_col[e] >= p) return fail(verbose, "cmfrec_hip: U index out of range.");
    for (size_t e = 0; spI && e < nnz_I; e++)
        if (I_row[e] < 0 || I_row[e] >= n_i || I_col[e] < 0 || I_col[e] >= q) return fail(verbose, "cmfrec_hip: I index out of range.");
    if ((l1_lam != 0 || l1_lam_unique) && (((U || nnz_U) && m_u > m) || ((II || nnz_I) && n_i > n)))
        return fail(verbose, "cmfrec_hip: L1 together with side information beyond X is not implemented.");
    if (nonneg || nonneg_C || nonneg_D || l1_lam != 0 || l1_lam_unique) use_cg = false;    // collective.c:9568-9571 (any of them, unlike the explicit model)
    // For rows with few missing values the reference corrects a precomputed Gramian instead of summing the present entries
    // (factors_closed_form, common.c:762-790): the same solution with the Cholesky solver; under CG such attributes are solved in
    // closed form or restart from zero with k steps (:2958-2985) -- DenseNanSide::rules, handed to the session below.  The
    // non-negative / L1 solvers see another matrix layout -- not restated, so not offered.
    if (nan_side && (nonneg || nonneg_C || nonneg_D || l1_lam != 0 || l1_lam_unique))
        return fail(verbose, "cmfrec_hip: NaN in dense side information: not together with nonneg / L1.");
    if (precompute_for_predictions && precomputedBtB == nullptr)
        return fail(verbose, "cmfrec_hip: precompute_for_predictions needs the output buffers (cmfrec.h.in:760-778).");
    if (m <= 0 || n <= 0 || k + k_main <= 0) return fail(verbose, "cmfrec_hip: invalid dimensions.");
    for (size_t e = 0; e < nnz; e++)
        if (ixA[e] < 0 || ixA[e] >= m || ixB[e] < 0 || ixB[e] >= n) return fail(verbose, "cmfrec_hip: X index out of range.");
    real_t w_mult = 1;
    if (adjust_weight) {                                                  // collective.c:9776-9783
        w_mult = (real_t)((long double)nnz / (long double)((size_t)m * (size_t)n));
        w_main *= w_mult;
    }
    if (w_main_multiplier) *w_main_multiplier = w_mult;
    // per-matrix penalties: entries 2..5 = A, B, C, D (the bias slots are unused by this model, collective.c:9793-9809)
    real_t lam6[6], l16[6];
    for (int e = 0; e < 6; e++) { lam6[e] = lam_unique ? lam_unique[e] : lam; l16[e] = l1_lam_unique ? l1_lam_unique[e] : l1_lam; }
    if (w_main != (real_t)1) {                                            // collective.c:9786-9811
        lam /= w_main; l1_lam /= w_main; w_user /= w_main; w_item /= w_main;
        for (int e = 2; e < 6; e++) { lam6[e] /= w_main; l16[e] /= w_main; }
    }

    PhaseTimer tm;
    tm.lap("validate");
    SigGuard sig(handle_interrupt);
    std::vector<real_t> Xs;                                              // X is copied before edits (:9578-9599)
    if (apply_log_transf) {
        Xs.assign(X, X + nnz);
        for (auto &x : Xs) x = std::log(x);
    }
    tm.lap("log transform");

    // ---- side information: column means + centering, common.c:4938-4997 (only when the means are asked for) ----
    std::vector<real_t> Uc, Ic;
    auto center_cols = [](const real_t *M, int rows, int cols, real_t *means, std::vector<real_t> &out) {
        out.assign(M, M + (size_t)rows * cols);
        if (means == nullptr) return;
        for (int c = 0; c < cols; c++) means[c] = 0;
        for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) means[c] += M[(size_t)r * cols + c];
        for (int c = 0; c < cols; c++) means[c] /= (double)rows;
        for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) out[(size_t)r * cols + c] -= means[c];
    };
    if (U) center_cols(U, m_u, p, U_colmeans, Uc);
    if (II) center_cols(II, n_i, q, I_colmeans, Ic);

    const int k_totA = k_user + k + k_main, k_totB = k_item + k + k_main;
    const int_t m_max = std::max(m, m_u), n_max = std::max(n, n_i);     // rows of A / B (collective.c:9437-9440)
    if (reset_values) {                                                  // :9750-9774
        const bool fill_B = (II != nullptr || spI);
        cmfrng::random_parallel<real_t>(A, (size_t)m_max * k_totA, fill_B ? B : nullptr, fill_B ? (size_t)n_max * k_totB : 0, seed, false);
        if (use_cg) {
            if (!fill_B) memset(B, 0, (size_t)n_max * k_totB * sizeof(real_t));
            if (U || spU) memset(C, 0, (size_t)p * (k_user + k) * sizeof(real_t));
            if (II || spI) memset(D, 0, (size_t)q * (k_item + k) * sizeof(real_t));
        }
        // Cholesky: start values that are never read (the C / D / B steps run first) are left as passed, like the reference
    }
    if (!use_cg) finalize_chol = false;                                  // :9518
    tm.lap("start values");

    cmfrec_hip_model mdl;
    memset(&mdl, 0, sizeof mdl);
    mdl.implicit = 1; mdl.m = m_max; mdl.n = n_max; mdl.m_x = m; mdl.n_x = n;
    mdl.k = k; mdl.k_main = k_main; mdl.k_user = k_user; mdl.k_item = k_item;
    mdl.use_cg = use_cg; mdl.max_cg_steps = max_cg_steps; mdl.precondition_cg = precondition_cg; mdl.lam = lam;
    mdl.p = p; mdl.q = q; mdl.m_u = m_u; mdl.n_i = n_i; mdl.w_user = w_user; mdl.w_item = w_item;
    mdl.row_begin = 0; mdl.row_end = m_max; mdl.col_begin = 0; mdl.col_end = n_max;
    // device selection without touching the signature: CMFREC_HIP_DEVICES (one ordinal: that device; several: row-block shards)
    const std::vector<int> devs = devices_from_env();
    // Sharded: the plain model and the one with DENSE side information (C / D by partial sums + all-reduce, multi_sideinfo_step),
    // all solvers, constraints and penalties; sparse side information and side information beyond the shape of X keep to the
    // first listed device.
    const bool multi_ok = sharded_fit_wanted(devs) && !spU && !spI && m_u <= m && n_i <= n && m >= (int_t)devs.size() &&
                          n >= (int_t)devs.size() && zero_rows_A.empty() && zero_rows_B.empty();
    if (devs.size() > 1 && !multi_ok && verbose)
        printf("cmfrec_hip: CMFREC_HIP_DEVICES lists %d devices; this configuration (sparse side information / side information beyond X) "
               "runs on the first\n", (int)devs.size());
    if (multi_ok) {
        MultiDev md;
        int rc_loop = build_shards(md, devs, mdl, m_max, n_max, ixA, ixB, apply_log_transf ? Xs.data() : X, nnz, (real_t)0, alpha,
                                   [&](cmfrec_hip_session *sd, int d) {
            int rc2 = set_sideinfo_shard(md, sd, d, U ? Uc.data() : nullptr, m_u, p, II ? Ic.data() : nullptr, n_i, q);
            if (!rc2 && (nonneg || nonneg_C || nonneg_D)) rc2 = cmfrec_hip_session_set_nonneg(sd, nonneg, nonneg_C, nonneg_D, (int)max_cd_steps);
            if (!rc2 && l1_lam != 0) rc2 = cmfrec_hip_session_set_l1(sd, l1_lam, (int)max_cd_steps);
            if (!rc2) rc2 = cmfrec_hip_session_set_lam_unique(sd, lam6, l1_lam_unique ? l16 : nullptr, (int)max_cd_steps);
            if (!rc2) rc2 = cmfrec_hip_session_set_factors(sd, A, B, nullptr, nullptr, C, D);
            return rc2;
        });
        if (!rc_loop) rc_loop = multi_loop(md, mdl, (int)niter, finalize_chol, verbose);
        cmfrec_hip_session *s0 = md.sess.empty() ? nullptr : md.sess[0];
        if ((rc_loop == 0 || rc_loop == 3) && s0) {
            int rc2 = cmfrec_hip_session_get_factors(s0, A, B, nullptr, nullptr, C, D);
            if (rc2) rc_loop = rc2;
        }
        if ((rc_loop == 0 || rc_loop == 3) && s0 && precompute_for_predictions) {
            const int last_chol = (!use_cg || (finalize_chol && niter > 0)) ? 1 : 0;
            int rc2 = cmfrec_hip_session_precompute(s0, last_chol, 0, precomputedBtB, nullptr, hadU ? precomputedBeTBe : nullptr,
                                                    hadU ? precomputedBeTBeChol : nullptr, nullptr, nullptr);
            if (rc2) rc_loop = rc2;
            if (naz_U && !rc2) fill_CtUbias(precomputedCtUbias, C, U_colmeans, p, k_user + k, w_user);   // collective.c:9244-9252, :10115-10123
        }
        if (verbose && rc_loop == 0) printf("ALS procedure terminated successfully\n");
        return rc_loop;
    }
    cmfrec_hip_session *s = cmfrec_hip_session_create(&mdl, devs.empty() ? -1 : devs[0]);
    if (!s) { if (verbose) fprintf(stderr, "%s\n", cmfrec_hip_last_error()); const int ec = cmfrec_hip_last_error_code(); return ec ? ec : 1; }
    tm.lap("session create");
    // X := alpha * X and COO -> CSR + CSC happen on the device (coo_device.hpp), same entry order as helpers.c:1375-1491
    int rc = cmfrec_hip_session_set_X_coo(s, ixA, ixB, apply_log_transf ? Xs.data() : X, nnz, (real_t)0, alpha);
    std::vector<real_t>().swap(Xs);
    tm.lap("set_X_coo (upload, sort, bins)");
    if (!rc && !zero_rows_A.empty()) rc = cmfrec_hip_session_set_zero_rows(s, 'A', zero_rows_A.data(), (int)zero_rows_A.size());
    if (!rc && !zero_rows_B.empty()) rc = cmfrec_hip_session_set_zero_rows(s, 'B', zero_rows_B.data(), (int)zero_rows_B.size());
    if (!rc) rc = cmfrec_hip_session_set_sideinfo(s, U ? Uc.data() : nullptr, II ? Ic.data() : nullptr);
    if (!rc && spU) rc = cmfrec_hip_session_set_sideinfo_sparse(s, 'U', U_row, U_col, U_sp, nnz_U);
    if (!rc && spI) rc = cmfrec_hip_session_set_sideinfo_sparse(s, 'I', I_row, I_col, I_sp, nnz_I);
    // dense side information with NaN: the per-attribute rules of the dense C / D update (DenseNanSide::rules)
    for (int side = 0; side < 2 && !rc; side++) {
        const DenseNanSide &ns = side ? nanI : nanU;
        if (ns.na_col.empty() || !(side ? spI : spU)) continue;
        std::vector<unsigned char> mask; std::vector<real_t> mult;
        ns.rules((side ? k_item : k_user) + k, false, mask, mult);
        if (use_cg) rc = cmfrec_hip_session_set_closed_form_rows(s, side ? 'D' : 'C', mask.data());
        if (!rc && !mult.empty()) rc = cmfrec_hip_session_set_lambda_multipliers(s, side ? 'D' : 'C', mult.data());
    }
    if (!rc && (nonneg || nonneg_C || nonneg_D)) rc = cmfrec_hip_session_set_nonneg(s, nonneg, nonneg_C, nonneg_D, (int)max_cd_steps);
    if (!rc && l1_lam != 0) rc = cmfrec_hip_session_set_l1(s, l1_lam, (int)max_cd_steps);
    if (!rc && (lam_unique || l1_lam_unique))
        rc = cmfrec_hip_session_set_lam_unique(s, lam_unique ? lam6 : nullptr, l1_lam_unique ? l16 : nullptr, (int)max_cd_steps);
    if (!rc) rc = cmfrec_hip_session_set_factors(s, A, B, nullptr, nullptr, C, D);
    if (verbose && !rc) { printf("Starting ALS optimization routine\n\n"); fflush(stdout); }
    if (tm.on) cmfrec_hip_session_sync(s);
    tm.lap("set_factors");
    int rc_loop = rc ? rc : run_loop(s, mdl, niter, finalize_chol, verbose);
    if (tm.on) cmfrec_hip_session_sync(s);
    tm.lap("ALS iterations");
    if (rc_loop == 0 || rc_loop == 3) {
        int rc2 = cmfrec_hip_session_get_factors(s, A, B, nullptr, nullptr, C, D);
        if (rc2) rc_loop = rc2;
    }
    tm.lap("get_factors");
    if ((rc_loop == 0 || rc_loop == 3) && precompute_for_predictions) {   // collective.c:10056-10115 (also after an interrupt, :10034-10043)
        if (verbose) { printf("Finishing precomputed matrices..."); fflush(stdout); }
        const int last_chol = (!use_cg || (finalize_chol && niter > 0)) ? 1 : 0;
        int rc2 = cmfrec_hip_session_precompute(s, last_chol, 0, precomputedBtB, nullptr, hadU ? precomputedBeTBe : nullptr,
                                                hadU ? precomputedBeTBeChol : nullptr, nullptr, nullptr);
        if (rc2) rc_loop = rc2;
        if (naz_U && !rc2) fill_CtUbias(precomputedCtUbias, C, U_colmeans, p, k_user + k, w_user);   // collective.c:9244-9252, :10115-10123
        if (verbose) printf("  done\n");
        tm.lap("precompute epilogue");
    }
    cmfrec_hip_session_destroy(s);
    tm.lap("session destroy");
    if (verbose && rc_loop == 0) printf("ALS procedure terminated successfully\n");
    return rc_loop;
}

int_t fit_collective_explicit_als(
    real_t *biasA, real_t *biasB, real_t *A, real_t *B, real_t *C, real_t *D, real_t *Ai, real_t *Bi,
    bool add_implicit_features, bool reset_values, int_t seed, real_t *glob_mean,
    real_t *U_colmeans, real_t *I_colmeans, int_t m, int_t n, int_t k,
    int_t ixA[], int_t ixB[], real_t *X, size_t nnz, real_t *Xfull, real_t *weight,
    bool user_bias, bool item_bias, bool center, real_t lam, real_t *lam_unique,
    real_t l1_lam, real_t *l1_lam_unique, bool scale_lam, bool scale_lam_sideinfo, bool scale_bias_const,
    real_t *scaling_biasA, real_t *scaling_biasB, real_t *U, int_t m_u, int_t p, real_t *II, int_t n_i, int_t q,
    int_t U_row[], int_t U_col[], real_t *U_sp, size_t nnz_U, int_t I_row[], int_t I_col[], real_t *I_sp, size_t nnz_I,
    bool NA_as_zero_X, bool NA_as_zero_U, bool NA_as_zero_I, int_t k_main, int_t k_user, int_t k_item,
    real_t w_main, real_t w_user, real_t w_item, real_t w_implicit, int_t niter, int nthreads,
    bool verbose, bool handle_interrupt, bool use_cg, int_t max_cg_steps, bool precondition_cg, bool finalize_chol,
    bool nonneg, int_t max_cd_steps, bool nonneg_C, bool nonneg_D, bool precompute_for_predictions,
    bool include_all_X, real_t *B_plus_bias, real_t *precomputedBtB, real_t *precomputedTransBtBinvBt,
    real_t *precomputedBtXbias, real_t *precomputedBeTBeChol, real_t *precomputedBiTBi,
    real_t *precomputedTransCtCinvCt, real_t *precomputedCtCw, real_t *precomputedCtUbias)
{
    (void)max_cd_steps;
    (void)precomputedBiTBi;        // with add_implicit_features the prediction matrices are not produced here
    // collective.c:7308-7329
    if (k_user && U == nullptr && nnz_U == 0) return fail(verbose, "Cannot pass 'k_user' without U data.");
    if (k_item && II == nullptr && nnz_I == 0) return fail(verbose, "Cannot pass 'k_item' without I data.");
    if (k_main && Xfull == nullptr && nnz == 0) return fail(verbose, "Cannot pass 'k_main' without X data.");
    // sparse side information whose absent entries are zeros -> the dense route on the zero-filled matrix (ZeroFilledSide)
    ZeroFilledSide zfU, zfI;
    const bool naz_U = NA_as_zero_U && U == nullptr && nnz_U > 0, naz_I = NA_as_zero_I && II == nullptr && nnz_I > 0;
    std::vector<int_t> zero_rows_A, zero_rows_B;
    if (naz_U) {
        zero_rows_A = rows_without_data(m, ixA, nnz, U_row, nnz_U);
        const int e = zfU.build(m, m_u, p, U_row, U_col, U_sp, nnz_U);
        if (e) return fail(verbose, e == 2 ? "cmfrec_hip: NA_as_zero_U with more rows of U than X is not implemented."
                                    : e == 3 ? "cmfrec_hip: NA_as_zero_U runs on the zero-filled dense matrix (rows of X x p), which is larger than "
                                               "CMFREC_HIP_ZEROFILL_MAX_GB (default 8) here: not implemented at this size."
                                             : "cmfrec_hip: U index out of range.");
        U = zfU.dense.data(); m_u = m; nnz_U = 0; U_row = U_col = nullptr; U_sp = nullptr;
    }
    if (naz_I) {
        zero_rows_B = rows_without_data(n, ixB, nnz, I_row, nnz_I);
        const int e = zfI.build(n, n_i, q, I_row, I_col, I_sp, nnz_I);
        if (e) return fail(verbose, e == 2 ? "cmfrec_hip: NA_as_zero_I with more rows of I than X has columns is not implemented."
                                    : e == 3 ? "cmfrec_hip: NA_as_zero_I runs on the zero-filled dense matrix (columns of X x q), which is larger than "
                                               "CMFREC_HIP_ZEROFILL_MAX_GB (default 8) here: not implemented at this size."
                                             : "cmfrec_hip: I index out of range.");
        II = zfI.dense.data(); n_i = n; nnz_I = 0; I_row = I_col = nullptr; I_sp = nullptr;
    }
    // Dense X: the rows of the present entries go through the same row kernels as a sparse X (a row's system is the sum over
    // its present entries either way: factors_closed_form, common.c:762-1075; factors_explicit_cg_dense, :1615-1749).  What the
    // reference's dense cases add is the CHOICE of solver (optimizeA Cases 1-2, common.c:2787-3116), followed here per
    // half-step.  Without weights: a half-step whose rows are all or nearly all complete is a closed-form solve whatever
    // use_cg says (Case 1; its rows with many missing entries run k CG steps from zero there, :2944-2983 -- the exact solution
    // in exact arithmetic -- and the closed form here); otherwise (Case 2) a row that misses fewer than twice as many entries
    // as its system has unknowns is solved in closed form from the precomputed B^T B (factors_closed_form, :662, :759-790:
    // that branch comes before the CG one), the others by the solver asked for.  A half-step that has rows of both kinds under
    // use_cg runs both solvers (dense_cf_A / _B).  With weights every row takes the solver asked for.
    DenseX dx;
    bool dense_chol_A = false, dense_chol_B = false;
    std::vector<unsigned char> dense_cf_A, dense_cf_B;
    std::vector<real_t> dense_mult_A, dense_mult_B;
    bool unit_weights = false;
    // With side information (round 6, fixture g33): the side that has it goes through optimizeA_collective, whose dense-X branches are the
    // same systems over the present entries (collective.c:5115-5565 shared factorisation + row-by-row corrections; :5566-5968 row by
    // row; lambda's multiplier n - cnt_NA_x = the number of present entries, :1296-1302, no precomputed matrix with lambda inside) with
    // this choice of solver: closed form whatever use_cg says when X is complete or nearly complete on that orientation and the side
    // information is dense or missing-as-zero (:5121-5130), the solver asked for otherwise.  The side without side information keeps
    // optimizeA's rules above.  Side information on exactly the rows / columns of X (the rows beyond it would take optimizeA's dense
    // cases on a sub-block, :4832-5099), complete (no NaN), no weights.
    const bool dense_side_A = Xfull && (U != nullptr || nnz_U > 0), dense_side_B = Xfull && (II != nullptr || nnz_I > 0);
    const bool had_dense_X = Xfull != nullptr;
    if (Xfull) {
        if (add_implicit_features || NA_as_zero_X)
            return fail(verbose, "cmfrec_hip: dense X is implemented for the model without implicit features and NA_as_zero_X.");
        // (with weights the compiled reference corrupts its heap -- "free(): invalid pointer" on a dense X with holes, dense U and I and
        //  dense weights, either solver, both precisions -- so that combination has nothing to be pinned against)
        if ((dense_side_A || dense_side_B) && (weight || NA_as_zero_U || NA_as_zero_I))
            return fail(verbose, "cmfrec_hip: dense X with side information: not together with observation weights or NA_as_zero_U / _I.");
        if ((dense_side_A && m_u != m) || (dense_side_B && n_i != n))
            return fail(verbose, "cmfrec_hip: dense X with side information: U / I must have exactly the rows / columns of X.");
        if (m <= 0 || n <= 0) return fail(verbose, "cmfrec_hip: invalid dimensions.");
        dx.convert(Xfull, weight, m, n);
        if (dx.val.empty()) return fail(verbose, "cmfrec_hip: 'X' has all entries missing.");
        const int_t fewA = 2 * (k + k_main + (user_bias ? 1 : 0)), fewB = 2 * (k + k_main + (item_bias ? 1 : 0));
        // Under scale_lam a row that misses fewer than 2 k entries keeps the n lam of a complete row -- its matrix is the precomputed
        // B^T B + n lam I minus the missing rows (factors_closed_form, common.c:759-790 with BtB_has_diag; :3031-3032) -- while the
        // others take lam times their present entries.  Restated through the weighted kernels: unit weights (a multiplication by
        // one: the unweighted numbers bit for bit) and the per-row multipliers handed over after the bias start values
        // (cmfrec_hip_session_set_lambda_multipliers).
        if (!weight && (scale_lam || scale_lam_sideinfo)) {
            bool any = false;
            for (int_t r = 0; r < m && !any && !dense_side_A; r++) any = (dx.na_row[r] > 0 && dx.na_row[r] < fewA);
            for (int_t c = 0; c < n && !any && !dense_side_B; c++) any = (dx.na_col[c] > 0 && dx.na_col[c] < fewB);
            if (any) {
                // (a side with side information: the present entries, like a sparse X -- collective.c:1296-1302)
                dense_mult_A.resize((size_t)m); dense_mult_B.resize((size_t)n);
                for (int_t r = 0; r < m; r++) dense_mult_A[r] = (!dense_side_A && dx.na_row[r] < fewA) ? (real_t)n : (dx.na_row[r] < n ? (real_t)(n - dx.na_row[r]) : (real_t)1);
                for (int_t c = 0; c < n; c++) dense_mult_B[c] = (!dense_side_B && dx.na_col[c] < fewB) ? (real_t)m : (dx.na_col[c] < m ? (real_t)(m - dx.na_col[c]) : (real_t)1);
                // (the bias start values of a weighted session do not restate scale_lam_sideinfo / scale_bias_const: say so here,
                //  before anything is uploaded, and in terms of what the caller passed -- no weights)
                if ((scale_lam_sideinfo || scale_bias_const) && user_bias && item_bias && reset_values)
                    return fail(verbose, "cmfrec_hip: dense X whose rows / columns miss only a few entries, under scale_lam together with "
                                         "scale_lam_sideinfo / scale_bias_const and both biases, is not implemented (its per-row lambda "
                                         "multipliers ride on the weighted kernels, whose bias start values do not restate those options).");
                dx.w.assign(dx.val.size(), (real_t)1);
                unit_weights = true;
            }
        }
        ixA = dx.row.data(); ixB = dx.col.data(); X = dx.val.data(); nnz = dx.val.size();
        if (weight || unit_weights) weight = dx.w.data();
        if (!weight || unit_weights) {
            // Case 2 under use_cg: closed form for the rows that miss few entries, CG for the others (rows without any entry are zero)
            auto case2_chol = [&](const std::vector<int_t> &na, int_t other, int_t few, bool &mixed) {
                size_t n_few = 0, n_many = 0;
                for (int_t v : na) { if (v < few) n_few++; else if (v < other) n_many++; }
                mixed = (n_few > 0 && n_many > 0);
                return n_many == 0;
            };
            bool mixA = false, mixB = false;
            dense_chol_A = dx.full || dx.near_row || case2_chol(dx.na_row, n, fewA, mixA);
            dense_chol_B = dx.full || dx.near_col || case2_chol(dx.na_col, m, fewB, mixB);
            if (dense_side_A) { dense_chol_A = (dx.full || dx.near_row) && U != nullptr; mixA = false; }
            if (dense_side_B) { dense_chol_B = (dx.full || dx.near_col) && II != nullptr; mixB = false; }
            // a half-step with rows of both kinds under use_cg: the rows that miss few entries are marked for the closed form
            // (cmfrec_hip_session_set_closed_form_rows), the update runs both solvers
            if (use_cg && !nonneg && l1_lam == 0 && !l1_lam_unique) {
                if (!(dx.full || dx.near_row) && mixA) { dense_cf_A.resize((size_t)m); for (int_t r = 0; r < m; r++) dense_cf_A[r] = dx.na_row[r] < fewA; }
                if (!(dx.full || dx.near_col) && mixB) { dense_cf_B.resize((size_t)n); for (int_t c = 0; c < n; c++) dense_cf_B[c] = dx.na_col[c] < fewB; }
            }
        }
        Xfull = nullptr;
    }
    // NA_as_zero_X (sparse X whose absent entries are zeros): every half-step shares one matrix over its rows -- optimizeA Case 3
    // without side information on that side (common.c:3118-3205: closed form whatever use_cg says), optimizeA_collective with
    // the factorised shared block matrix (collective.c:5607-5617, :5700-5716) with dense complete side information
    if (NA_as_zero_X && (nonneg || l1_lam != 0 || l1_lam_unique ||
                         (scale_bias_const && (scale_lam || scale_lam_sideinfo) && (user_bias || item_bias))))
        return fail(verbose, "cmfrec_hip: NA_as_zero_X is implemented for the model without nonneg / L1 and scale_bias_const.");
    // the matrices for predictions (round 5): for the model without side information -- B_plus_bias, BtB, TransBtBinvBt as ever, plus
    // BtXbias, the constant every new row's right-hand side receives (collective.c:8938-8986)
    if (NA_as_zero_X && precompute_for_predictions && (U || II || nnz_U || nnz_I || add_implicit_features))
        return fail(verbose, "cmfrec_hip: NA_as_zero_X with precompute_for_predictions: the model without side information and implicit features.");
    // ... with implicit features (round 5): the model without side information and weights, closed form (optimizeA_collective's
    // general branch on a matrix all rows share, collective.c:8612 / :8783 -> :1534-1846)
    // (use_cg is accepted: the reference takes its closed-form Case 1 whatever the solver asked for, collective.c:5121-5130)
    // (round 6, fixture g37: with dense or sparse side information on exactly the rows / columns of X)
    // (and with observation weights: fixture g39)
    // ... with SPARSE side information (round 5): row by row on the shared B^T B plus the rank-1 terms of the row's own attributes
    // (collective_closed_form_block with prefer_BtB, collective.c:1534-1846) -- closed form, side information on exactly the rows /
    // columns of X, no weights
    if (NA_as_zero_X && ((U == nullptr && nnz_U) || (II == nullptr && nnz_I))) {
        if (NA_as_zero_U || NA_as_zero_I)
            return fail(verbose, "cmfrec_hip: NA_as_zero_X with sparse side information: not together with NA_as_zero_U / _I.");
        if ((U == nullptr && nnz_U && m_u != m) || (II == nullptr && nnz_I && n_i != n))
            return fail(verbose, "cmfrec_hip: NA_as_zero_X with side information: U / I must have exactly the rows / columns of X.");
    }
    // ... with observation weights (round 5): optimizeA Case 4's NA_as_zero + weight branches (common.c:3209-3302, :846-907,
    // :1293-1441) without side information; with DENSE complete side information the rows with entries leave the shared
    // factorisation for collective_closed_form_block's general branch (collective.c:1367-1372, :1534-1846), closed form.  Not
    // with start values for the biases: the reference's own (initialize_biases with NA_as_zero and weights) index the item biases
    // by row inside the item sweep (common.c:4727-4731).
    if (NA_as_zero_X && weight != nullptr) {
        if ((user_bias || item_bias) && reset_values)
            return fail(verbose, "cmfrec_hip: NA_as_zero_X with observation weights: pass start values for the biases (reset_values = false); "
                                 "the reference's own start values are not defined for this combination.");
        if (use_cg && !precondition_cg && k + k_main + 1 > 64)
            return fail(verbose, "cmfrec_hip: NA_as_zero_X with observation weights under CG: k + k_main + bias <= 64 (precondition_cg takes more).");
    }
    if (NA_as_zero_X && (U || II)) {
        // (use_cg is accepted: with a factorised shared block matrix the reference takes the closed form whatever the solver asked for)
        if ((U && m_u != m) || (II && n_i != n))
            return fail(verbose, "cmfrec_hip: NA_as_zero_X with side information: U / I must have exactly the rows / columns of X.");
        for (size_t e = 0; U && e < (size_t)m_u * (size_t)p; e++)
            if (std::isnan(U[e])) return fail(verbose, "cmfrec_hip: NA_as_zero_X with side information: NaN in U is not implemented.");
        for (size_t e = 0; II && e < (size_t)n_i * (size_t)q; e++)
            if (std::isnan(II[e])) return fail(verbose, "cmfrec_hip: NA_as_zero_X with side information: NaN in I is not implemented.");
    }
    // dense side information with NaN -> the sparse route on its centred present entries
    DenseNanSide nanU, nanI;
    const bool hadU = (U != nullptr);
    bool nan_side = false;
    if (U && m_u > 0 && p > 0 && nanU.convert(U, m_u, p, U_colmeans)) {
        U = nullptr; U_row = nanU.row.data(); U_col = nanU.col.data(); U_sp = nanU.val.data(); nnz_U = nanU.val.size(); nan_side = true;
    } else if (U == nullptr && nnz_U > 0 && U_row && U_col && U_sp) {
        bool ok = true;
        for (size_t e = 0; e < nnz_U && ok; e++) ok = (U_col[e] >= 0 && U_col[e] < p);
        if (ok) sparse_colmeans(U_col, U_sp, nnz_U, p, U_colmeans);
    }
    if (II && n_i > 0 && q > 0 && nanI.convert(II, n_i, q, I_colmeans)) {
        II = nullptr; I_row = nanI.row.data(); I_col = nanI.col.data(); I_sp = nanI.val.data(); nnz_I = nanI.val.size(); nan_side = true;
    } else if (II == nullptr && nnz_I > 0 && I_row && I_col && I_sp) {
        bool ok = true;
        for (size_t e = 0; e < nnz_I && ok; e++) ok = (I_col[e] >= 0 && I_col[e] < q);
        if (ok) sparse_colmeans(I_col, I_sp, nnz_I, q, I_colmeans);
    }
    if (nan_side && nnz_U == 0 && nanU.row.empty() && hadU && U == nullptr)
        return fail(verbose, "cmfrec_hip: U has no present entries.");
    // implicit features (Ai, Bi on the binary "was observed" matrix): closed-form solves, dense, sparse (round 6, fixture g32) or no
    // side information inside the shape of X, prediction matrices not produced
    if (add_implicit_features) {
        if (!Ai || !Bi) return fail(verbose, "cmfrec_hip: add_implicit_features needs the Ai and Bi outputs.");
        if (((U || nnz_U) && m_u > m) || ((II || nnz_I) && n_i > n))
            return fail(verbose, "cmfrec_hip: implicit features with side information beyond X are not implemented.");
        if (precompute_for_predictions)
            return fail(verbose, "cmfrec_hip: implicit features: precompute_for_predictions is not implemented.");
        // the reference itself crashes on add_implicit_features with nonneg or an L1 penalty (its Ai / Bi updates are handed
        // a NULL thread-local buffer, collective.c:8487-8489): nothing to pin against, so not offered
        if (nonneg || l1_lam != 0 || l1_lam_unique)
            return fail(verbose, "cmfrec_hip: implicit features with nonneg / L1 are not implemented.");
        if (!(w_implicit > 0)) return fail(verbose, "cmfrec_hip: w_implicit must be positive.");
    }
    // side information: dense (no NaN) or sparse COO (missing = absent).  Sparse: Cholesky updates only, rows within X.
    const bool spU = (U == nullptr && nnz_U > 0), spI = (II == nullptr && nnz_I > 0);
    if ((spU && (m_u > m || !U_row || !U_col || !U_sp)) || (spI && (n_i > n || !I_row || !I_col || !I_sp)))
        return fail(verbose, "cmfrec_hip: sparse side information must be COO triplets with rows inside X.");
    for (size_t e = 0; spU && e < nnz_U; e++)
        if (U_row[e] < 0 || U_row[e] >= m_u || U_col[e] < 0 || U_col[e] >= p) return fail(verbose, "cmfrec_hip: U index out of range.");
    for (size_t e = 0; spI && e < nnz_I; e++)
        if (I_row[e] < 0 || I_row[e] >= n_i || I_col[e] < 0 || I_col[e] >= q) return fail(verbose, "cmfrec_hip: I index out of range.");
    if ((l1_lam != 0 || l1_lam_unique) && (((U || nnz_U) && m_u > m) || ((II || nnz_I) && n_i > n)))
        return fail(verbose, "cmfrec_hip: L1 together with side information beyond X is not implemented.");
    if (nonneg || l1_lam != 0 || l1_lam_unique) use_cg = false;           // collective.c:7474-7479
    // For rows with few missing values the reference corrects a precomputed Gramian instead of summing the present entries
    // (factors_closed_form, common.c:762-790).  Unscaled lambda + Cholesky: the same solution.  Under scale_lam that Gramian
    // already carries lam x (all rows) (:3031-3032) and under CG such attributes are solved in closed form or restart from zero
    // with k steps (:2958-2985): per-attribute rules (DenseNanSide::rules, round 5) handed to the session below.  The
    // non-negative / L1 solvers see another matrix layout -- not restated, so not offered.
    if (nan_side && (nonneg || nonneg_C || nonneg_D || l1_lam != 0 || l1_lam_unique))
        return fail(verbose, "cmfrec_hip: NaN in dense side information: not together with nonneg / L1.");
    if (nan_side && had_dense_X)
        return fail(verbose, "cmfrec_hip: dense X with side information: NaN in U / I is not implemented.");
    if (precompute_for_predictions && precomputedBtB == nullptr)
        return fail(verbose, "cmfrec_hip: precompute_for_predictions needs the output buffers (cmfrec.h.in:760-778).");
    // observation weights (one per entry of X): every row solver and the start values of the biases take them; the lambda
    // multipliers of scale_lam become sums of weights (collective.c:7931-8008).  Not together with the options whose weight
    // bookkeeping is not restated: NaN side information, scale_lam_sideinfo, scale_bias_const.
    // (scale_lam_sideinfo with weights under NA_as_zero_X: the multiplier is the weights' sum + the absent entries + p, no start values)
    // Sparse side information (round 6): the row's attributes are the second gather source of the weighted row solvers, unweighted
    // themselves (collective.c:1636-1653 beside :1673-1699; block CG :2187-2208 beside :2292-2298) -- fixture g31.
    // (weights + sparse side information + scale_lam_sideinfo: the reference's multipliers add `U_csr_p[row+1] - U_csr[row]` -- a VALUE of
    //  U where its row pointer is meant, collective.c:8087, :8106 -- so there is no meaningful number to agree with: refused)
    if (weight && scale_lam_sideinfo && (spU || spI))
        return fail(verbose, "cmfrec_hip: observation weights with sparse side information under scale_lam_sideinfo are not implemented "
                             "(the reference's lambda multipliers for this combination read a value of U / I in place of a row pointer).");
    // (round 6, fixture g38: with implicit features too -- the weighted row solvers with the implicit-features term, whose own
    //  gather-sum and Ai / Bi updates take no weights, collective.c:1757-1771, :8449-8535)
    if (weight && (nan_side || (scale_lam_sideinfo && !NA_as_zero_X) ||
                   (scale_bias_const && scale_lam && (user_bias || item_bias))))
        return fail(verbose, "cmfrec_hip: observation weights together with NaN side information / scale_lam_sideinfo / "
                             "scale_bias_const are not implemented.");
    if (U == nullptr && !spU) { m_u = 0; p = 0; }
    if (II == nullptr && !spI) { n_i = 0; q = 0; }
    if (m <= 0 || n <= 0 || nnz == 0) return fail(verbose, "cmfrec_hip: invalid dimensions.");
    for (size_t e = 0; e < nnz; e++)
        if (ixA[e] < 0 || ixA[e] >= m || ixB[e] < 0 || ixB[e] >= n) return fail(verbose, "cmfrec_hip: X index out of range.");

    SigGuard sig(handle_interrupt);
    scale_lam = scale_lam || scale_lam_sideinfo;                          // :7465
    if (!use_cg) finalize_chol = false;                                   // :7481
    // per-matrix penalties, order: user bias, item bias, A, B, C, D (collective.c:430)
    real_t lam6[6], l16[6];
    for (int e = 0; e < 6; e++) { lam6[e] = lam_unique ? lam_unique[e] : lam; l16[e] = l1_lam_unique ? l1_lam_unique[e] : l1_lam; }
    if (w_main != (real_t)1) {                                            // :7497-7521
        lam /= w_main; l1_lam /= w_main; w_user /= w_main; w_item /= w_main; w_implicit /= w_main;
        for (int e = 0; e < 6; e++) { lam6[e] /= w_main; l16[e] /= w_main; }
    }
    const bool has_bias = user_bias || item_bias;
    // scale_bias_const: the biases' lambda is scaled by one constant -- the mean over the rows of (entries + attributes
    // counted under scale_lam_sideinfo) -- instead of row by row (collective.c:7555-7556, :8026-8048, :8071-8160)
    if (!scale_lam || !has_bias) scale_bias_const = false;
    if (scale_bias_const) {
        if (add_implicit_features || spU || spI || nan_side)
            return fail(verbose, "cmfrec_hip: scale_bias_const with implicit features / sparse or NaN side information is not implemented.");
        if (item_bias && !user_bias && !use_cg)
            return fail(verbose, "cmfrec_hip: scale_bias_const with only an item bias and the Cholesky solver: the reference leaves "
                                 "scaling_biasB unset (collective.c:7934).");
        if (!scaling_biasA || !scaling_biasB) return fail(verbose, "cmfrec_hip: scale_bias_const needs the scaling_biasA / scaling_biasB outputs.");
        auto mean_count = [&](const int_t *ix, int_t rows, int_t extra, int_t extra_rows) {
            std::vector<size_t> cnt((size_t)rows, 0);
            for (size_t e = 0; e < nnz; e++) cnt[ix[e]]++;
            double wmean = 0;
            for (int_t r = 0; r < rows; r++) {
                const real_t w = (real_t)(cnt[r] + (cnt[r] == 0)) + (real_t)((scale_lam_sideinfo && r < extra_rows) ? extra : 0);
                wmean += ((double)w - wmean) / (double)(r + 1);
            }
            return (real_t)wmean;
        };
        if (user_bias) { *scaling_biasA = mean_count(ixA, m, U ? p : 0, m_u); lam6[0] *= *scaling_biasA; l16[0] *= *scaling_biasA; }
        if (item_bias) { *scaling_biasB = mean_count(ixB, n, II ? q : 0, n_i); lam6[1] *= *scaling_biasB; l16[1] *= *scaling_biasB; }
    }
    const int k_totA = k_user + k + k_main, k_totB = k_item + k + k_main;
    const int_t m_max = std::max(m, m_u), n_max = std::max(n, n_i);      // rows of A / B (collective.c:7332-7335)

    // ---- global mean, common.c:3494-3524 + :3603 (nthreads selects running mean vs sum/cnt); the
    //      subtraction itself happens on the device while the CSR / CSC are built ----
    PhaseTimer tm;
    real_t gm = 0;
    if (center && NA_as_zero_X && weight) {
        // weighted mean of the entries (common.c:3558-3584; with 8 threads or more the unweighted sum over the sum of the weights,
        // quirk Q13), then DIVIDED by the weights' share of all cells, wsum / (wsum + m n - nnz), as the reference does
        // (:3590-3594; the sum of the weights there is a compensated sum, helpers.c:1691-1707)
        double xsum = 0, wsum = 2.220446049250313e-16;
        if (nthreads >= 8) {
            wsum = 0;
            for (size_t e = 0; e < nnz; e++) { xsum += (double)X[e]; wsum += (double)weight[e]; }
            gm = (real_t)(xsum / wsum);
        } else {
            for (size_t e = 0; e < nnz; e++) { wsum += (double)weight[e]; xsum += (((double)X[e] - xsum) * (double)weight[e]) / wsum; }
            gm = (real_t)xsum;
        }
        double err = 0, res = 0;
        for (size_t e = 0; e < nnz; e++) { const double diff = (double)weight[e] - err; const double temp = res + diff; err = (temp - res) - diff; res = temp; }
        const long double wl = (long double)res;
        gm = (real_t)((long double)gm / (wl / (wl + ((long double)m * (long double)n - (long double)nnz))));
        if (std::fabs(gm) < std::sqrt(EPS_T)) gm = 0;
    } else if (center && NA_as_zero_X) {
        // mean over all m x n cells: the mean of the entries x nnz / (m n) (common.c:3494-3523); the stored values stay as
        // they are (:3600-3607), the mean enters every right-hand side instead (collective.c:8573-8600, :8756-8787)
        double xsum = 0;
        if (nthreads >= 8) {
            for (size_t e = 0; e < nnz; e++) xsum += X[e];
            gm = (real_t)(xsum / (double)nnz);
        } else {
            size_t cnt = 0;
            for (size_t e = 0; e < nnz; e++) xsum += (X[e] - xsum) / (double)(++cnt);
            gm = (real_t)xsum;
        }
        gm = (real_t)((long double)gm * ((long double)nnz / ((long double)m * (long double)n)));
        if (std::fabs(gm) < std::sqrt(EPS_T)) gm = 0;
    } else if (center && weight) {
        // weighted running mean, common.c:3574-3584.  With 8 threads or more the reference divides the UNWEIGHTED sum of X by
        // the sum of the weights (:3561-3571) -- not a mean, but the number a caller with nthreads >= 8 receives (Python:
        // nthreads = -1 on a host of 8 cores or more), so it is reproduced (fixture g23).
        double xsum = 0, wsum = 2.220446049250313e-16;
        if (nthreads >= 8) {
            wsum = 0;
            for (size_t e = 0; e < nnz; e++) { xsum += (double)X[e]; wsum += (double)weight[e]; }
            xsum = (double)(real_t)(xsum / wsum);
        } else
            for (size_t e = 0; e < nnz; e++) { wsum += (double)weight[e]; xsum += (((double)X[e] - xsum) * (double)weight[e]) / wsum; }
        gm = (real_t)xsum;
        if (nonneg) gm = std::max(gm, (real_t)0);                         // :3604-3605
        if (std::fabs(gm) < std::sqrt(EPS_T)) gm = 0;
    } else if (center) {
        double xsum = 0;
        if (nthreads >= 8) {
            for (size_t e = 0; e < nnz; e++) xsum += X[e];
            gm = (real_t)(xsum / (double)nnz);
        } else {
            size_t cnt = 0;
            for (size_t e = 0; e < nnz; e++) xsum += (X[e] - xsum) / (double)(++cnt);
            gm = (real_t)xsum;
        }
        if (nonneg) gm = std::max(gm, (real_t)0);                         // common.c:3604-3605
        if (std::fabs(gm) < std::sqrt(EPS_T)) gm = 0;
    }
    *glob_mean = gm;
    tm.lap("global mean");

    // ---- side information: column means + centering, common.c:4938-4997 ----
    std::vector<real_t> Uc, Ic;
    auto center_cols = [](const real_t *M, int rows, int cols, real_t *means, std::vector<real_t> &out) {
        out.assign(M, M + (size_t)rows * cols);
        if (means == nullptr) return;                                     // no centring asked for (common.c:4938)
        for (int c = 0; c < cols; c++) means[c] = 0;
        for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) means[c] += M[(size_t)r * cols + c];
        for (int c = 0; c < cols; c++) means[c] /= (double)rows;
        for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) out[(size_t)r * cols + c] -= means[c];
    };
    if (U) center_cols(U, m_u, p, U_colmeans, Uc);
    if (II) center_cols(II, n_i, q, I_colmeans, Ic);

    tm.lap("side info centring");
    // ---- factor start values, collective.c:8241-8274 ----
    if (reset_values) {
        const bool fill_B = (II != nullptr || spI || add_implicit_features);
        cmfrng::random_parallel<real_t>(A, (size_t)m_max * k_totA, fill_B ? B : nullptr, fill_B ? (size_t)n_max * k_totB : 0, seed, true);
        if (nonneg) {                                                     // :8256-8263: non-negative start values
            for (size_t e = 0; e < (size_t)m_max * k_totA; e++) A[e] = std::fabs(A[e]);
            if (fill_B) for (size_t e = 0; e < (size_t)n_max * k_totB; e++) B[e] = std::fabs(B[e]);
        }
        if (use_cg) {
            if (!fill_B) memset(B, 0, (size_t)n_max * k_totB * sizeof(real_t));
            if (U || spU) memset(C, 0, (size_t)p * (k_user + k) * sizeof(real_t));
            if (II || spI) memset(D, 0, (size_t)q * (k_item + k) * sizeof(real_t));
        }
    }

    // dense X: rows / columns without a present entry are zero in the reference (optimizeA, common.c:2925-2929; factors_closed_form,
    // :667-677) -- factors and bias -- whereas the sparse path leaves such rows at their start values
    auto zero_empty_dense = [&]() {
        if (dx.na_row.empty()) return;
        // (a side with side information solves such rows from their attributes: optimizeA_collective, collective.c:1285-1331)
        for (int_t r = 0; r < m && !dense_side_A; r++) if (dx.na_row[r] == n) { memset(A + (size_t)r * k_totA, 0, (size_t)k_totA * sizeof(real_t)); if (biasA) biasA[r] = 0; }
        for (int_t c = 0; c < n && !dense_side_B; c++) if (dx.na_col[c] == m) { memset(B + (size_t)c * k_totB, 0, (size_t)k_totB * sizeof(real_t)); if (biasB) biasB[c] = 0; }
    };
    if (niter > 0) zero_empty_dense();
    cmfrec_hip_model mdl;
    memset(&mdl, 0, sizeof mdl);
    mdl.implicit = 0; mdl.m = m_max; mdl.n = n_max; mdl.m_x = m; mdl.n_x = n;
    mdl.k = k; mdl.k_main = k_main; mdl.k_user = k_user; mdl.k_item = k_item;
    mdl.user_bias = user_bias; mdl.item_bias = item_bias; mdl.scale_lam = scale_lam; mdl.scale_lam_sideinfo = scale_lam_sideinfo;
    mdl.use_cg = use_cg; mdl.max_cg_steps = max_cg_steps; mdl.precondition_cg = precondition_cg; mdl.p = p; mdl.q = q; mdl.m_u = m_u; mdl.n_i = n_i;
    mdl.lam = lam; mdl.w_user = w_user; mdl.w_item = w_item;
    mdl.row_begin = 0; mdl.row_end = m_max; mdl.col_begin = 0; mdl.col_end = n_max;
    const std::vector<int> devs = devices_from_env();                  // CMFREC_HIP_DEVICES: one ordinal = that device, several = row-block shards
    // Several devices: the plain model and the one with DENSE side information (biases, centring, CG / Cholesky, non-negativity,
    // L1, per-matrix penalties) as row-block shards that exchange the updated rows (MultiDev).  The bias start values are those
    // of the single-device driver: they are computed by a temporary session that holds the whole X on the first device (the
    // sweeps alternate over all rows and all columns, common.c:4410-4909), then handed to every shard.
    // (a dense X keeps to one device: its half-steps follow the reference's per-half-step choice of solver, dense_chol_A / _B,
    // and its empty rows are zeroed afterwards -- neither is part of multi_loop)
    const bool multi_ok = sharded_fit_wanted(devs) && !spU && !spI && m_u <= m && n_i <= n && !add_implicit_features && !NA_as_zero_X &&
                          !weight && dx.na_row.empty() && m >= (int_t)devs.size() && n >= (int_t)devs.size() && zero_rows_A.empty() &&
                          zero_rows_B.empty();
    if (devs.size() > 1 && !multi_ok && verbose)
        printf("cmfrec_hip: CMFREC_HIP_DEVICES lists %d devices; this configuration (sparse side information / side information beyond X / "
               "weights / implicit features / NA_as_zero / dense X) runs on the first\n", (int)devs.size());
    if (multi_ok) {
        auto configure = [&](cmfrec_hip_session *sd) {
            int rc2 = 0;
            if (nonneg || nonneg_C || nonneg_D) rc2 = cmfrec_hip_session_set_nonneg(sd, nonneg, nonneg_C, nonneg_D, (int)max_cd_steps);
            if (!rc2 && l1_lam != 0) rc2 = cmfrec_hip_session_set_l1(sd, l1_lam, (int)max_cd_steps);
            if (!rc2 && (lam_unique || l1_lam_unique || scale_bias_const))
                rc2 = cmfrec_hip_session_set_lam_unique(sd, (lam_unique || scale_bias_const) ? lam6 : nullptr,
                                                        (l1_lam_unique || (scale_bias_const && l1_lam != 0)) ? l16 : nullptr, (int)max_cd_steps);
            if (!rc2 && scale_bias_const) rc2 = cmfrec_hip_session_set_scale_bias_const(sd, 1);
            return rc2;
        };
        int rc = 0;
        if (has_bias && reset_values) {
            cmfrec_hip_session *t = cmfrec_hip_session_create(&mdl, devs[0]);
            if (!t) { if (verbose) fprintf(stderr, "%s\n", cmfrec_hip_last_error()); const int ec = cmfrec_hip_last_error_code(); return ec ? ec : 1; }
            rc = cmfrec_hip_session_set_X_coo_weighted(t, ixA, ixB, X, nullptr, nnz, gm, (real_t)1);
            if (!rc) rc = configure(t);          // (the sweeps need the attribute COUNTS under scale_lam_sideinfo, mdl.p / mdl.q, not U / I)
            if (!rc) rc = cmfrec_hip_session_set_factors(t, A, B, nullptr, nullptr, nullptr, nullptr);
            real_t lam_u = lam6[0], lam_i = lam6[1];                      // collective.c:8178, :8197, :8218-8219
            if (std::fabs(lam_u) < EPS_T) lam_u = EPS_T;
            if (std::fabs(lam_i) < EPS_T) lam_i = EPS_T;
            if (!rc) rc = cmfrec_hip_session_init_biases(t, lam_u, lam_i);
            if (!rc) rc = cmfrec_hip_session_get_factors(t, nullptr, nullptr, biasA, biasB, nullptr, nullptr);
            cmfrec_hip_session_destroy(t);
            if (rc) return rc;
        }
        MultiDev md;
        if (!rc) rc = build_shards(md, devs, mdl, m_max, n_max, ixA, ixB, X, nnz, gm, (real_t)1, [&](cmfrec_hip_session *sd, int d) {
            int rc2 = set_sideinfo_shard(md, sd, d, U ? Uc.data() : nullptr, m_u, p, II ? Ic.data() : nullptr, n_i, q);
            if (!rc2) rc2 = configure(sd);
            if (!rc2) rc2 = cmfrec_hip_session_set_factors(sd, A, B, has_bias ? biasA : nullptr, has_bias ? biasB : nullptr, C, D);
            return rc2;
        });
        int rc_loop = rc ? rc : multi_loop(md, mdl, (int)niter, finalize_chol, verbose);
        cmfrec_hip_session *s0 = md.sess.empty() ? nullptr : md.sess[0];
        if ((rc_loop == 0 || rc_loop == 3) && s0) {
            const int rc2 = cmfrec_hip_session_get_factors(s0, A, B, biasA, biasB, C, D);
            if (rc2) rc_loop = rc2;
        }
        if ((rc_loop == 0 || rc_loop == 3) && s0 && precompute_for_predictions) {   // collective.c:8936-9249
            const int last_chol = (!use_cg || (finalize_chol && niter > 0)) ? 1 : 0;
            int rc2 = cmfrec_hip_session_precompute(s0, last_chol, include_all_X ? 1 : 0, precomputedBtB, precomputedTransBtBinvBt, nullptr,
                                                    hadU ? precomputedBeTBeChol : nullptr, hadU ? precomputedCtCw : nullptr,
                                                    hadU ? precomputedTransCtCinvCt : nullptr);
            if (rc2) rc_loop = rc2;
            if (naz_U && !rc2) fill_CtUbias(precomputedCtUbias, C, U_colmeans, p, k_user + k, w_user);   // collective.c:9244-9252, :10115-10123
            if (!rc2 && user_bias && B_plus_bias) {                           // append_ones_last_col, :8908-8920
                for (int_t c = 0; c < n_max; c++) {
                    memcpy(B_plus_bias + (size_t)c * (k_totB + 1), B + (size_t)c * k_totB, (size_t)k_totB * sizeof(real_t));
                    B_plus_bias[(size_t)c * (k_totB + 1) + k_totB] = 1;
                }
            }
        }
        if (verbose && rc_loop == 0) printf("ALS procedure terminated successfully\n");
        return rc_loop;
    }
    cmfrec_hip_session *s = cmfrec_hip_session_create(&mdl, devs.empty() ? -1 : devs[0]);
    if (!s) { if (verbose) fprintf(stderr, "%s\n", cmfrec_hip_last_error()); const int ec = cmfrec_hip_last_error_code(); return ec ? ec : 1; }
    tm.lap("start values + session");
    // X - mean, COO -> CSR + CSC and the bias start values are computed on the device (coo_device.hpp)
    int rc = cmfrec_hip_session_set_X_coo_weighted(s, ixA, ixB, X, weight, nnz, NA_as_zero_X ? (real_t)0 : gm, (real_t)1);
    if (!rc && NA_as_zero_X) rc = cmfrec_hip_session_set_NA_as_zero_X(s, 1, center ? 1 : 0, gm);
    tm.lap("set_X_coo (upload, sort, bins)");
    if (!rc && !dense_cf_A.empty()) rc = cmfrec_hip_session_set_closed_form_rows(s, 'A', dense_cf_A.data());
    if (!rc && !dense_cf_B.empty()) rc = cmfrec_hip_session_set_closed_form_rows(s, 'B', dense_cf_B.data());
    if (!rc && !zero_rows_A.empty()) rc = cmfrec_hip_session_set_zero_rows(s, 'A', zero_rows_A.data(), (int)zero_rows_A.size());
    if (!rc && !zero_rows_B.empty()) rc = cmfrec_hip_session_set_zero_rows(s, 'B', zero_rows_B.data(), (int)zero_rows_B.size());
    if (!rc) rc = cmfrec_hip_session_set_sideinfo(s, U ? Uc.data() : nullptr, II ? Ic.data() : nullptr);
    if (!rc && spU) rc = cmfrec_hip_session_set_sideinfo_sparse(s, 'U', U_row, U_col, U_sp, nnz_U);
    if (!rc && spI) rc = cmfrec_hip_session_set_sideinfo_sparse(s, 'I', I_row, I_col, I_sp, nnz_I);
    // dense side information with NaN: the per-attribute rules of the dense C / D update (DenseNanSide::rules)
    for (int side = 0; side < 2 && !rc; side++) {
        const DenseNanSide &ns = side ? nanI : nanU;
        if (ns.na_col.empty() || !(side ? spI : spU)) continue;
        std::vector<unsigned char> mask; std::vector<real_t> mult;
        ns.rules((side ? k_item : k_user) + k, (scale_lam || scale_lam_sideinfo), mask, mult);
        if (use_cg) rc = cmfrec_hip_session_set_closed_form_rows(s, side ? 'D' : 'C', mask.data());
        if (!rc && !mult.empty()) rc = cmfrec_hip_session_set_lambda_multipliers(s, side ? 'D' : 'C', mult.data());
    }
    if (!rc && (nonneg || nonneg_C || nonneg_D)) rc = cmfrec_hip_session_set_nonneg(s, nonneg, nonneg_C, nonneg_D, (int)max_cd_steps);
    if (!rc && l1_lam != 0) rc = cmfrec_hip_session_set_l1(s, l1_lam, (int)max_cd_steps);
    if (!rc && (lam_unique || l1_lam_unique || scale_bias_const))
        rc = cmfrec_hip_session_set_lam_unique(s, (lam_unique || scale_bias_const) ? lam6 : nullptr,
                                               (l1_lam_unique || (scale_bias_const && l1_lam != 0)) ? l16 : nullptr, (int)max_cd_steps);
    if (!rc && scale_bias_const) rc = cmfrec_hip_session_set_scale_bias_const(s, 1);
    if (!rc) rc = cmfrec_hip_session_set_factors(s, A, B, reset_values ? nullptr : biasA, reset_values ? nullptr : biasB, C, D);
    // Ai / Bi need no start values: their first update is a closed-form solve (collective.c:8236-8240)
    if (!rc && add_implicit_features) rc = cmfrec_hip_session_set_implicit_features(s, w_implicit, nullptr, nullptr);
    if (tm.on) cmfrec_hip_session_sync(s);
    tm.lap("side info + factors upload");
    if (!rc && has_bias && reset_values && NA_as_zero_X) {
        // bias start values, missing-as-zero branches (common.c:4207-4237 one-sided; :4453-4476, :4693-4710, :4849-4868
        // two-sided): O(nnz) running means per row and column, computed here on the host and handed over
        real_t lam_u = lam6[0], lam_i = lam6[1];
        if (std::fabs(lam_u) < EPS_T) lam_u = EPS_T;
        if (std::fabs(lam_i) < EPS_T) lam_i = EPS_T;
        auto scaled_means = [&](const int_t *ix, int_t rows, int_t other) {       // running mean of the row's entries (COO order) x cnt / other
            std::vector<double> mean((size_t)rows, 0.); std::vector<size_t> cnt((size_t)rows, 0);
            for (size_t e = 0; e < nnz; e++) { const size_t r = (size_t)ix[e]; mean[r] += ((double)X[e] - mean[r]) / (double)(++cnt[r]); }
            std::vector<double> raw = mean;
            for (int_t r = 0; r < rows; r++) mean[r] *= (double)cnt[r] / (double)other;
            return std::make_tuple(mean, raw, cnt);
        };
        std::vector<real_t> bA((size_t)m_max, 0), bB((size_t)n_max, 0);
        if (user_bias && item_bias) {
            auto [meanA, rawA, cntA] = scaled_means(ixA, m, n);
            auto [meanB, rawB, cntB] = scaled_means(ixB, n, m);
            (void)rawA; (void)rawB; (void)cntA; (void)cntB;
            const double fB = (double)m / ((double)m + (double)lam_i * (scale_lam ? (double)m : 1.));
            const double fA = (double)n / ((double)n + (double)lam_u * (scale_lam ? (double)n : 1.));
            for (int iter = 0; iter < 5; iter++) {
                double bmean = 0;
                // (the reference averages biasA over `row < n` here, common.c:4697-4698; past m it reads beyond the array)
                if (iter > 0) for (int_t r = 0; r < std::min(m, n); r++) bmean += ((double)bA[r] - bmean) / (double)(r + 1);
                for (int_t c = 0; c < n; c++) bB[c] = (real_t)((meanB[c] - bmean - (double)gm) * fB);
                bmean = 0;
                if (iter > 0) for (int_t c = 0; c < n; c++) bmean += ((double)bB[c] - bmean) / (double)(c + 1);
                for (int_t r = 0; r < m; r++) bA[r] = (real_t)((meanA[r] - bmean - (double)gm) * fA);
            }
        } else if (user_bias || use_cg) {                                  // collective.c:8166-8204 (item bias alone: only with use_cg)
            const bool users = user_bias;
            const int_t rows = users ? m : n, other = users ? n : m;
            auto [mean, raw, cnt] = scaled_means(users ? ixA : ixB, rows, other);
            (void)mean;
            const double lam_b = users ? (double)lam_u : (double)lam_i;
            const double den = (double)other + lam_b * (scale_lam ? (double)other : 1.);
            std::vector<real_t> &b = users ? bA : bB;
            for (int_t r = 0; r < rows; r++) {
                if (cnt[r] > 0) {
                    double bm = raw[r];
                    bm -= (double)gm / ((double)cnt[r] / (double)other);
                    bm *= (double)cnt[r] / den;
                    b[r] = (real_t)bm;
                } else b[r] = (real_t)(-(double)gm / ((double)other / den));
            }
        }
        if (user_bias || item_bias) {
            memcpy(biasA, bA.data(), (size_t)m_max * sizeof(real_t)); memcpy(biasB, bB.data(), (size_t)n_max * sizeof(real_t));
            rc = cmfrec_hip_session_set_factors(s, nullptr, nullptr, user_bias ? biasA : nullptr, item_bias ? biasB : nullptr, nullptr, nullptr);
        }
    } else
    if (!rc && has_bias && reset_values) {                                // common.c:4410-4909; lambdas clipped like :4449-4452
        real_t lam_u = lam6[0], lam_i = lam6[1];                          // collective.c:8178, :8197, :8218-8219
        if (std::fabs(lam_u) < EPS_T) lam_u = EPS_T;
        if (std::fabs(lam_i) < EPS_T) lam_i = EPS_T;
        rc = cmfrec_hip_session_init_biases(s, lam_u, lam_i);
    }
    if (tm.on) cmfrec_hip_session_sync(s);
    tm.lap("bias init");
    // dense X under scale_lam: the rows' lambda multipliers (n for a row that misses few entries, its present entries otherwise)
    if (!rc && !dense_mult_A.empty()) rc = cmfrec_hip_session_set_lambda_multipliers(s, 'A', dense_mult_A.data());
    if (!rc && !dense_mult_B.empty()) rc = cmfrec_hip_session_set_lambda_multipliers(s, 'B', dense_mult_B.data());
    if (verbose && !rc) { printf("Starting ALS optimization routine\n\n"); fflush(stdout); }
    int rc_loop = rc ? rc : run_loop(s, mdl, niter, finalize_chol, verbose, add_implicit_features, dense_chol_A, dense_chol_B);
    if (tm.on) cmfrec_hip_session_sync(s);
    tm.lap("ALS iterations");
    if (rc_loop == 0 || rc_loop == 3) {
        int rc2 = cmfrec_hip_session_get_factors(s, A, B, biasA, biasB, C, D);
        if (!rc2 && add_implicit_features) rc2 = cmfrec_hip_session_get_implicit_features(s, Ai, Bi);
        if (rc2) rc_loop = rc2;
    }
    if (rc_loop == 0 || rc_loop == 3) {                                   // no bias beyond the shape of X (collective.c:8296, :8923-8925)
        if (user_bias) for (int_t r = m; r < m_max; r++) biasA[r] = 0;
        if (item_bias) for (int_t c = n; c < n_max; c++) biasB[c] = 0;
        if (niter > 0) zero_empty_dense();
    }
    if ((rc_loop == 0 || rc_loop == 3) && precompute_for_predictions) {   // collective.c:8936-9249
        if (verbose) { printf("Finishing precomputed matrices..."); fflush(stdout); }
        const int last_chol = (!use_cg || (finalize_chol && niter > 0)) ? 1 : 0;
        int rc2 = cmfrec_hip_session_precompute(s, last_chol, include_all_X ? 1 : 0, precomputedBtB, precomputedTransBtBinvBt, nullptr,
                                                hadU ? precomputedBeTBeChol : nullptr, hadU ? precomputedCtCw : nullptr,
                                                hadU ? precomputedTransCtCinvCt : nullptr);
        if (rc2) rc_loop = rc2;
        if (naz_U && !rc2) fill_CtUbias(precomputedCtUbias, C, U_colmeans, p, k_user + k, w_user);   // collective.c:9244-9252, :10115-10123
        if (!rc2 && NA_as_zero_X && precomputedBtXbias != nullptr) {
            // minus the sum over the items of (their bias + the mean) x their factors, the bias column as 1 (collective.c:8938-8986)
            const int_t kp = k + k_main + (user_bias ? 1 : 0);
            std::vector<double> acc((size_t)kp, 0.);
            if (item_bias || center)
                for (int_t c = 0; c < n; c++) {
                    const double coef = -((item_bias ? (double)biasB[c] : 0.) + (double)gm);
                    const real_t *b = B + (size_t)c * k_totB + k_item;
                    for (int_t f = 0; f < k + k_main; f++) acc[(size_t)f] += coef * (double)b[f];
                    if (user_bias) acc[(size_t)(kp - 1)] += coef;
                }
            for (int_t f = 0; f < kp; f++) precomputedBtXbias[f] = (real_t)acc[(size_t)f];
        }
        if (!rc2 && user_bias && B_plus_bias) {                           // append_ones_last_col, :8908-8920
            for (int_t c = 0; c < n_max; c++) {
                memcpy(B_plus_bias + (size_t)c * (k_totB + 1), B + (size_t)c * k_totB, (size_t)k_totB * sizeof(real_t));
                B_plus_bias[(size_t)c * (k_totB + 1) + k_totB] = 1;
            }
        }
        if (verbose) printf("  done\n");
    }
    cmfrec_hip_session_destroy(s);
    tm.lap("get_factors + precompute + destroy");
    if (verbose && rc_loop == 0) printf("ALS procedure terminated successfully\n");
    return rc_loop;
}

// ---- factors of new rows (the step after the path) -----------------------------------------------------
// Same positional signatures as the reference (src/cmfrec.h:2004-2071).  Supported: sparse X (COO or CSR, missing = not
// observed), dense U without NaN, L1 penalties, non-negativity; no binary side information / weights / implicit features /
// NA_as_zero.  Anything else returns 2 with a message on stderr.  The precomputed matrices of the reference's
// signature are optional accelerators there; here the small Gramians are rebuilt on the device from B and C (BtB of the
// implicit model and TransCtCinvCt are used when given, because they decide the result: collective.c:11270-11280, :3380).
static int unsupported_multiple(const char *what)
{
    fprintf(stderr, "cmfrec_hip: factors_collective_*_multiple: %s is not supported by the HIP build\n", what);
    return 2;
}

int_t factors_collective_explicit_multiple(
    real_t *A, real_t *biasA, int_t m,
    real_t *U, int_t m_u, int_t p,
    bool NA_as_zero_U, bool NA_as_zero_X,
    bool nonneg,
    int_t U_row[], int_t U_col[], real_t *U_sp, size_t nnz_U,
    size_t U_csr_p[], int_t U_csr_i[], real_t *U_csr,
    real_t *Ub, int_t m_ubin, int_t pbin,
    real_t *C, real_t *Cb,
    real_t glob_mean, real_t *biasB,
    real_t *U_colmeans,
    real_t *X, int_t ixA[], int_t ixB[], size_t nnz,
    size_t *Xcsr_p, int_t *Xcsr_i, real_t *Xcsr,
    real_t *Xfull, int_t n,
    real_t *weight,
    real_t *B,
    real_t *Bi, bool add_implicit_features,
    int_t k, int_t k_user, int_t k_item, int_t k_main,
    real_t lam, real_t *lam_unique,
    real_t l1_lam, real_t *l1_lam_unique,
    bool scale_lam, bool scale_lam_sideinfo,
    bool scale_bias_const, real_t scaling_biasA,
    real_t w_main, real_t w_user, real_t w_implicit,
    int_t n_max, bool include_all_X,
    real_t *BtB, real_t *TransBtBinvBt, real_t *BtXbias, real_t *BeTBeChol, real_t *BiTBi,
    real_t *TransCtCinvCt, real_t *CtCw, real_t *CtUbias, real_t *B_plus_bias,
    int nthreads)
{
    (void)m_ubin; (void)pbin; (void)Cb; (void)w_implicit;
    (void)BtB; (void)TransBtBinvBt; (void)BtXbias; (void)BeTBeChol; (void)BiTBi; (void)CtCw; (void)CtUbias; (void)B_plus_bias;
    (void)nthreads;
    if (NA_as_zero_U || NA_as_zero_X) return unsupported_multiple("NA_as_zero");
    const bool spU = (U == nullptr && (nnz_U || U_csr_p));
    if (Ub) return unsupported_multiple("binary side information");
    if (Xfull) return unsupported_multiple("dense X");
    if (weight) return unsupported_multiple("observation weights");
    if (Bi || add_implicit_features) return unsupported_multiple("implicit features");
    if (U == nullptr && !spU) { m_u = 0; }
    if (std::max(m, m_u) <= 0) return 0;
    if (U) for (size_t e = 0; e < (size_t)m_u * (size_t)p; e++) if (std::isnan(U[e])) return unsupported_multiple("missing values in U");
    // factors_collective_explicit_single, collective.c:10611-10630
    real_t lam_bias = lam, l1_lam_bias = l1_lam;
    if (lam_unique) { lam_bias = lam_unique[biasA ? 0 : 2]; lam = lam_unique[2]; }
    if (l1_lam_unique) { l1_lam_bias = l1_lam_unique[biasA ? 0 : 2]; l1_lam = l1_lam_unique[2]; }
    if (!biasA) scale_bias_const = false;
    if ((scale_lam || scale_lam_sideinfo) && scale_bias_const) { lam_bias *= scaling_biasA; l1_lam_bias *= scaling_biasA; }
    if (w_main != 1) {                                                          // collective_factors_warm, :3694-3713
        w_user /= w_main; lam /= w_main; lam_bias /= w_main; l1_lam /= w_main; l1_lam_bias /= w_main;
    }
    const bool l1on = l1_lam != 0 || (biasA && l1_lam_bias != 0);
    if (l1on && nonneg) return unsupported_multiple("L1 regularisation together with non-negativity");
    // rows without side information and without a bias: the reference passes scale_lam where factors_closed_form
    // expects scale_bias_const (:3789-3799), so the last factor keeps the unscaled lam
    if (!biasA) scale_bias_const = scale_lam || scale_lam_sideinfo;
    // preprocess_vec, :6337-6388: x -= biasB[col] + glob_mean.  The mean goes here, the bias is fused into the gather.
    const size_t nz = Xcsr_p ? Xcsr_p[m] : nnz;
    const real_t *vals = Xcsr_p ? Xcsr : X;
    std::vector<real_t> shifted;
    if (glob_mean != 0 && nz) {
        shifted.assign(vals, vals + nz);
        for (size_t e = 0; e < nz; e++) shifted[e] -= glob_mean;
        vals = shifted.data();
    }
    const int_t n_rows_B = include_all_X ? std::max(n, n_max) : n;
    // (TransCtCinvCt: not with non-negativity or an L1 penalty, collective.c:3378)
    int rc = cmfrec_hip_factors_multiple_l1(A, biasA, m, m_u, (U || spU) ? p : 0, U, U_colmeans, ixA, ixB, Xcsr_p ? nullptr : vals, nnz,
                                            Xcsr_p, Xcsr_i, Xcsr_p ? vals : nullptr, B, n_rows_B, C, biasB, k, k_user, k_item,
                                            k_main, lam, lam_bias, lam, w_user, false, scale_lam, scale_lam_sideinfo,
                                            scale_bias_const, nullptr, (nonneg || l1on) ? nullptr : TransCtCinvCt, U_row, U_col,
                                            U_sp, nnz_U, U_csr_p, U_csr_i, U_csr, nonneg, l1_lam, l1_lam_bias);
    if (rc == 2) fprintf(stderr, "%s\n", cmfrec_hip_last_error());
    return rc > 3 ? 1 : rc;
}

int_t factors_collective_implicit_multiple(
    real_t *A, int_t m,
    real_t *U, int_t m_u, int_t p,
    bool NA_as_zero_U,
    bool nonneg,
    int_t U_row[], int_t U_col[], real_t *U_sp, size_t nnz_U,
    size_t U_csr_p[], int_t U_csr_i[], real_t *U_csr,
    real_t *X, int_t ixA[], int_t ixB[], size_t nnz,
    size_t *Xcsr_p, int_t *Xcsr_i, real_t *Xcsr,
    real_t *B, int_t n,
    real_t *C,
    real_t *U_colmeans,
    int_t k, int_t k_user, int_t k_item, int_t k_main,
    real_t lam, real_t l1_lam, real_t alpha, real_t w_main, real_t w_user,
    real_t w_main_multiplier,
    bool apply_log_transf,
    real_t *BeTBe, real_t *BtB, real_t *BeTBeChol, real_t *CtUbias,
    int nthreads)
{
    (void)BeTBe; (void)CtUbias; (void)nthreads;
    if (NA_as_zero_U) return unsupported_multiple("NA_as_zero");
    const bool spU = (U == nullptr && (nnz_U || U_csr_p));
    if (U == nullptr && !spU) m_u = 0;
    // L1 penalty: rows with observations keep the l1_lam of the call (collective_factors_warm_implicit rescales lam and w_user by
    // w_main only, :4000-4004).
    // With side information the reference's own result is not finite (its elastic-net sweeps diverge on the block system of
    // collective_closed_form_block_implicit: +-inf in most rows of a seeded problem, tests/golden_cases.py) -- refused.
    if (l1_lam != 0 && (U || spU))
        return unsupported_multiple("L1 regularisation together with side information (the reference's result is not finite)");
    if (l1_lam != 0 && nonneg) return unsupported_multiple("L1 regularisation together with non-negativity");
    if (std::max(m, m_u) <= 0) return 0;                                        // rows out: max(m, m_u), collective.c:11210
    if (U) for (size_t e = 0; e < (size_t)m_u * (size_t)p; e++) if (std::isnan(U[e])) return unsupported_multiple("missing values in U");
    // BtB as the reference builds it when none is passed: + the lam of the call, before the w_main rescaling
    // (collective.c:11270-11280; not built for a single row without BeTBeChol either, where the row function does the same)
    const real_t lam_x = lam;
    real_t wm = w_main * w_main_multiplier;                                     // collective_factors_warm_implicit, :4000-4004
    if (wm != 1) { lam /= wm; w_user /= wm; }
    const size_t nz = Xcsr_p ? Xcsr_p[m] : nnz;
    const real_t *vals = Xcsr_p ? Xcsr : X;
    std::vector<real_t> scaled;
    if ((apply_log_transf || alpha != 1) && nz) {
        scaled.assign(vals, vals + nz);
        if (apply_log_transf) for (size_t e = 0; e < nz; e++) scaled[e] = std::log(scaled[e]);   // :10802-10810
        if (alpha != 1) for (size_t e = 0; e < nz; e++) scaled[e] *= alpha;                      // :4006-4016
        vals = scaled.data();
    }
    // a precomputed BeTBeChol without BtB means the caller's BtB is unknown: rebuild (equal for consistent inputs)
    (void)BeTBeChol;
    int rc = cmfrec_hip_factors_multiple_l1(A, nullptr, m, m_u, (U || spU) ? p : 0, U, U_colmeans, ixA, ixB, Xcsr_p ? nullptr : vals, nnz,
                                            Xcsr_p, Xcsr_i, Xcsr_p ? vals : nullptr, B, n, C, nullptr, k, k_user, k_item, k_main,
                                            lam, lam, BtB ? lam : lam_x, w_user, true, false, false, false, BtB, nullptr,
                                            U_row, U_col, U_sp, nnz_U, U_csr_p, U_csr_i, U_csr, nonneg, l1_lam, l1_lam);
    return rc > 3 ? 1 : rc;
}

}  // extern "C"
