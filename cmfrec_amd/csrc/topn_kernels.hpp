// topn_kernels.hpp -- batched top-N scoring (the step after the path).
//
// Device-side batch counterpart of the reference's per-user topN (/root/reference/src/common.c:5127-5380):
//     score(u, i) = A_u . B_i (+ biasB[i]),  items of the user's exclusion list skipped,
//     the n_top best item ids per user in descending score (ties: lower item id first).
// The reference scores all n items of ONE user with a gemv and partially sorts them; ranking many users that way
// reads B once per user.  Here a workgroup owns a tile of TOPN_UT users whose factors sit in LDS, streams the
// item factors ONCE per tile (each thread holds one item's row in registers and scores it against the 16 users),
// and keeps a per-user sorted top list in LDS: a score enters only if it beats the user's current n_top-th
// best (then the exclusion list is binary-searched -- a rare event once the threshold has risen), candidates
// are appended to a per-user buffer and merged by a bitonic sort of (list + candidates) by one wavefront.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cmfhip {

constexpr int TOPN_UT = 16;        // users per workgroup
constexpr int TOPN_TH = 256;       // threads = items per round
constexpr int TOPN_KMAX = 64;      // factors per item held in registers
constexpr int TOPN_NMAX = 128;     // largest n_top
constexpr int TOPN_SORT = 512;     // bitonic width: TOPN_NMAX list entries + up to TOPN_TH candidates, padded

template <typename T>
struct TopnParams {
    const T *A; size_t lda; int nu;          // the users to rank for, first used column
    const T *B; size_t ldb; int n, k;        // item factors, first used column
    const T *biasB;                          // or null
    const size_t *excl_p; const int *excl_i; // per-user exclusion lists, each sorted ascending; or null
    int n_top;
    int *out_ids; T *out_scores;             // [nu, n_top]; out_scores may be null
};

// (score, id) order: higher score first, then lower id
template <typename T>
__device__ __forceinline__ bool topn_before(T sa, int ia, T sb, int ib)
{
    return (sa > sb) || (sa == sb && ia < ib);
}

__host__ __device__ inline size_t topn_lds_bytes(size_t sizeof_real)
{
    return (size_t)TOPN_UT * TOPN_KMAX * sizeof_real + (size_t)TOPN_UT * TOPN_SORT * (sizeof_real + sizeof(int)) +
           (size_t)TOPN_UT * (sizeof_real + sizeof(int));
}

template <typename T>
__global__ void __launch_bounds__(TOPN_TH)
topn_kernel(const TopnParams<T> P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char topn_smem[];
    T *As = reinterpret_cast<T *>(topn_smem);                       // [UT][KMAX]
    T *ssc = As + TOPN_UT * TOPN_KMAX;                              // [UT][SORT]: [0, NP) sorted list, [NP, NP + cnt) candidates
    T *thr = ssc + TOPN_UT * TOPN_SORT;                             // [UT] score of the current n_top-th best (-inf until full)
    int *sid = reinterpret_cast<int *>(thr + TOPN_UT);              // [UT][SORT]
    int *cnt = sid + TOPN_UT * TOPN_SORT;                           // [UT]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = P.k, ntop = P.n_top;
    int NP = 1;
    while (NP < ntop) NP <<= 1;                                     // list region, a power of two
    const T NEG = -INFINITY;
    for (int u0 = blockIdx.x * TOPN_UT; u0 < P.nu; u0 += gridDim.x * TOPN_UT) {
        const int nu_t = min(TOPN_UT, P.nu - u0);
        __syncthreads();
        for (int e = tid; e < TOPN_UT * TOPN_KMAX; e += TOPN_TH) {
            const int u = e / TOPN_KMAX, f = e % TOPN_KMAX;
            As[e] = (u < nu_t && f < k) ? P.A[(size_t)(u0 + u) * P.lda + f] : T(0);
        }
        for (int e = tid; e < TOPN_UT * TOPN_SORT; e += TOPN_TH) { ssc[e] = NEG; sid[e] = 0x7fffffff; }
        if (tid < TOPN_UT) { cnt[tid] = 0; thr[tid] = NEG; }
        __syncthreads();
        for (int c0 = 0; c0 < P.n; c0 += TOPN_TH) {
            const int item = c0 + tid;
            const bool live = item < P.n;
            T b[TOPN_KMAX];
            const T *brow = P.B + (size_t)(live ? item : 0) * P.ldb;
#pragma unroll
            for (int f = 0; f < TOPN_KMAX; f++) b[f] = (f < k) ? brow[f] : T(0);
            const T bias = (P.biasB != nullptr && live) ? P.biasB[item] : T(0);
#pragma unroll 1
            for (int u = 0; u < nu_t; u++) {
                T s = bias;
                const T *au = As + u * TOPN_KMAX;
#pragma unroll
                for (int f = 0; f < TOPN_KMAX; f++) s += au[f] * b[f];
                if (live && s >= thr[u]) {                                   // NaN never enters
                    bool skip = false;
                    if (P.excl_p != nullptr) {                               // sorted list of the user: binary search
                        const size_t end = P.excl_p[u0 + u + 1];
                        size_t lo = P.excl_p[u0 + u], hi = end;
                        while (lo < hi) {
                            const size_t mid = (lo + hi) >> 1;
                            if (P.excl_i[mid] < item) lo = mid + 1; else hi = mid;
                        }
                        skip = (lo < end) && (P.excl_i[lo] == item);
                    }
                    if (!skip) {
                        const int pos = atomicAdd(&cnt[u], 1);               // at most one candidate per thread and round
                        ssc[u * TOPN_SORT + NP + pos] = s;
                        sid[u * TOPN_SORT + NP + pos] = item;
                    }
                }
            }
            __syncthreads();
            // merge: one wavefront per user, bitonic sort of (list + this round's candidates), as narrow as they need
            for (int u = wave; u < nu_t; u += TOPN_TH / 64) {
                const int c = cnt[u];
                if (c == 0) continue;                                        // wave-uniform
                T *sc = ssc + u * TOPN_SORT; int *id = sid + u * TOPN_SORT;
                int W = NP;
                while (W < NP + c) W <<= 1;                                  // <= TOPN_SORT
                for (int size = 2; size <= W; size <<= 1) {
                    for (int stride = size >> 1; stride > 0; stride >>= 1) {
                        for (int e = lane; e < W / 2; e += 64) {
                            const int i = 2 * e - (e & (stride - 1)), j = i + stride;
                            const bool up = ((i & size) == 0) || (size == W);   // final merge: best first everywhere
                            const T si = sc[i], sj = sc[j]; const int ii = id[i], ij = id[j];
                            const bool swap = up ? topn_before(sj, ij, si, ii) : topn_before(si, ii, sj, ij);
                            if (swap) { sc[i] = sj; sc[j] = si; id[i] = ij; id[j] = ii; }
                        }
                        __builtin_amdgcn_wave_barrier();
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                }
                for (int e = ntop + lane; e < W; e += 64) { sc[e] = NEG; id[e] = 0x7fffffff; }   // keep the best n_top
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) { cnt[u] = 0; thr[u] = sc[ntop - 1]; }
            }
            __syncthreads();
        }
        for (int e = tid; e < nu_t * ntop; e += TOPN_TH) {
            const int u = e / ntop, j = e % ntop;
            const T sv = ssc[u * TOPN_SORT + j];
            P.out_ids[(size_t)(u0 + u) * ntop + j] = (sv == NEG) ? -1 : sid[u * TOPN_SORT + j];
            if (P.out_scores != nullptr) P.out_scores[(size_t)(u0 + u) * ntop + j] = sv;
        }
    }
}

// scores of ONE user over a list of candidate items (the reference's per-user topN under its own name, session.hip topn_one_user)
template <typename T>
__global__ void topn_one_user_scores_kernel(const T *__restrict__ a, int k, const T *__restrict__ B, size_t ldb, const T *__restrict__ biasB,
                                            const int *__restrict__ ids, int ncand, T *__restrict__ score)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= ncand) return;
    const int item = ids[e];
    const T *b = B + (size_t)item * ldb;
    T s = T(0);
    for (int f = 0; f < k; f++) s += a[f] * b[f];
    score[e] = s + (biasB != nullptr ? biasB[item] : T(0));
}

}  // namespace cmfhip
