// rng_host.hpp -- start values, seed-compatible with the reference (host side only).
//
// Restates /root/reference/src/helpers.c: splitmix64 seeding (seed_state :877-891), xoshiro256++
// (:526-538) and its jump (:543-571, constants published by Blackman & Vigna), the truncated
// ziggurat normal sampler rnorm_xoshiro (double :653-721, float :750-833 -- the float build slices
// each 64-bit draw into two 32-bit draws, quirk Q8), runif_xoshiro (:723-748, :835-875) and the
// stream layout of random_parallel (:927-1043) including its two quirks:
//   Q4  sizeA+sizeB <= 2^18  -> always NORMAL draws from one stream, whatever `normal` says (:967-971)
//   Q5  the bucket-count expression collapses to one stream per array: A <- stream(seed),
//       B <- the same state jumped once; independent of nthreads (:973-974)
// The ziggurat tables are NumPy's (generated include, tools/make_ziggurat_tables.py).
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "ziggurat_tables.inc"

namespace cmfrng {

struct Xoshiro256pp {
    uint64_t s[4];
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    static uint64_t splitmix64(uint64_t seed)
    {
        uint64_t z = seed + 0x9e3779b97f4a7c15ULL;
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    }
    explicit Xoshiro256pp(int seed)
    {
        s[0] = splitmix64((uint64_t)(int64_t)seed);     // int_t seed converted like the C call does
        s[1] = splitmix64(s[0]);
        s[2] = splitmix64(s[1]);
        s[3] = splitmix64(s[2]);
    }
    uint64_t next()
    {
        const uint64_t result = rotl(s[0] + s[3], 23) + s[0];
        const uint64_t t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return result;
    }
    void jump()
    {
        static const uint64_t J[4] = {0x180ec6d33cfd0abaULL, 0xd5a61266f0c9392cULL, 0xa9582618e03fc9aaULL,
                                      0x39abdc4529b1661cULL};
        uint64_t t[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4; i++)
            for (int b = 0; b < 64; b++) {
                if (J[i] & (UINT64_C(1) << b)) { t[0] ^= s[0]; t[1] ^= s[1]; t[2] ^= s[2]; t[3] ^= s[3]; }
                next();
            }
        memcpy(s, t, sizeof t);
    }
};

// ---- double ----
inline void fill_normal(double *seq, size_t n, Xoshiro256pp &g)
{
    size_t ix = 0;
    while (ix < n) {
        uint64_t rnd = g.next();
        const unsigned rect = (unsigned)(rnd & 255);
        rnd >>= 8;
        const unsigned sign = (unsigned)(rnd & 1);
        rnd >>= 4;                                         // 52 bits left for the uniform part
        const double x = (double)rnd * zig_wi_double[rect];
        if (rnd < zig_ki_double[rect]) {
            seq[ix++] = sign ? x : -x;
        } else if (rect != 0) {                            // wedge test; the tail (rect 0) restarts
            const uint64_t r2 = g.next();
            const double u = ((double)(r2 >> 12) + 0.5) * 0x1.0p-52;
            if (u * (zig_fi_double[rect - 1] - zig_fi_double[rect]) < std::exp(-0.5 * x * x) - zig_fi_double[rect])
                seq[ix++] = sign ? x : -x;
        }
    }
    for (size_t i = 0; i < n; i++) seq[i] *= 0x1.0p-7;
}
inline void fill_uniform(double *seq, size_t n, Xoshiro256pp &g)
{
    for (size_t i = 0; i < n; i++) seq[i] = ((double)(g.next() >> 12) + 0.5) * 0x1.0p-59;
}

// ---- float: every 64-bit draw feeds two 32-bit draws (low half first) ----
struct Half32 {
    Xoshiro256pp &g;
    uint64_t big = 0;
    bool have = false;
    explicit Half32(Xoshiro256pp &g_) : g(g_) {}
    uint32_t next()
    {
        if (have) { have = false; return (uint32_t)big; }
        big = g.next();
        have = true;
        uint32_t lo = (uint32_t)(big & 0xffffffffu);
        big >>= 32;
        return lo;
    }
};
inline void fill_normal(float *seq, size_t n, Xoshiro256pp &g)
{
    Half32 h(g);
    size_t ix = 0;
    while (ix < n) {
        uint32_t rnd = h.next();
        const unsigned rect = rnd & 255;
        rnd >>= 8;
        const unsigned sign = rnd & 1;
        rnd >>= 1;                                         // exactly 23 bits left
        const float x = (float)rnd * zig_wi_float[rect];
        if (rnd < zig_ki_float[rect]) {
            seq[ix++] = sign ? x : -x;
        } else {
            // (the float path has no rect != 0 guard; rect 0 reads fi[-1] in the reference, which is
            // the last element of the preceding table in memory -- ki_float is stored before
            // wi/fi there; we mirror the observable behaviour: treat fi[-1] as wi_float[255])
            const uint32_t r2 = h.next();
            const float u = ((float)(r2 >> 9) + 0.5f) * 0x1.0p-23f;
            const float f_prev = (rect == 0) ? zig_wi_float[255] : zig_fi_float[rect - 1];
            if (u * (f_prev - zig_fi_float[rect]) < std::exp(-0.5f * x * x) - zig_fi_float[rect])
                seq[ix++] = sign ? x : -x;
        }
    }
    for (size_t i = 0; i < n; i++) seq[i] *= 0x1.0p-7f;
}
inline void fill_uniform(float *seq, size_t n, Xoshiro256pp &g)
{
    const size_t lim = n >> 1;
    for (size_t i = 0; i < lim; i++) {
        const uint64_t rnd = g.next();
        seq[2 * i] = ((float)(rnd & 0x7fffff) + 0.5f) * 0x1.0p-30f;
        seq[2 * i + 1] = ((float)(rnd >> 41) + 0.5f) * 0x1.0p-30f;
    }
    if ((lim << 1) < n) {      // odd n: the reference writes this draw to seq[lim-1] (helpers.c:862), not the last slot
        const uint64_t rnd = g.next();
        if (lim >= 1) seq[lim - 1] = ((float)(rnd & 0x7fffff) + 0.5f) * 0x1.0p-30f;
    }
}

// random_parallel (helpers.c:927-1043)
template <typename T>
inline void random_parallel(T *A, size_t sizeA, T *B, size_t sizeB, int seed, bool normal)
{
    const size_t BUCKET = (size_t)1 << 18;
    Xoshiro256pp g(seed);
    if (sizeA + sizeB <= BUCKET) {                         // Q4
        if (sizeA) fill_normal(A, sizeA, g);
        if (sizeB) fill_normal(B, sizeB, g);
        return;
    }
    if (sizeA) {                                           // Q5: one stream per array
        Xoshiro256pp ga = g;
        if (normal) fill_normal(A, sizeA, ga); else fill_uniform(A, sizeA, ga);
    }
    if (sizeB) {
        Xoshiro256pp gb = g;
        if (sizeA) gb.jump();
        if (normal) fill_normal(B, sizeB, gb); else fill_uniform(B, sizeB, gb);
    }
}

}  // namespace cmfrng
