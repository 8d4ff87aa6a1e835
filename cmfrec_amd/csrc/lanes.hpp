// lanes.hpp -- cross-lane primitives for wave64 on gfx950 without touching the LDS pipe:
// DPP (quad_perm / row_shl / row_shr / row_ror with bank masks) inside 16-lane rows and
// v_permlane16_swap / v_permlane32_swap across rows.  All helpers work on float and double
// (doubles are moved as two dwords).  Verified on device by cmfrec_hip_selftest_lanes().
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace cmfhip {

// compile-time loop: f(std::integral_constant<int, I>{}) for I = I0 .. N-1
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

namespace lanes {

constexpr int QP_XOR1 = 0xB1;      // quad_perm:[1,0,3,2]
constexpr int QP_XOR2 = 0x4E;      // quad_perm:[2,3,0,1]
constexpr int ROW_SHL4 = 0x104;    // lane i <- lane i+4 (inside a 16-lane row)
constexpr int ROW_SHR4 = 0x114;    // lane i <- lane i-4
constexpr int ROW_ROR8 = 0x128;    // lane i <- lane (i+8)%16
constexpr int ROW_HALF_MIRROR = 0x141;   // lane i <- lane 7 - i inside each group of 8 lanes
constexpr int ROW_NEWBCAST = 0x150;      // + n: every lane of a 16-lane row <- lane n of the row (gfx90a+; also as v_mov_b64_dpp)

template <int CTRL, int BANK>
__device__ __forceinline__ int dpp(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, BANK, false);
}
// same without a previous value: every lane the bank mask enables is written (all sources lie inside the row), the other
// lanes are left undefined -- the caller overwrites them with a second move.  Saves the register copy update_dpp needs.
template <int CTRL, int BANK>
__device__ __forceinline__ int dpp_new(int src)
{
    return __builtin_amdgcn_mov_dpp(src, CTRL, 0xF, BANK, false);
}
// full-mask permutation inside a row (every lane has a valid source): bound_ctrl with a zero `old` is the form the compiler folds
// into the consuming 32-bit VALU instruction (v_add_f32_dpp ...) instead of a v_mov_b32_dpp of its own
template <int CTRL>
__device__ __forceinline__ int dpp_full(int src)
{
#ifndef CMF_DPP_NOFOLD
    return __builtin_amdgcn_update_dpp(0, src, CTRL, 0xF, 0xF, true);
#else
    return __builtin_amdgcn_mov_dpp(src, CTRL, 0xF, 0xF, false);
#endif
}

// ---- apply a dword -> dword lane operation to float / double / int ---------------------------
template <typename F> __device__ __forceinline__ int map(int x, F f) { return f(x); }
template <typename F> __device__ __forceinline__ float map(float x, F f) { return __int_as_float(f(__float_as_int(x))); }
template <typename F> __device__ __forceinline__ double map(double x, F f)
{
    return __hiloint2double(f(__double2hiint(x)), f(__double2loint(x)));
}
template <typename F> __device__ __forceinline__ int map2(int a, int b, F f) { return f(a, b); }
template <typename F> __device__ __forceinline__ float map2(float a, float b, F f)
{
    return __int_as_float(f(__float_as_int(a), __float_as_int(b)));
}
template <typename F> __device__ __forceinline__ double map2(double a, double b, F f)
{
    return __hiloint2double(f(__double2hiint(a), __double2hiint(b)), f(__double2loint(a), __double2loint(b)));
}

template <typename T> __device__ __forceinline__ T xor1(T x) { return map(x, [](int v) { return dpp_full<QP_XOR1>(v); }); }
template <typename T> __device__ __forceinline__ T xor2(T x) { return map(x, [](int v) { return dpp_full<QP_XOR2>(v); }); }
template <typename T> __device__ __forceinline__ T xor4(T x)
{
    return map(x, [](int v) { int t = dpp_new<ROW_SHL4, 0x5>(v); return dpp<ROW_SHR4, 0xA>(t, v); });
}
// x[lane ^ 4] for the lanes with (lane & 4) == 0 only -- one masked move per dword; the other lanes are left undefined (callers
// whose upper half-groups carry nothing they read: treduce8_low with fewer than 8 values)
template <typename T> __device__ __forceinline__ T xor4_lower(T x)
{
    return map(x, [](int v) { return dpp_new<ROW_SHL4, 0x5>(v); });
}
template <typename T> __device__ __forceinline__ T xor8(T x) { return map(x, [](int v) { return dpp_full<ROW_ROR8>(v); }); }

// lanes with (lane & M) == 0 receive a[lane ^ M], the others b[lane ^ M]  (M = 4 or 8)
template <typename T> __device__ __forceinline__ T recv_xor4(T a, T b)
{
    return map2(a, b, [](int x, int y) { int t = dpp_new<ROW_SHL4, 0x5>(x); return dpp<ROW_SHR4, 0xA>(t, y); });
}
template <typename T> __device__ __forceinline__ T recv_xor8(T a, T b)
{
    return map2(a, b, [](int x, int y) { int t = dpp_new<ROW_ROR8, 0x3>(x); return dpp<ROW_ROR8, 0xC>(t, y); });
}

// result: lanes < 32 : a[l] + a[l+32] ;  lanes >= 32 : b[l-32] + b[l]
template <typename T> __device__ __forceinline__ T tswap32_add(T a, T b);
template <> __device__ __forceinline__ float tswap32_add(float a, float b)
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <> __device__ __forceinline__ double tswap32_add(double a, double b)
{
    auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// result: lanes with (lane&16)==0 : a[l] + a[l+16] ;  others : b[l-16] + b[l]
template <typename T> __device__ __forceinline__ T tswap16_add(T a, T b);
template <> __device__ __forceinline__ float tswap16_add(float a, float b)
{
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <> __device__ __forceinline__ double tswap16_add(double a, double b)
{
    auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

// value of lane ((lane & ~7) | t) for every lane (t compile-time): two row_newbcast moves, one per half of the 16-lane row
// (bank masks 0x3 / 0xC).  A double moves as ONE v_mov_b64_dpp per half -- row_newbcast is the one DPP control the 64-bit move
// has on this part -- instead of two 32-bit moves per half.
template <int t> __device__ __forceinline__ int bcast8(int x)
{
    int y = __builtin_amdgcn_mov_dpp(x, ROW_NEWBCAST + t, 0xF, 0x3, false);
    return __builtin_amdgcn_update_dpp(y, x, ROW_NEWBCAST + 8 + t, 0xF, 0xC, false);
}
template <int t> __device__ __forceinline__ float bcast8(float x) { return __int_as_float(bcast8<t>(__float_as_int(x))); }
template <int t> __device__ __forceinline__ double bcast8(double x)
{
    double y = __builtin_amdgcn_mov_dpp(x, ROW_NEWBCAST + t, 0xF, 0x3, false);
    return __builtin_amdgcn_update_dpp(y, x, ROW_NEWBCAST + 8 + t, 0xF, 0xC, false);
}
// value of lane ((lane & ~15) | t): lane t of the lane's 16-lane row (one v_mov_b64_dpp for a double)
template <int t> __device__ __forceinline__ int row_bcast16(int x) { return __builtin_amdgcn_mov_dpp(x, ROW_NEWBCAST + t, 0xF, 0xF, false); }
template <int t> __device__ __forceinline__ float row_bcast16(float x) { return __int_as_float(row_bcast16<t>(__float_as_int(x))); }
template <int t> __device__ __forceinline__ double row_bcast16(double x) { return __builtin_amdgcn_mov_dpp(x, ROW_NEWBCAST + t, 0xF, 0xF, false); }
// value of lane (lane ^ 7) (the mirror image inside the lane's group of 8)
template <typename T> __device__ __forceinline__ T half_mirror(T x) { return map(x, [](int v) { return dpp_full<ROW_HALF_MIRROR>(v); }); }

// x[lane ^ 4] for an x whose four lanes of every quad hold the same bits (the state after the xor1 and xor2 stages of a sum:
// a + b == b + a bit for bit): the mirror lane 7 - i lies in the partner quad, so ONE full-mask move (folded into the consuming
// add in single precision) replaces the two masked moves of xor4 -- the same bits in every lane
template <typename T> __device__ __forceinline__ T qxor4(T x)
{
#ifndef CMF_DPP_NOFOLD
    return half_mirror(x);
#else
    return xor4(x);
#endif
}

// full wave sum, identical on every lane
// (tried in round 4: rows 1 / 3 adding lane 15 of the row below, rows 2 / 3 lane 31 -- the row_bcast controls -- and v_readlane 63:
//  the compiler does not fold the masked move into the add, 801 -> 815 vector instructions in the fp32 tiny kernel; not kept)
template <typename T> __device__ __forceinline__ T wave_sum(T v)
{
    v += xor1(v);
    v += xor2(v);
    v += qxor4(v);
    v += xor8(v);
    v = tswap16_add(v, v);
    v = tswap32_add(v, v);
    return v;
}

}  // namespace lanes
}  // namespace cmfhip
