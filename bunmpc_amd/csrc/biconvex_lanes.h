// Lane-level helpers shared by the centroidal ADMM kernels (biconvex_admm.hip: one knot per lane, batch throughput;
// biconvex_latency.hip: one problem per wave, knots x component groups across the lanes).  Included inside each translation
// unit's anonymous namespace.
#pragma once

#define UNROLL _Pragma("unroll")
typedef unsigned long long mask_t;      // a 64-bit lane mask in scalar registers (see `lanes` at the end of this file)

// DPP controls (LLVM SIDefines.h DppCtrl)
constexpr int DPP_QUAD_XOR1 = 0xB1;     // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;     // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141;
constexpr int DPP_ROW_MIRROR = 0x140;
constexpr int DPP_WAVE_SHL1 = 0x130;    // lane i <- lane i+1
constexpr int DPP_WAVE_SHR1 = 0x138;    // lane i <- lane i-1

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    // bound_ctrl:1 -- a lane without a source lane reads 0, so no destination pre-initialisation
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
// value held by the previous / next knot's lane (0 at the wave ends)
template <typename R> __device__ __forceinline__ R from_prev(R v) { return dpp_mov<DPP_WAVE_SHR1>(v); }
template <typename R> __device__ __forceinline__ R from_next(R v) { return dpp_mov<DPP_WAVE_SHL1>(v); }

__device__ __forceinline__ double swap16_sum(double v) {
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}
__device__ __forceinline__ double swap32_sum(double v) {
    unsigned lo = __double2loint(v), hi = __double2hiint(v);
    auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double(b[0], a[0]) + __hiloint2double(b[1], a[1]);
}

// ---- LPP = 21: THREE problems per wave for horizons of 17..21 knots (the headline shape, H = 20: with 32-lane segments 22 of the
// 64 lanes own no knot).  Segments are the lane ranges 0..20, 21..41, 42..62 (lane 63 idle); the neighbour exchanges (wave
// shifts by one lane) do not care where a segment ends, the segment SUMS do: 21 lanes are not a butterfly's power of two and
// every segment straddles two DPP rows of 16 lanes.  Each row holds lanes of at most two segments -- the one that CONTINUES
// into the next row (part P of the row; all of row 0) and the one that came from the previous row (part Q; all of row 3).  The
// two parts are summed within the row by the usual four-stage butterfly (both at once), row_bcast:15 carries P to the next
// row, and P(row k-1) + Q(row k) is the sum of segment k-1: every lane of row k holds it, so it is read at lanes 16 / 32 / 48 (the
// first lanes of rows 1 / 2 / 3: `seg_desig`).  The decisions of a FISTA step are lane masks anyway: they are taken at those three
// lanes and spread over their segments by scalar bit operations (`seg_uniform`).  Fixed order of additions; a lane's value never
// meets another segment's (selects, not multiplications by 0 / 1: a diverged problem's NaNs stay its own).
constexpr int DPP_ROW_BCAST15 = 0x142;  // lane 15 of each row -> every lane of the next row (gfx9)
__device__ __forceinline__ double dpp_bcast15(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, DPP_ROW_BCAST15, 0xe, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, DPP_ROW_BCAST15, 0xe, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ bool seg21_is_p(int lane) { const int row = lane >> 4; return row == 0 || lane / 21 == (16 * row) / 21 + 1; }
constexpr mask_t kSeg21Desig = (1ull << 16) | (1ull << 32) | (1ull << 48);
constexpr mask_t kSeg21Lanes[3] = {(1ull << 21) - 1, ((1ull << 21) - 1) << 21, ((1ull << 21) - 1) << 42};
// the sums of a and b over the 21-lane segments; valid at the designated lanes (16, 32, 48) only
__device__ __forceinline__ void seg21_sum2(double &a, double &b, bool is_p) {
    double pa = is_p ? a : 0.0, pb = is_p ? b : 0.0;
    double qa = is_p ? 0.0 : a, qb = is_p ? 0.0 : b;        // (selects: a - pa would turn a diverged problem's NaN into a NaN of the other part)
    double t0, t1, t2, t3;
    t0 = dpp_mov<DPP_QUAD_XOR1>(pa); t1 = dpp_mov<DPP_QUAD_XOR1>(qa); t2 = dpp_mov<DPP_QUAD_XOR1>(pb); t3 = dpp_mov<DPP_QUAD_XOR1>(qb);
    pa += t0; qa += t1; pb += t2; qb += t3;
    t0 = dpp_mov<DPP_QUAD_XOR2>(pa); t1 = dpp_mov<DPP_QUAD_XOR2>(qa); t2 = dpp_mov<DPP_QUAD_XOR2>(pb); t3 = dpp_mov<DPP_QUAD_XOR2>(qb);
    pa += t0; qa += t1; pb += t2; qb += t3;
    t0 = dpp_mov<DPP_ROW_HALF_MIRROR>(pa); t1 = dpp_mov<DPP_ROW_HALF_MIRROR>(qa); t2 = dpp_mov<DPP_ROW_HALF_MIRROR>(pb); t3 = dpp_mov<DPP_ROW_HALF_MIRROR>(qb);
    pa += t0; qa += t1; pb += t2; qb += t3;
    t0 = dpp_mov<DPP_ROW_MIRROR>(pa); t1 = dpp_mov<DPP_ROW_MIRROR>(qa); t2 = dpp_mov<DPP_ROW_MIRROR>(pb); t3 = dpp_mov<DPP_ROW_MIRROR>(qb);
    pa += t0; qa += t1; pb += t2; qb += t3;
    a = dpp_bcast15(pa) + qa;
    b = dpp_bcast15(pb) + qb;
}
// value of v at lane `src` (compile-time constant), wave-uniform
__device__ __forceinline__ double lane_bcast(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
// ... the sum of v over the lane's segment in EVERY lane of it (per ADMM iteration, not per FISTA iteration: three readlanes)
__device__ __forceinline__ double seg21_sum(double v, int lane) {
    double z = 0.0;
    seg21_sum2(v, z, seg21_is_p(lane));
    const double s0 = lane_bcast(v, 16), s1 = lane_bcast(v, 32), s2 = lane_bcast(v, 48);
    return lane < 21 ? s0 : (lane < 42 ? s1 : s2);
}
// a mask whose bits at the designated lanes say what each segment decided -> that decision on every lane of the segment
__device__ __forceinline__ mask_t seg21_spread(mask_t m) {
    return ((m >> 16) & 1 ? kSeg21Lanes[0] : 0ull) | ((m >> 32) & 1 ? kSeg21Lanes[1] : 0ull) | ((m >> 48) & 1 ? kSeg21Lanes[2] : 0ull);
}
// lanes whose segment sums (seg_sum2) are meaningful; a decision mask made segment-uniform
template <int LPP> __device__ __forceinline__ mask_t seg_desig() { return LPP == 21 ? kSeg21Desig : ~mask_t(0); }
template <int LPP> __device__ __forceinline__ mask_t seg_uniform(mask_t m) { return LPP == 21 ? seg21_spread(m) : m; }

// Sum over the LPP-lane segment, result in every lane of it (fixed butterfly order,
// so all lanes hold the same bits).
template <int LPP>
__device__ __forceinline__ double seg_sum(double v) {
    if (LPP == 21) return seg21_sum(v, (int)(threadIdx.x & 63));
    v += dpp_mov<DPP_QUAD_XOR1>(v);
    v += dpp_mov<DPP_QUAD_XOR2>(v);
    v += dpp_mov<DPP_ROW_HALF_MIRROR>(v);
    v += dpp_mov<DPP_ROW_MIRROR>(v);
    if (LPP >= 32) v = swap16_sum(v);
    if (LPP >= 64) v = swap32_sum(v);
    return v;
}

// Two segment sums at once, step by step: the steps of one butterfly depend on each other (and a DPP move may not follow the
// write of its source by less than two cycles), so two sums issued one after the other cost two full dependency chains; issued
// side by side each hides the other's waits.  Same order of additions per sum as seg_sum, hence the same bits.
template <int LPP>
__device__ __forceinline__ void seg_sum2(double &a, double &b) {
    if (LPP == 21) { seg21_sum2(a, b, seg21_is_p((int)(threadIdx.x & 63))); return; }    // (valid at seg_desig<21>() only)
    double ta, tb;
    ta = dpp_mov<DPP_QUAD_XOR1>(a); tb = dpp_mov<DPP_QUAD_XOR1>(b); a += ta; b += tb;
    ta = dpp_mov<DPP_QUAD_XOR2>(a); tb = dpp_mov<DPP_QUAD_XOR2>(b); a += ta; b += tb;
    ta = dpp_mov<DPP_ROW_HALF_MIRROR>(a); tb = dpp_mov<DPP_ROW_HALF_MIRROR>(b); a += ta; b += tb;
    ta = dpp_mov<DPP_ROW_MIRROR>(a); tb = dpp_mov<DPP_ROW_MIRROR>(b); a += ta; b += tb;
    if (LPP >= 32) {
        unsigned al = __double2loint(a), ah = __double2hiint(a), bl = __double2loint(b), bh = __double2hiint(b);
        auto x0 = __builtin_amdgcn_permlane16_swap(al, al, false, false);
        auto x1 = __builtin_amdgcn_permlane16_swap(ah, ah, false, false);
        auto y0 = __builtin_amdgcn_permlane16_swap(bl, bl, false, false);
        auto y1 = __builtin_amdgcn_permlane16_swap(bh, bh, false, false);
        a = __hiloint2double(x1[0], x0[0]) + __hiloint2double(x1[1], x0[1]);
        b = __hiloint2double(y1[0], y0[0]) + __hiloint2double(y1[1], y0[1]);
    }
    if (LPP >= 64) {
        unsigned al = __double2loint(a), ah = __double2hiint(a), bl = __double2loint(b), bh = __double2hiint(b);
        auto x0 = __builtin_amdgcn_permlane32_swap(al, al, false, false);
        auto x1 = __builtin_amdgcn_permlane32_swap(ah, ah, false, false);
        auto y0 = __builtin_amdgcn_permlane32_swap(bl, bl, false, false);
        auto y1 = __builtin_amdgcn_permlane32_swap(bh, bh, false, false);
        a = __hiloint2double(x1[0], x0[0]) + __hiloint2double(x1[1], x0[1]);
        b = __hiloint2double(y1[0], y0[0]) + __hiloint2double(y1[1], y0[1]);
    }
}

// The two decisions of a FISTA step -- retry (cv > (L/2) g2) and exit (g2 < tol^2) -- from WAVE sums of g2 and cv taken in fp32:
// 4 DPP-fused v_add_f32 per value, one v_permlane16_swap that leaves the sums of g2 in rows 0 / 2 and of cv in rows 1 / 3, one
// v_permlane32_swap, two v_readlane: 19 instructions against the 36 of the fp64 butterfly (seg_sum2<64>).  The fp32 sums of
// non-negative terms are within 5e-7 of the exact ones; a decision is taken from them only when both comparisons are clear of
// their thresholds by more than 1e-5 (relative) and the sums are of ordinary size -- the caller falls back to the fp64
// butterfly and the reference expression otherwise (returns false), so every decision is the one fp64 arithmetic makes.
__device__ __forceinline__ bool banded_decisions(double g2, double cv, double Lh, double tol2, bool &retry, bool &finished) {
    float a = (float)g2, b = (float)cv;
    a += dpp_mov<DPP_QUAD_XOR1>(a); b += dpp_mov<DPP_QUAD_XOR1>(b);
    a += dpp_mov<DPP_QUAD_XOR2>(a); b += dpp_mov<DPP_QUAD_XOR2>(b);
    a += dpp_mov<DPP_ROW_HALF_MIRROR>(a); b += dpp_mov<DPP_ROW_HALF_MIRROR>(b);
    a += dpp_mov<DPP_ROW_MIRROR>(a); b += dpp_mov<DPP_ROW_MIRROR>(b);
    auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    const float s = __uint_as_float(x[0]) + __uint_as_float(x[1]);        // rows 0, 2: g2 over a row pair; rows 1, 3: cv
    auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
    const float w = __uint_as_float(y[0]) + __uint_as_float(y[1]);
    const float g2w = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w), 0));
    const float cvw = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w), 16));
    const double g2s = (double)g2w, cvs = (double)cvw, rhs = Lh * g2s;
    retry = cvs > rhs;
    finished = g2s < tol2;
    const bool ordinary = g2w > 1e-30f && g2w < 1e30f && cvw < 1e30f;
    return ordinary && fabs(cvs - rhs) > 1e-5 * rhs && fabs(g2s - tol2) > 1e-5 * tol2;
}

// element of a global array: wave-uniform base + 32-bit per-lane byte offset (global_load v, v_off, s[base:base+1])
__device__ __forceinline__ const double *at(const double *ubase, unsigned byte_off) {
    return reinterpret_cast<const double *>(reinterpret_cast<const char *>(ubase) + byte_off);
}
__device__ __forceinline__ double *at(double *ubase, unsigned byte_off) {
    return reinterpret_cast<double *>(reinterpret_cast<char *>(ubase) + byte_off);
}
// A zero the optimiser cannot see through, added to the offsets of a phase's input loads: the inputs of a problem do not change
// during a solve, so with plain offsets hipcc hoists the LOADED VALUES (weights, contact plan, bounds: ~60 registers) out of the
// ADMM loop and carries them across both FISTA loops; re-reading them in each of the ten ADMM iterations (L2-resident) is free.
__device__ __forceinline__ unsigned opaque_zero() { unsigned z = 0; asm volatile("" : "+v"(z)); return z; }
// v, as a value the optimiser knows nothing about (no instruction): what is computed from it stays where it is written
__device__ __forceinline__ unsigned opaque_copy(unsigned v) { asm volatile("" : "+v"(v)); return v; }
// a use of v the optimiser cannot move or remove (no instruction)
__device__ __forceinline__ void keep_here(double &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void keep_here(float &v) { asm volatile("" : "+v"(v)); }
// HBM holds fp64 whatever the arithmetic type R of the kernel; conversion happens at the load / store
// (the constant element index stays outside the 32-bit offset: it folds into the instruction's immediate)
// The load is UNCONDITIONAL (the caller's offset names an existing element in every lane); `ok` only decides what is kept.
template <typename R> __device__ __forceinline__ R ldz(const double *ubase, unsigned byte_off, int idx, bool ok) {
    const double v = at(ubase, byte_off)[idx];
    return ok ? (R)v : R(0);
}

// v where m is all ones, +0.0 where m is 0 -- two v_and_b32, no branch, and (unlike a multiply by
// 0/1) it also wipes NaN/inf, which keeps a diverged problem from leaking into its wave-mate
__device__ __forceinline__ double keep_if(double v, int m) {
    return __hiloint2double(__double2hiint(v) & m, __double2loint(v) & m);
}
__device__ __forceinline__ float keep_if(float v, int m) { return __int_as_float(__float_as_int(v) & m); }

// a / b to ~1 ulp without the IEEE division sequence: v_rcp_f64 (2^-26 or better) + two Newton
// steps + one residual correction.  Used only inside the cone branch of the projection.
__device__ __forceinline__ double fast_div(double a, double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}
__device__ __forceinline__ float fast_div(float a, float b) { return a / b; }

// type-exact fma / min / max (the unsuffixed C names would promote float operands to double)
__device__ __forceinline__ double fmaR(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fmaR(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fmaxR(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float fmaxR(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double fminR(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float fminR(float a, float b) { return __builtin_fminf(a, b); }

// The box projection max(min(v, hi), lo) as the two bare instructions.  Through fmin / fmax hipcc re-quiets the bounds inside
// the FISTA loop (`v_max_f64 x, x, x` in front of every min / max: six extra instructions per motion iteration), although
// they are loop constants quieted once outside; the instructions themselves return the non-NaN operand, as minnum / maxnum do.
__device__ __forceinline__ double clamp_box(double v, double lo, double hi) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(v), "v"(hi));
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(r), "v"(lo));
    return r;
}
__device__ __forceinline__ float clamp_box(float v, float lo, float hi) { return fmaxR(fminR(v, hi), lo); }

constexpr double kGravity = 9.81;  // centroidal.cpp:63

// Control state of the FISTA / ADMM loops (which problems of the wave still iterate, which are retrying a step, which have
// finished) is carried as 64-bit LANE MASKS in scalar registers: every decision comes out of a v_cmp as such a mask anyway,
// combining them is scalar-unit work, and `lanes(m)` turns a mask back into a per-lane predicate (selects and branches take
// the mask as it is).  As per-lane bools across loop iterations the compiler kept them as 0 / 1 in vector registers: a
// dozen vector instructions per iteration of pure bookkeeping in an issue-bound kernel.
__device__ __forceinline__ bool lanes(mask_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

