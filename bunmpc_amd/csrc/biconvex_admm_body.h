// The body of the batched centroidal ADMM kernel (one knot per lane): template admm_body<R, LPP, E, RAW, HASQF>.  Included inside
// the anonymous namespace of biconvex_admm.hip (fp64 instantiations) and of biconvex_admm_f32.hip (fp32 instantiations, built
// with other compiler flags: see bunmpc_amd/build.py); the mapping, the reference lines and the algebra are described at the
// top of biconvex_admm.hip.
#pragma once

// ------------------------------------------------------------------------------
// Per-iteration algebra (what differs from the reference's formulation, all of it exact
// algebra on the same quadratic):
//   * acceptance test.  fista.cpp:16 tests  f(y+) - f(y) > g.d + (L/2)|d|^2 ; for the quadratic
//     f = y'Qy + q'y + rho|Ay - b + P|^2 that difference is  g.d + d'Qd + rho|A d|^2  identically,
//     and A d = (A y+ + bPk) - (A y + bPk) is at hand, so the test becomes
//     d'Qd + rho|A d|^2 > (L/2)|d|^2 : two segment sums instead of three, no cancellation.
//   * x_init rows.  A_f's last nine rows are the identity on X_0 (centroidal.hpp:22-27), i.e. the
//     term rho|X_0 - x_init + P_H|^2: a diagonal quadratic in X_0.  Lane 0 adds rho to its Q and
//     2 rho (P_H - x_init) to its q instead of every lane carrying nine extra residual rows.
//   * momentum coefficients (t_k - 1)/t_{k+1} (fista.cpp:34-35) depend on the iteration index
//     only: one table per device (biconvex_admm.hip: momentum_table), read by scalar loads.
//   * a problem that finishes (|d| < tol or maxit) has its iterate latched into `fin` registers
//     at that moment; the loop body itself carries no per-lane freeze selects.
//
// STEAL (round 4; LPP = 21, many ADMM iterations): the grid is what the chip holds at once and a SEGMENT whose problem has finished
// -- the ADMM's early exit (biconvex.cpp:111-114) makes the iteration counts differ per problem: 34 .. 100 at num_iters = 100 --
// stores its results and takes the next unsolved problem from a device counter, instead of idling until the slowest problem of its
// wave is done (a wave's time was the MAXIMUM over its three problems; the batch's now approaches the SUM over all problems / the
// number of segments).  The arithmetic of a problem does not depend on which segment runs it, or when.  Addresses in this mode are
// the arrays' own bases + 32-bit byte offsets from the problem index (the host checks that they fit).
//
// XLDS (round 4; the build for TWO waves per SIMD, 256 registers): the FISTA loops keep y and its image in registers only; x_k and
// ITS image rest in LDS (the problem's X / F block and an R block beside them).  An iteration reads them once -- while the segment
// sums of its step are being reduced -- for the momentum step, and leaves x_{k+1} there under the mask of the problems still
// iterating, which is also the latch of a finishing problem.  36 registers less across the loops than two register copies whose
// roles swap: both loops are free of scratch accesses at 256 registers.  Same operations in the same order: same bits.
//
// WAVES > 1 (round 4; horizons of 64 .. 255 knots: the reference's own sweep of solve times goes to 10 s horizons,
// examples/analysis/solve_times_test.py): ONE problem per WORKGROUP of WAVES wave64s, knot t on thread t.  The knot t <-> t +- 1
// exchanges cross the wave boundaries through LDS (lane 0 / lane 63 of every wave leave their values, one workgroup barrier, the
// neighbour wave's edge lane picks them up), the segment sums are the waves' sums added in wave order by every wave (same bits
// everywhere, so every decision stays workgroup-uniform and the barriers sit in uniform control flow); buffers alternate between
// two copies, so one barrier per exchange is enough.  Everything else is the one-wave code.
template <typename R, int LPP, int E, bool RAW, bool HASQF, bool STEAL = false, bool XLDS = false, int WAVES = 1>
__device__ __forceinline__ void admm_body(const BatchArgs &a) {
    constexpr bool MW = WAVES > 1;
    static_assert(!MW || (LPP == 64 && !STEAL && sizeof(R) == sizeof(double)), "several waves per problem: fp64, one problem per workgroup");
    constexpr bool PARK = XLDS && !MW;      // (the LDS header is written by lane 0 and read by the whole problem: across waves that would take barriers)
    extern __shared__ double lds_raw[];
    constexpr int NF = 3 * E;           // force variables per knot
    constexpr int NB = RAW ? 9 : 3;     // bounded components per knot
    const int lane = threadIdx.x & 63, wv = MW ? (int)(threadIdx.x >> 6) : 0;
    const int t = MW ? (int)threadIdx.x : lane % LPP;           // knot owned by this lane
    const int seg = MW ? 0 : lane / LPP;
    const int H = a.H;
    long prob = MW ? (long)blockIdx.x : (long)blockIdx.x * (64 / LPP) + seg;      // (STEAL: the segment's FIRST problem)
    // STEAL: the lane's place in its segment decides what it owns; whether the segment has a problem at all is the `alive` mask's business
    const bool pvalid = seg < 64 / LPP && (STEAL || prob < a.B);      // (LPP = 21: lane 63 belongs to no segment)
    const bool kvalid = pvalid && t <= H;  // owns knot t (X block t)
    const bool rvalid = pvalid && t < H;   // owns dynamics row-block t and force block t
    const bool l0 = pvalid && t == 0;      // also owns the x_init rows 9H..9H+8
    const long nx = 9L * (H + 1), nf = (long)NF * H;
    const mask_t rvalid_m = __ballot(rvalid), kvalid_m = __ballot(kvalid);

    const R m = (R)a.c.m, rho = (R)a.c.rho, mu = (R)a.c.mu, beta = (R)a.c.beta;
    const double tol = a.c.tol, exit_tol = a.c.exit_tol;   // exit tests are evaluated in fp64 whatever R is
    const int maxit = a.c.maxit;
    const R rho2 = R(2) * rho;

    // Iterates at phase boundaries (X, F, P of this segment's problem) live in LDS, each lane touching
    // only its own knot's blocks (lane 0 also the x_init rows of P): HBM sees the inputs once and the
    // results once.
    // Layout: kLdsZeros zeros, then per segment the x_init rows' multipliers (kSegLds elements; lane 0 works on them) and one record
    // per KNOT, [X 9 | P 9 | F NF | R 9] (kKnotLds = 39 elements: an odd stride, no two lanes of a segment share a bank).  Every block
    // of a lane -- lane 0's x_init block included -- is ONE address (the record's, less kSegLds elements) plus a constant, which the
    // LDS instructions carry as their immediate offset: one address register per lane instead of one per block (with separate
    // arrays per block the two-waves build kept reloading five of them from scratch memory).  R: the affine image of the FISTA
    // loops' x_k (XLDS).
    static_assert(9 + 9 + NF + 9 == kKnotLds && kSegLds + kKnotLds <= kLdsZeros && kSegLds >= 9, "one LDS record per knot");
    R *zeros = reinterpret_cast<R *>(lds_raw);      // what a lane without a knot reads for x_k (XLDS)
    R *Sg = zeros + kLdsZeros + (long)seg * (kSegLds + (long)(H + 1) * kKnotLds) + (long)t * kKnotLds;
    R *PIg = Sg, *Xg = Sg + kSegLds, *Pg = Xg + 9, *Fg = Xg + 18, *Rg = Xg + 18 + NF;      // (PIg: lane 0's only)
    const R *Szr = rvalid ? Sg : zeros, *Szk = kvalid ? Sg : zeros;
    const R *Fz = Szr + kSegLds + 18, *RFz = Szr + kSegLds + 18 + NF, *Xz = Szk + kSegLds, *RXz = Szk + kSegLds + 18 + NF;
    // knot t <-> t +- 1 and the sums over a problem's knots: within the wave by DPP, across a workgroup's waves (MW) through LDS
    double *const xch = reinterpret_cast<double *>(zeros) + kLdsZeros + kSegLds + (long)(H + 1) * kKnotLds;      // MW: [2][WAVES][9] next, [2][WAVES][9] previous, [2][WAVES][2] sums
    int par_n = 0, par_p = 0, par_s = 0;
    auto shift_next = [&](const auto &v, auto &o) {      // o = v of knot t + 1 (0 behind the last lane)
        constexpr int N = (int)(sizeof(v) / sizeof(v[0]));
        UNROLL for (int l = 0; l < N; ++l) o[l] = from_next(v[l]);
        if (MW) {
            double *buf = xch + par_n * (WAVES * 9);
            if (lane == 0) { UNROLL for (int l = 0; l < N; ++l) buf[wv * 9 + l] = (double)v[l]; }
            __syncthreads();
            if (lane == 63 && wv + 1 < WAVES) { UNROLL for (int l = 0; l < N; ++l) o[l] = (R)buf[(wv + 1) * 9 + l]; }
            par_n ^= 1;
        }
    };
    auto shift_prev = [&](const auto &v, auto &o) {      // o = v of knot t - 1 (0 in front of the first lane)
        constexpr int N = (int)(sizeof(v) / sizeof(v[0]));
        UNROLL for (int l = 0; l < N; ++l) o[l] = from_prev(v[l]);
        if (MW) {
            double *buf = xch + 2 * WAVES * 9 + par_p * (WAVES * 9);
            if (lane == 63) { UNROLL for (int l = 0; l < N; ++l) buf[wv * 9 + l] = (double)v[l]; }
            __syncthreads();
            if (lane == 0 && wv > 0) { UNROLL for (int l = 0; l < N; ++l) o[l] = (R)buf[(wv - 1) * 9 + l]; }
            par_p ^= 1;
        }
    };
    auto sum2 = [&](double &s0, double &s1) {      // both sums over the problem's knots (seg_sum2's contract; MW: in every lane)
        seg_sum2<LPP>(s0, s1);
        if (MW) {
            double *buf = xch + 4 * WAVES * 9 + par_s * (WAVES * 2);
            if (lane == 0) { buf[2 * wv] = s0; buf[2 * wv + 1] = s1; }
            __syncthreads();
            double t0 = buf[0], t1 = buf[1];
            UNROLL for (int w = 1; w < WAVES; ++w) { t0 += buf[2 * w]; t1 += buf[2 * w + 1]; }
            s0 = t0; s1 = t1;
            par_s ^= 1;
        }
    };
    // Global arrays are addressed as a WAVE-UNIFORM base (the block of the wave's first problem: scalar registers) plus a 32-bit
    // per-lane byte offset (problem within the wave, knot): `global_load v, v_off, s[base]`.  A 64-bit pointer per lane and array
    // -- what `a.X + pb * nx + 9 * t` makes -- held some thirty vector registers over both FISTA loops, and they were what the
    // fp32 build (256 registers, two waves per SIMD) parked in scratch memory.  Every lane's offset is that of an EXISTING
    // element (lanes past the horizon take the last knot's, lanes of a padding problem the wave's first problem): the loads
    // are unconditional -- straight-line code in which base + offset folds into the instruction, instead of some sixty
    // exec-masked blocks each needing the address as a 64-bit register pair -- and what a lane has no business with is
    // replaced by zero after the load (ldz).  Stores stay conditional.
    const long wave0 = STEAL ? 0L : (MW ? (long)blockIdx.x : (long)blockIdx.x * (64 / LPP));
    unsigned sl = STEAL ? (unsigned)(pvalid && prob < a.B ? prob : 0) : (pvalid ? (unsigned)seg : 0u);      // STEAL: the problem index itself
    const unsigned tk = (unsigned)(t <= H ? t : H), tr = (unsigned)(t < H ? t : H - 1);
    struct Off { unsigned X, PI, F, K, P9; };
    auto make_off = [&](unsigned slv) {
        Off o;
        o.X = 8u * (slv * (unsigned)nx + 9u * tk);                  // X, P, Qx, qx, lbx, ubx: [B][9 (H + 1)]
        o.PI = 8u * (slv * (unsigned)nx + 9u * (unsigned)H);
        o.F = 8u * (slv * (unsigned)nf + (unsigned)NF * tr);        // F, Qf, qf: [B][3 E H]
        o.K = 8u * (slv * (unsigned)H + tr);                        // dt: [B][H]; cnt_plan: E * 4 doubles per entry
        o.P9 = 8u * 9u * slv;                                        // x_init, X_ter: [B][9]
        return o;
    };
    unsigned oX, oPI, oF, oK, oP9;
    auto set_offsets = [&]() { const Off o = make_off(sl); oX = o.X; oPI = o.PI; oF = o.F; oK = o.K; oP9 = o.P9; };
    set_offsets();
    double *const Xu = a.X + wave0 * nx, *const Fu = a.F + wave0 * nf, *const Pu = a.P + wave0 * nx;
    const double *const xinit_u = a.x_init + wave0 * 9;

    if (lane < kLdsZeros) zeros[lane] = R(0);
    __syncthreads();
    const double *const cmtab = a.cmtab;      // (wave-uniform reads: through the scalar cache)

    R dt = ldz<R>(a.dt + wave0 * H, oK, 0, rvalid);
    R dtp;  // dt of knot t-1 (0 for t == 0: previous lane is a dead/terminal lane)
    { const R d1[1] = {dt}; R o1[1]; shift_prev(d1, o1); dtp = o1[0]; }
    const bool cold = a.cold_start != 0;      // 1: fresh solver object (iterates and step constants reset); 2: iterates only --
    const bool fresh_L = a.cold_start == 1;   // FISTA's L_ is set in the constructor and survives every optimize call (fista.hpp:52)
    const double L0x = a.L0_x, L0f = a.L0_f;      // (locals: read through `a` inside the lambda below, the two arguments got a stack copy)
    R L_x, L_f;
    int n_admm = 0, it_f = 0, it_x = 0, bt_f = 0, bt_x = 0, status = 0;
    double last_viol = 0.0;
    // XLDS: the bookkeeping of a problem -- step constants, counters, the last violation: segment-uniform values every lane carries --
    // is in registers only during the phase that changes it; across the OTHER phase's FISTA loop it rests in the problem's LDS
    // header (behind the x_init block; lane 0 writes, every lane of the segment reads the same words back).  In registers throughout
    // they were what the 256-register build stored to scratch memory in every ADMM iteration.
    static_assert(!XLDS || sizeof(R) == sizeof(double), "the LDS header holds doubles");
    // (the header's address is made where it is used, from the lane number as a value the optimiser cannot trace: hoisted out of
    // the ADMM loop it was itself kept in scratch memory)
    auto header = [&]() {
        const int sg = (int)(opaque_copy((unsigned)lane) / (unsigned)LPP);
        return reinterpret_cast<double *>(zeros) + kLdsZeros + (long)(sg < 64 / LPP ? sg : 0) * (kSegLds + (long)(H + 1) * kKnotLds);
    };
    auto lds_fence = [&]() { asm volatile("" ::: "memory"); };       // (the compiler may not carry a parked value past this in a register)
    auto park_x = [&]() {      // before the force loop: everything the motion step and the end of the ADMM iteration work on
        if (l0) { double *Hd = reinterpret_cast<double *>(Sg); int *Hi = reinterpret_cast<int *>(Hd + 12);      // (lane 0's record starts at the header)
                  Hd[9] = (double)L_x; Hd[11] = last_viol; Hi[0] = it_x; Hi[1] = bt_x; Hi[2] = n_admm; Hi[3] = status; }
        lds_fence();
    };
    auto park_f = [&]() {      // before the motion step: what the force loop works on
        if (l0) { double *Hd = reinterpret_cast<double *>(Sg); int *Hi = reinterpret_cast<int *>(Hd + 12); Hd[10] = (double)L_f; Hi[4] = it_f; Hi[5] = bt_f; }
        lds_fence();
    };
    auto load_xloop = [&]() {      // in front of the motion loop
        lds_fence();
        const double *Hd = header(); const int *Hi = reinterpret_cast<const int *>(Hd + 12);
        L_x = (R)Hd[9]; it_x = Hi[0]; bt_x = Hi[1];
    };
    auto load_rest = [&]() {       // behind it: the end of the ADMM iteration reads and updates all of it
        lds_fence();
        const double *Hd = header(); const int *Hi = reinterpret_cast<const int *>(Hd + 12);
        last_viol = Hd[11]; n_admm = Hi[2]; status = Hi[3]; L_f = (R)Hd[10]; it_f = Hi[4]; bt_f = Hi[5];
    };
    // the problem `sl` names (every offset set) comes on chip: step constants, iterates, counters (lanes of the segments in m)
    auto load_problem = [&](mask_t m) {
        const bool on = lanes(m);
        const R nLx = (R)(fresh_L ? L0x : *at(a.L_x + wave0, 8u * sl)), nLf = (R)(fresh_L ? L0f : *at(a.L_f + wave0, 8u * sl));
        if (on) { L_x = nLx; L_f = nLf; n_admm = 0; it_f = 0; it_x = 0; bt_f = 0; bt_x = 0; status = 0; last_viol = 0.0; }
        if (cold) {  // KinoDynMP::set_warm_starts (kino_dyn.cpp:83-99): X = tile(x_init), F = 0, P = 0
            if (on && kvalid) { UNROLL for (int l = 0; l < 9; ++l) Xg[l] = (R)at(xinit_u, oP9)[l]; }
            if (on && rvalid) {
                UNROLL for (int j = 0; j < NF; ++j) Fg[j] = R(0);
                UNROLL for (int l = 0; l < 9; ++l) Pg[l] = R(0);
            }
            if (on && l0) { UNROLL for (int l = 0; l < 9; ++l) PIg[l] = R(0); }
        } else {     // set_warm_start_vars: bring the caller's iterates on chip
            if (on && kvalid) { UNROLL for (int l = 0; l < 9; ++l) Xg[l] = (R)at(Xu, oX)[l]; }
            if (on && rvalid) {
                UNROLL for (int j = 0; j < NF; ++j) Fg[j] = (R)at(Fu, oF)[j];
                UNROLL for (int l = 0; l < 9; ++l) Pg[l] = (R)at(Pu, oX)[l];
            }
            if (on && l0) { UNROLL for (int l = 0; l < 9; ++l) PIg[l] = (R)at(Pu, oPI)[l]; }
        }
    };
    // ... and its results leave: one pass from LDS to the output blocks
    auto store_problem = [&](mask_t m) {
        const bool on = lanes(m);
        if (on && kvalid) { UNROLL for (int l = 0; l < 9; ++l) at(Xu, oX)[l] = (double)Xg[l]; }
        if (on && rvalid) {
            UNROLL for (int j = 0; j < NF; ++j) at(Fu, oF)[j] = (double)Fg[j];
            UNROLL for (int l = 0; l < 9; ++l) at(Pu, oX)[l] = (double)Pg[l];
        }
        if (on && l0) { UNROLL for (int l = 0; l < 9; ++l) at(Pu, oPI)[l] = (double)PIg[l]; }
        if (on && l0) {
            *at(a.L_x + wave0, 8u * sl) = (double)L_x;
            *at(a.L_f + wave0, 8u * sl) = (double)L_f;
            if (a.dyn_viol) *at(a.dyn_viol + wave0, 8u * sl) = last_viol;
            if (a.stats) {
                int *s = a.stats + (wave0 + sl) * kStats;
                s[0] = n_admm; s[1] = it_f; s[2] = it_x; s[3] = bt_f; s[4] = bt_x; s[5] = status;
            }
        }
    };
    mask_t alive = __ballot(pvalid && prob < a.B);
    if (STEAL) {
        L_x = (R)L0x; L_f = (R)L0f;      // (lanes of no problem: finite step constants, whatever they then compute is masked)
        load_problem(alive);
    } else {
        // (the plain kernels keep the straight-line prologue and epilogue they were tuned with: written through the lambdas above --
        // the same operations under a lane mask -- the headline kernel came out 2.5 % slower, 4.09 against 3.98 ms on one box)
        L_x = (R)(fresh_L ? L0x : *at(a.L_x + wave0, 8u * sl));
        L_f = (R)(fresh_L ? L0f : *at(a.L_f + wave0, 8u * sl));
        if (cold) {  // KinoDynMP::set_warm_starts (kino_dyn.cpp:83-99): X = tile(x_init), F = 0, P = 0
            if (kvalid) { UNROLL for (int l = 0; l < 9; ++l) Xg[l] = (R)at(xinit_u, oP9)[l]; }
            if (rvalid) {
                UNROLL for (int j = 0; j < NF; ++j) Fg[j] = R(0);
                UNROLL for (int l = 0; l < 9; ++l) Pg[l] = R(0);
            }
            if (l0) { UNROLL for (int l = 0; l < 9; ++l) PIg[l] = R(0); }
        } else {     // set_warm_start_vars: bring the caller's iterates on chip
            if (kvalid) { UNROLL for (int l = 0; l < 9; ++l) Xg[l] = (R)at(Xu, oX)[l]; }
            if (rvalid) {
                UNROLL for (int j = 0; j < NF; ++j) Fg[j] = (R)at(Fu, oF)[j];
                UNROLL for (int l = 0; l < 9; ++l) Pg[l] = (R)at(Pu, oX)[l];
            }
            if (l0) { UNROLL for (int l = 0; l < 9; ++l) PIg[l] = (R)at(Pu, oPI)[l]; }
        }
    }

    for (int it = 0; STEAL || it < a.c.num_iters; ++it) {
        if (alive == 0) break;
        // contact data of this knot: flags c_n, positions r_n  (centroidal.cpp:39-49); re-read in
        // each phase (L2-resident) rather than held in registers across the FISTA loops
        const double *const cnt_u = a.cnt_plan + wave0 * H * (E * 4);

        // =================================================================== F step
        {
            if (PARK) park_x();
            const unsigned ph = opaque_zero();      // see opaque_zero (biconvex_lanes.h): the inputs are re-read in each phase
            // XLDS: ... and their offsets re-made from the problem's index (hoisted out of the ADMM loop they were a dozen registers
            // that the 256-register build kept in scratch memory)
            const unsigned sl_ = XLDS ? opaque_copy(sl) : sl;
            const Off o = XLDS ? make_off(sl_) : Off{oX, oPI, oF, oK, oP9};
            const unsigned oC = o.K * (unsigned)(E * 4);
            R c[E], r[E][3];
            UNROLL for (int n = 0; n < E; ++n) {
                c[n] = ldz<R>(cnt_u, oC + ph, 4 * n, rvalid);
                UNROLL for (int k = 0; k < 3; ++k) r[n][k] = ldz<R>(cnt_u, oC + ph, 4 * n + 1 + k, rvalid);
            }
            R X[9];
            UNROLL for (int l = 0; l < 9; ++l) X[l] = kvalid ? Xg[l] : R(0);
            // bPk rows 9t+3..8 = -b_x + P, b_x = X_{t+1} - X_t (+g dt)   (centroidal.cpp:60-65)
            R bpk[6], Xv[6], Xvn[6];
            UNROLL for (int k = 0; k < 6; ++k) Xv[k] = X[3 + k];
            shift_next(Xv, Xvn);
            UNROLL for (int k = 0; k < 6; ++k) {
                const R xn = Xvn[k];
                R bx = xn - X[3 + k];
                if (k == 2) bx += R(kGravity) * dt;
                bpk[k] = rvalid ? (-bx + Pg[3 + k]) : R(0);
            }
            // A_x entries of this knot (centroidal.cpp:67-81)
            R an[E], sp[E][3];
            UNROLL for (int n = 0; n < E; ++n) {
                an[n] = c[n] * (dt / m);
                UNROLL for (int k = 0; k < 3; ++k) sp[n][k] = c[n] * (X[k] - r[n][k]) * dt;
            }
            // The gradient is carried as HALF of itself, gh = Q y + q/2 + rho A^T(A y + bPk), and the step as y - (2/L) gh:
            // scaling by two is exact in binary floating point, so every iterate has the bits of the reference's
            // y - g/L, and the doubled copies of the weights (2 Q, 2 rho) need no registers.
            R wf[NF], qf[HASQF ? NF : 1];
            UNROLL for (int j = 0; j < NF; ++j) {
                wf[j] = RAW ? ldz<R>(a.Qf + wave0 * nf, o.F + ph, j, rvalid)
                            : ldz<R>(a.W_F + wave0 * a.sW_F, 8u * (sl_ * (unsigned)a.sW_F + (unsigned)NF * tr) + ph, j, rvalid);
                if (HASQF) qf[j] = R(0.5) * ldz<R>(a.qf + wave0 * nf, o.F + ph, j, rvalid);
            }
            // u = A v + bPk on rows 9t+3..8
            auto applyA = [&](const R (&v)[NF], R (&u)[6]) {
                R s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
                UNROLL for (int n = 0; n < E; ++n) {
                    const R vx = v[3 * n], vy = v[3 * n + 1], vz = v[3 * n + 2];
                    s0 += an[n] * vx; s1 += an[n] * vy; s2 += an[n] * vz;
                    s3 += sp[n][2] * vy - sp[n][1] * vz;
                    s4 += sp[n][0] * vz - sp[n][2] * vx;
                    s5 += sp[n][1] * vx - sp[n][0] * vy;
                }
                u[0] = s0 + bpk[0]; u[1] = s1 + bpk[1]; u[2] = s2 + bpk[2];
                u[3] = s3 + bpk[3]; u[4] = s4 + bpk[4]; u[5] = s5 + bpk[5];
            };

            // FISTA state.  x lives in two buffers (xa/xb, A-images ra/rb) whose roles swap every
            // iteration, so "x_k = x_k_1" (fista.cpp:37) costs no register moves.
            R xa[NF], xb[NF], y[NF], ra[6], rb[6], ry[6];
            UNROLL for (int j = 0; j < NF; ++j) { xa[j] = rvalid ? Fg[j] : R(0); y[j] = xa[j]; }
            applyA(y, ry);
            UNROLL for (int k = 0; k < 6; ++k) ra[k] = ry[k];
            if (XLDS && rvalid) { UNROLL for (int k = 0; k < 6; ++k) Rg[k] = ry[k]; }      // (x_0 itself is in the F block already)
            const R mu2 = mu * mu, imu = R(1) / (mu * mu + R(1));
            const double tol2 = tol * tol;
            R invL = R(2) * (R(1) / L_f);      // 2 / L, see above
            mask_t act = alive;
            // one FISTA iteration: reads x from xo/ro, leaves x_{k+1} in xn/rn, advances y/ry (XLDS: x_k from LDS, x_{k+1} to LDS;
            // the four arrays are then no more than the iteration's temporaries)
            auto iterate = [&](const R (&xo_reg)[NF], const R (&ro_reg)[6], R (&xn)[NF], R (&rn)[6], int i) {
                const R cm = (R)cmtab[i];
                R xo[NF], ro[6];
                if (!XLDS) {
                    UNROLL for (int j = 0; j < NF; ++j) xo[j] = xo_reg[j];
                    UNROLL for (int k = 0; k < 6; ++k) ro[k] = ro_reg[k];
                }
                mask_t done;
                mask_t pend = act;
                for (;;) {  // backtracking (fista.cpp:8-26); segments that accepted recompute the same values
                    // g/2 = Q y + q/2 + rho A^T (A y + bPk)          (problem.cpp:36-38,54-56), the step y - (2/L) g/2 and the
                    // "SoC" projection exactly as fista.cpp:52-70 writes it (zeroing by a 0 / 1 factor: one select per foot)
                    unsigned long long anycone = 0;     // lanes with a force on the cone branch, as a scalar mask
                    R fr[NF];
                    UNROLL for (int n = 0; n < E; ++n) {
                        const R zx = an[n] * ry[0] - sp[n][2] * ry[4] + sp[n][1] * ry[5];
                        const R zy = an[n] * ry[1] + sp[n][2] * ry[3] - sp[n][0] * ry[5];
                        const R zz = an[n] * ry[2] - sp[n][1] * ry[3] + sp[n][0] * ry[4];
                        R gx = fmaR(wf[3 * n], y[3 * n], rho * zx), gy = fmaR(wf[3 * n + 1], y[3 * n + 1], rho * zy),
                          gz = fmaR(wf[3 * n + 2], y[3 * n + 2], rho * zz);
                        if (HASQF) { gx += qf[3 * n]; gy += qf[3 * n + 1]; gz += qf[3 * n + 2]; }
                        fr[3 * n] = fmaR(-gx, invL, y[3 * n]);
                        fr[3 * n + 1] = fmaR(-gy, invL, y[3 * n + 1]);
                        fr[3 * n + 2] = fmaR(-gz, invL, y[3 * n + 2]);
                        const R s = fmaR(fr[3 * n], fr[3 * n], fr[3 * n + 1] * fr[3 * n + 1]);
                        const R fz = fr[3 * n + 2];
                        const bool zero = (s * mu < -fz) || (fz < 0);
                        anycone |= __ballot(!zero && (s > mu * fz));
                        const R keep = zero ? R(0) : R(1);
                        xn[3 * n] = keep * fr[3 * n];
                        xn[3 * n + 1] = keep * fr[3 * n + 1];
                        xn[3 * n + 2] = keep * fz;
                    }
                    if (anycone != 0) {   // cone branch (fista.cpp:64-68); skipped while no lane needs it
                        UNROLL for (int n = 0; n < E; ++n) {
                            const R s = fmaR(fr[3 * n], fr[3 * n], fr[3 * n + 1] * fr[3 * n + 1]);
                            const R fz = fr[3 * n + 2];
                            const bool zero = (s * mu < -fz) || (fz < 0);
                            const bool cone = !zero && (s > mu * fz);
                            const R k = fast_div(fmaR(mu2, s, mu * fz), (mu2 + R(1)) * s);
                            xn[3 * n] = cone ? fr[3 * n] * k : xn[3 * n];
                            xn[3 * n + 1] = cone ? fr[3 * n + 1] * k : xn[3 * n + 1];
                            xn[3 * n + 2] = cone ? fmaR(mu, s, fz) * imu : xn[3 * n + 2];
                        }
                    }
                    applyA(xn, rn);
                    R g2 = 0, cv = 0, e2 = 0, dv[NF];
                    UNROLL for (int j = 0; j < NF; ++j) {
                        const R d = xn[j] - y[j];
                        dv[j] = d;
                        g2 = fmaR(d, d, g2);
                        cv = fmaR(wf[j] * d, d, cv);
                    }
                    if (sizeof(R) == sizeof(double)) {
                        UNROLL for (int k = 0; k < 6; ++k) { const R e = rn[k] - ry[k]; e2 = fmaR(e, e, e2); }
                    } else {
                        // fp32: A d = (A y+ + bPk) - (A y + bPk) by subtraction carries the rounding of the two images (1e-7 of
                        // |A y|, whatever |d| is); near convergence rho |noise|^2 then exceeds (L/2)|d|^2 and the test retries
                        // for ever (L_f x 1.5 until it overflows: seen on ~1.5 % of the trot problems).  A applied to d itself
                        // has the rounding of |A d|.
                        R s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0;
                        UNROLL for (int n = 0; n < E; ++n) {
                            const R vx = dv[3 * n], vy = dv[3 * n + 1], vz = dv[3 * n + 2];
                            s0 += an[n] * vx; s1 += an[n] * vy; s2 += an[n] * vz;
                            s3 += sp[n][2] * vy - sp[n][1] * vz;
                            s4 += sp[n][0] * vz - sp[n][2] * vx;
                            s5 += sp[n][1] * vx - sp[n][0] * vy;
                        }
                        e2 = s0 * s0 + s1 * s1 + s2 * s2 + s3 * s3 + s4 * s4 + s5 * s5;
                    }
                    cv = fmaR(rho, e2, cv);
                    if (XLDS) {     // x_k and its image come in while the sums are reduced (unconditional: lanes without a knot read zeros)
                        UNROLL for (int j = 0; j < NF; ++j) xo[j] = Fz[j];
                        UNROLL for (int k = 0; k < 6; ++k) ro[k] = RFz[k];
                    }
                    double g2s = (double)g2, cvs = (double)cv;
                    sum2(g2s, cvs);
                    // fista.cpp:14-17: G = sqrt(g2); retry if cv > (L/2) G*G; done if G < tol.  G*G and g2
                    // differ by a few ulp, so outside a 1e-14 relative band the sqrt cannot change either
                    // decision; inside it the reference expression is evaluated as written.
                    const double Lh = (double)L_f * 0.5, rhs = Lh * g2s;
                    mask_t bt = __ballot(cvs > rhs);
                    done = __ballot(g2s < tol2);
                    const mask_t edge = __ballot((fabs(cvs - rhs) <= 1e-14 * rhs) || (fabs(g2s - tol2) <= 1e-14 * tol2)) & seg_desig<LPP>();
                    if (edge != 0) {
                        const double Gn = sqrt(g2s);
                        bt = __ballot(cvs > Lh * (Gn * Gn));
                        done = __ballot(Gn < tol);
                    }
                    bt = seg_uniform<LPP>(bt);      // (LPP = 21: the sums live at three lanes; their decisions go to their segments)
                    done = seg_uniform<LPP>(done);
                    if (XLDS) {     // a use that stays in this loop: without it hipcc sinks the reads to the momentum step, where their latency shows
                        UNROLL for (int j = 0; j < NF; ++j) keep_here(xo[j]);
                        UNROLL for (int k = 0; k < 6; ++k) keep_here(ro[k]);
                    }
                    bt &= pend;
                    pend = bt;
                    if (bt == 0) break;
                    if (lanes(bt)) { L_f *= beta; ++bt_f; }
                    invL = R(2) * (R(1) / L_f);
                }
                if (!XLDS) {
                    const mask_t last = act & (i == maxit - 1 ? ~mask_t(0) : done) & rvalid_m;
                    if (lanes(last)) { UNROLL for (int j = 0; j < NF; ++j) Fg[j] = xn[j]; }   // x_k of a finishing problem is latched
                }
                // momentum (fista.cpp:33-47); A-images follow by linearity
                UNROLL for (int j = 0; j < NF; ++j) y[j] = fmaR(cm, xn[j] - xo[j], xn[j]);
                UNROLL for (int k = 0; k < 6; ++k) ry[k] = fmaR(cm, rn[k] - ro[k], rn[k]);
                if (XLDS && lanes(act & rvalid_m)) {      // problems still iterating (a finished one keeps the x_k it finished with)
                    UNROLL for (int j = 0; j < NF; ++j) Fg[j] = xn[j];
                    UNROLL for (int k = 0; k < 6; ++k) Rg[k] = rn[k];
                }
                it_f += lanes(act) ? 1 : 0;
                act &= ~done;
            };
            for (int i = 0; i < maxit; i += 2) {
                if (act == 0) break;
                iterate(xa, ra, xb, rb, i);
                if (i + 1 >= maxit || act == 0) break;
                iterate(xb, rb, xa, ra, i + 1);
            }
        }

        // =================================================================== X step
        {
            if (PARK) park_f();
            const unsigned ph = opaque_zero();
            const unsigned sl_ = XLDS ? opaque_copy(sl) : sl;
            const Off o = XLDS ? make_off(sl_) : Off{oX, oPI, oF, oK, oP9};
            const unsigned oC = o.K * (unsigned)(E * 4);
            R c[E], r[E][3];
            UNROLL for (int n = 0; n < E; ++n) {
                c[n] = ldz<R>(cnt_u, oC + ph, 4 * n, rvalid);
                UNROLL for (int k = 0; k < 3; ++k) r[n][k] = ldz<R>(cnt_u, oC + ph, 4 * n + 1 + k, rvalid);
            }
            // A_f / b_f entries of this knot from the new forces (centroidal.cpp:86-127)
            R SX = 0, SY = 0, SZ = 0, bf[9];
            auto make_bf = [&](const R (&cc)[E], const R (&rr)[E][3], R (&b)[9], R &sx, R &sy, R &sz) {
                R b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0, b8 = 0;
                sx = 0; sy = 0; sz = 0;
                UNROLL for (int n = 0; n < E; ++n) {
                    const R fx = rvalid ? Fg[3 * n] : R(0), fy = rvalid ? Fg[3 * n + 1] : R(0),
                            fz = rvalid ? Fg[3 * n + 2] : R(0);
                    sx += cc[n] * fx * dt; sy += cc[n] * fy * dt; sz += cc[n] * fz * dt;
                    b3 += -cc[n] * fx * dt / m; b4 += -cc[n] * fy * dt / m; b5 += -cc[n] * fz * dt / m;
                    b6 += (cc[n] * fy * rr[n][2] - cc[n] * fz * rr[n][1]) * dt;
                    b7 += (cc[n] * fz * rr[n][0] - cc[n] * fx * rr[n][2]) * dt;
                    b8 += (cc[n] * fx * rr[n][1] - cc[n] * fy * rr[n][0]) * dt;
                }
                b[0] = 0; b[1] = 0; b[2] = 0;
                b[3] = b3; b[4] = b4; b[5] = b5 + R(kGravity) * dt;
                b[6] = b6; b[7] = b7; b[8] = b8;
            };
            make_bf(c, r, bf, SX, SY, SZ);
            R bpk[9];
            UNROLL for (int l = 0; l < 9; ++l) bpk[l] = rvalid ? (-bf[l] + Pg[l]) : R(0);
            // cost and bounds of this knot
            R qd[9], q[9], lb[NB], ub[NB];
            if (RAW) {
                UNROLL for (int l = 0; l < 9; ++l) {
                    qd[l] = ldz<R>(a.Qx + wave0 * nx, o.X + ph, l, kvalid);
                    q[l] = R(0.5) * ldz<R>(a.qx + wave0 * nx, o.X + ph, l, kvalid);     // q/2
                }
                UNROLL for (int l = 0; l < NB; ++l) {
                    const R lo = (R)at(a.lbx + wave0 * nx, o.X + ph)[l], hi = (R)at(a.ubx + wave0 * nx, o.X + ph)[l];
                    lb[l] = kvalid ? lo : R(-INFINITY);
                    ub[l] = kvalid ? hi : R(INFINITY);
                }
            } else {
                // create_cost_X (biconvex.cpp:57-72)
                UNROLL for (int l = 0; l < 9; ++l) {
                    const double w_run = at(a.W_X + wave0 * a.sW_X, 8u * (sl_ * (unsigned)a.sW_X + 9u * tr) + ph)[l];
                    const double w_ter = at(a.W_X_ter + wave0 * a.sW_X_ter, 8u * sl_ * (unsigned)a.sW_X_ter + ph)[l];
                    const double x_run = at(a.X_nom + wave0 * 9L * H, 8u * 9u * (sl_ * (unsigned)H + tr) + ph)[l];
                    const double x_ter = at(a.X_ter + wave0 * 9, o.P9 + ph)[l];
                    const R w = (R)(rvalid ? w_run : (kvalid ? w_ter : 0.0));
                    const R xr = (R)(rvalid ? x_run : (kvalid ? x_ter : 0.0));
                    qd[l] = w;
                    q[l] = -(xr * w);              // q/2 (create_cost_X: q = -2 W x_ref)
                }
                // create_bound_constraints (biconvex.cpp:27-55): CoM box around the feet
                R csum = 0;
                UNROLL for (int n = 0; n < E; ++n) csum += c[n];
                const bool bounded = rvalid && csum > 0;
                UNROLL for (int k = 0; k < 3; ++k) {
                    R mx = r[0][k], mn = r[0][k];
                    UNROLL for (int n = 1; n < E; ++n) { mx = fmaxR(mx, r[n][k]); mn = fminR(mn, r[n][k]); }
                    const double *bnd = at(a.bounds + wave0 * a.sbounds, 8u * (sl_ * (unsigned)a.sbounds + 6u * tr) + ph);
                    const double b_lo = bnd[k], b_hi = bnd[3 + k];
                    const R blo = (R)(bounded ? b_lo : 0.0);
                    const R bhi = (R)(bounded ? b_hi : 0.0);
                    lb[k] = bounded ? mx + blo : R(-INFINITY);
                    ub[k] = bounded ? mn + bhi : R(INFINITY);
                }
            }
            // x_init rows folded into lane 0's diagonal cost:  rho |X_0 + (P_H - x_init)|^2   (q holds q/2: half-gradient form,
            // see the force step)
            R pi[9];
            UNROLL for (int l = 0; l < 9; ++l) pi[l] = R(0);
            if (l0) { UNROLL for (int l = 0; l < 9; ++l) pi[l] = PIg[l]; }
            UNROLL for (int l = 0; l < 9; ++l) {
                const R xi = (R)at(xinit_u, o.P9 + ph)[l];
                const R bpi = l0 ? (pi[l] - xi) : R(0);
                qd[l] += l0 ? rho : R(0);
                q[l] = fmaR(rho, bpi, q[l]);
            }
            UNROLL for (int l = 0; l < NB; ++l) {   // quieted once, so the clamp is a bare min/max pair
                lb[l] = __builtin_canonicalize(lb[l]);
                ub[l] = __builtin_canonicalize(ub[l]);
            }
            const int rmask = rvalid ? -1 : 0;      // row-block mask: lanes t >= H own no dynamics rows
            // u = A_f v + bPk on row-block t; vn = v of knot t+1
            auto applyA = [&](const R (&v)[9], R (&u)[9]) {
                R vn[9];
                shift_next(v, vn);
                R w[9];
                UNROLL for (int l = 0; l < 9; ++l) w[l] = v[l] - vn[l];
                UNROLL for (int k = 0; k < 3; ++k) w[k] += dt * vn[3 + k];
                w[6] += SY * v[2] - SZ * v[1];
                w[7] += SZ * v[0] - SX * v[2];
                w[8] += SX * v[1] - SY * v[0];
                UNROLL for (int l = 0; l < 9; ++l) u[l] = keep_if(w[l] + bpk[l], rmask);
            };

            R xa[9], xb[9], y[9], ra[9], rb[9], ry[9];
            UNROLL for (int l = 0; l < 9; ++l) { xa[l] = kvalid ? Xg[l] : R(0); y[l] = xa[l]; }
            applyA(y, ry);
            UNROLL for (int l = 0; l < 9; ++l) ra[l] = ry[l];
            if (XLDS && kvalid) { UNROLL for (int l = 0; l < 9; ++l) Rg[l] = ry[l]; }
            const double tol2 = tol * tol;
            if (PARK) load_xloop();
            R invL = R(2) * (R(1) / L_x);
            mask_t act = alive;
            auto iterate = [&](const R (&xo_reg)[9], const R (&ro_reg)[9], R (&xn)[9], R (&rn)[9], int i) {
                const R cm = (R)cmtab[i];
                R xo[9], ro[9];
                if (!XLDS) { UNROLL for (int l = 0; l < 9; ++l) { xo[l] = xo_reg[l]; ro[l] = ro_reg[l]; } }
                mask_t done;
                mask_t pend = act;
                for (;;) {
                    {   // half gradient Q y + q/2 + rho A_f^T (A_f y + bPk), step, box projection (fista.cpp:10); inside the retry
                        // loop like the force step's
                        R z[9], wp[9];
                        shift_prev(ry, wp);  // row-block t-1 (0 for t == 0)
                        UNROLL for (int l = 0; l < 9; ++l) z[l] = ry[l] - wp[l];
                        UNROLL for (int k = 0; k < 3; ++k) z[3 + k] = fmaR(dtp, wp[k], z[3 + k]);
                        z[0] += SZ * ry[7] - SY * ry[8];
                        z[1] += SX * ry[8] - SZ * ry[6];
                        z[2] += SY * ry[6] - SX * ry[7];
                        UNROLL for (int l = 0; l < 9; ++l) {
                            const R g = fmaR(qd[l], y[l], fmaR(rho, z[l], q[l]));
                            R v = fmaR(-g, invL, y[l]);
                            if (l < NB) v = clamp_box(v, lb[l], ub[l]);
                            xn[l] = v;
                        }
                    }
                    applyA(xn, rn);
                    R g2 = 0, cv = 0, e2 = 0;
                    UNROLL for (int l = 0; l < 9; ++l) {
                        const R d = xn[l] - y[l];
                        const R e = rn[l] - ry[l];
                        g2 = fmaR(d, d, g2);
                        cv = fmaR(qd[l] * d, d, cv);
                        e2 = fmaR(e, e, e2);
                    }
                    cv = fmaR(rho, e2, cv);
                    if (XLDS) { UNROLL for (int l = 0; l < 9; ++l) { xo[l] = Xz[l]; ro[l] = RXz[l]; } }      // (see the force step)
                    double g2s = (double)g2, cvs = (double)cv;
                    sum2(g2s, cvs);
                    const double Lh = (double)L_x * 0.5, rhs = Lh * g2s;   // see the force loop for the sqrt-free form
                    mask_t bt = __ballot(cvs > rhs);
                    done = __ballot(g2s < tol2);
                    const mask_t edge = __ballot((fabs(cvs - rhs) <= 1e-14 * rhs) || (fabs(g2s - tol2) <= 1e-14 * tol2)) & seg_desig<LPP>();
                    if (edge != 0) {
                        const double Gn = sqrt(g2s);
                        bt = __ballot(cvs > Lh * (Gn * Gn));
                        done = __ballot(Gn < tol);
                    }
                    bt = seg_uniform<LPP>(bt);      // (LPP = 21: the sums live at three lanes; their decisions go to their segments)
                    done = seg_uniform<LPP>(done);
                    if (XLDS) { UNROLL for (int l = 0; l < 9; ++l) { keep_here(xo[l]); keep_here(ro[l]); } }
                    bt &= pend;
                    pend = bt;
                    if (bt == 0) break;
                    if (lanes(bt)) { L_x *= beta; ++bt_x; }
                    invL = R(2) * (R(1) / L_x);
                }
                if (!XLDS) {
                    const mask_t last = act & (i == maxit - 1 ? ~mask_t(0) : done) & kvalid_m;
                    if (lanes(last)) { UNROLL for (int l = 0; l < 9; ++l) Xg[l] = xn[l]; }
                }
                UNROLL for (int l = 0; l < 9; ++l) {
                    y[l] = fmaR(cm, xn[l] - xo[l], xn[l]);
                    ry[l] = fmaR(cm, rn[l] - ro[l], rn[l]);
                }
                if (XLDS && lanes(act & kvalid_m)) { UNROLL for (int l = 0; l < 9; ++l) { Xg[l] = xn[l]; Rg[l] = rn[l]; } }
                it_x += lanes(act) ? 1 : 0;
                act &= ~done;
            };
            for (int i = 0; i < maxit; i += 2) {
                if (act == 0) break;
                iterate(xa, ra, xb, rb, i);
                if (i + 1 >= maxit || act == 0) break;
                iterate(xb, rb, xa, ra, i + 1);
            }
            R fin[9];
            UNROLL for (int l = 0; l < 9; ++l) fin[l] = kvalid ? Xg[l] : R(0);
            if (PARK) load_rest();
            if (XLDS) {     // b_f made again from the contact plan and the forces (same expressions, same bits) instead of six registers held
                            // across the FISTA loop -- which the 256-register build held in scratch memory
                const unsigned ph2 = opaque_zero();
                R c2[E], r2[E][3], s0, s1, s2;
                UNROLL for (int n = 0; n < E; ++n) {
                    c2[n] = ldz<R>(cnt_u, oC + ph2, 4 * n, rvalid);
                    UNROLL for (int k = 0; k < 3; ++k) r2[n][k] = ldz<R>(cnt_u, oC + ph2, 4 * n + 1 + k, rvalid);
                }
                make_bf(c2, r2, bf, s0, s1, s2);
            }

            // dyn_violation = A_f X - b_f ; P += dyn_violation          (biconvex.cpp:98-99)
            double v2 = 0;   // the dynamics violation is accumulated in fp64 whatever R is
            {
                R xn[9], w[9];
                shift_next(fin, xn);
                UNROLL for (int l = 0; l < 9; ++l) w[l] = fin[l] - xn[l];
                UNROLL for (int k = 0; k < 3; ++k) w[k] += dt * xn[3 + k];
                w[6] += SY * fin[2] - SZ * fin[1];
                w[7] += SZ * fin[0] - SX * fin[2];
                w[8] += SX * fin[1] - SY * fin[0];
                const bool al = lanes(alive);
                R dr[9], dx0[9];
                UNROLL for (int l = 0; l < 9; ++l) {
                    const R d = rvalid ? (w[l] - bf[l]) : R(0);
                    const R xi = (R)at(xinit_u, o.P9 + ph)[l];
                    const R di = l0 ? (fin[l] - xi) : R(0);
                    dr[l] = d; dx0[l] = di;
                    v2 += (double)d * (double)d + (double)di * (double)di;
                }
                if (al && rvalid) { UNROLL for (int l = 0; l < 9; ++l) Pg[l] += dr[l]; }
                if (al && l0) { UNROLL for (int l = 0; l < 9; ++l) PIg[l] += dx0[l]; }
            }
            if (MW) { double z2 = 0.0; sum2(v2, z2); } else v2 = seg_sum<LPP>(v2);
            const double nrm = sqrt(v2);
            if (lanes(alive)) {
                last_viol = nrm;
                const unsigned row = STEAL ? (unsigned)n_admm : (unsigned)it;      // the ADMM iteration this was, counted per problem
                ++n_admm;
                if (a.hist && l0) *at(a.hist + wave0 * a.c.num_iters, 8u * (sl_ * (unsigned)a.c.num_iters + row)) = nrm;
#ifndef BMPC_NO_TRACE
                if (a.trace && l0) {
                    int *tr = a.trace + ((wave0 + sl_) * a.c.num_iters + row) * 4;
                    tr[0] = it_f; tr[1] = it_x; tr[2] = bt_f; tr[3] = bt_x;
                }
#endif
                if (isnan(nrm)) status = 2;                                   // biconvex.cpp:106-109
            }
            const mask_t ex = __ballot(isnan(nrm) || nrm < exit_tol);         // biconvex.cpp:106-109, 111-114
            if (!STEAL) alive &= ~ex;
            else {
                // segments whose problem is over (exit, NaN, or all its iterations run): results out, the next problem in
                const mask_t fin = alive & (ex | __ballot(n_admm >= a.c.num_iters));
                if (fin != 0) {
                    store_problem(fin);
                    int np = -1;
                    if (lanes(fin) && l0) np = (int)((long)gridDim.x * (64 / LPP)) + atomicAdd(a.queue, 1);
                    static_assert(!STEAL || LPP == 21, "the segment broadcast below is written for three segments of 21 lanes");
                    const int n0 = __builtin_amdgcn_readlane(np, 0), n1 = __builtin_amdgcn_readlane(np, 21), n2 = __builtin_amdgcn_readlane(np, 42);
                    const int mine = seg == 0 ? n0 : (seg == 1 ? n1 : n2);
                    const mask_t got = fin & __ballot(pvalid && mine >= 0 && mine < a.B);
                    if (lanes(got)) { prob = mine; sl = (unsigned)mine; set_offsets(); }
                    const R ndt = ldz<R>(a.dt, oK, 0, rvalid);
                    if (lanes(got)) dt = ndt;
                    dtp = from_prev(dt);
                    load_problem(got);
                    alive = (alive & ~fin) | got;
                }
            }
        }
    }
    if (!STEAL) {     // ---- results: one pass from LDS to the output blocks
        if (kvalid) { UNROLL for (int l = 0; l < 9; ++l) at(Xu, oX)[l] = (double)Xg[l]; }
        if (rvalid) {
            UNROLL for (int j = 0; j < NF; ++j) at(Fu, oF)[j] = (double)Fg[j];
            UNROLL for (int l = 0; l < 9; ++l) at(Pu, oX)[l] = (double)Pg[l];
        }
        if (l0) { UNROLL for (int l = 0; l < 9; ++l) at(Pu, oPI)[l] = (double)PIg[l]; }
        if (l0) {
            *at(a.L_x + wave0, 8u * sl) = (double)L_x;
            *at(a.L_f + wave0, 8u * sl) = (double)L_f;
            if (a.dyn_viol) *at(a.dyn_viol + wave0, 8u * sl) = last_viol;
            if (a.stats) {
                int *s = a.stats + (wave0 + sl) * kStats;
                s[0] = n_admm; s[1] = it_f; s[2] = it_x; s[3] = bt_f; s[4] = bt_x; s[5] = status;
            }
        }
    }
}
