// pnec_es_schemes.inl -- part of pnec_frontend.hip (inside namespace pnec_hip, after es_value_grad and the quad helpers).
//
// Eigensolver schemes 1 and 2 (include/pnec_hip.h: pnec_hip_eigensolver_scheme): the two RECOLLECTIONS of the iteration
// opengv::relative_pose::eigensolver runs [EXT: opengv is not in the reference tree; the CPU checker's
// opengv file under oracle/ writes down what is remembered, how surely, and holds the sequential forms these are checked
// against line by line].
// Both run on the four lanes of a quad over a QUEUE of problems, like es_minimise_queue (scheme 0): problem slot s has
// its 36 sums at Gtab[s] and its start at tv[s]; on return tv[s] is the minimiser, te[s] the eigenvector of the smallest
// eigenvalue of M there (unit, sign arbitrary), tits[s] the iterations, tflag[s] (optional) scheme 2's "ran into maxfev".
// Quad q starts on tlist[q]; a quad that is done takes the next entry.  One trip of the loop = one es_value_grad per lane.
//
//   scheme 1 (descent): a trip tries FOUR step lengths along -g/|g| at once, each lane evaluating value AND gradient at
//     its point -- the first iteration's doubling ladder lam0 {1, 2, 4, 8} or a halving ladder lam {1, 1/2, 1/4, 1/8} --
//     and applies the sequential rule to the four values; the winner's gradient is the next iteration's, so an
//     iteration is one trip (17..40 trips per minimisation where Newton takes 5..9).
//   scheme 2 (LM): a trip evaluates F = grad f at a point (lane 0) and at its three forward-difference probes (lanes
//     1..3): the start, then every trial point x + p -- a trial that MINPACK's test accepts has its Jacobian already.
//     nfev is counted as Eigen counts it (1 + 4 per outer iteration + 1 per trial), so maxfev = 100 cuts where it would.
// Scheme 1's decisions use IEEE sqrt and division, written as in the checker (a handful per trip); scheme 2's head is forty
// of them per trip and goes through the refined reciprocal / reciprocal square root (below).  The evaluation itself is
// es_value_grad's (eigenpair by Rayleigh-quotient iteration from the neighbouring point's), which differs from the
// checker's Jacobi sweeps in the last bits only.

// lane r's copy of x within the quad (r: the same value in the quad's four lanes)
__device__ __forceinline__ double quad_pick(double x, int r) {
  const double b0 = quad_broadcast<0>(x), b1 = quad_broadcast<1>(x), b2 = quad_broadcast<2>(x), b3 = quad_broadcast<3>(x);
  return r == 0 ? b0 : (r == 1 ? b1 : (r == 2 ? b2 : b3));
}

// ---- scheme 2's linear algebra: MINPACK lmpar on the normal equations (the checker's lmpar3, same order of operations).
// Square roots and divisions go through v_rsq_f64 / v_rcp_f64 + refinement (fast_rsqrt / fast_rcp: <= 1 ulp from the IEEE
// sequences, a fifth of their instructions -- this head runs once per trip on every quad, and with IEEE operations it was as
// long as the evaluation it sits behind): the Cholesky factor is kept as its off-diagonal entries and the INVERSE diagonal.
struct LmChol3 { double i0, l10, i1, l20, l21, i2; };
__device__ __forceinline__ bool lm_chol3(const double (&A)[9], LmChol3 &L) {
  if (!(A[0] > 0.0)) return false;
  L.i0 = fast_rsqrt(A[0]);
  L.l10 = A[3] * L.i0;
  const double d1 = A[4] - L.l10 * L.l10;
  if (!(d1 > 0.0)) return false;
  L.i1 = fast_rsqrt(d1);
  L.l20 = A[6] * L.i0;
  L.l21 = (A[7] - L.l20 * L.l10) * L.i1;
  const double d2 = A[8] - L.l20 * L.l20 - L.l21 * L.l21;
  if (!(d2 > 0.0)) return false;
  L.i2 = fast_rsqrt(d2);
  return true;
}
__device__ __forceinline__ void lm_forward(const LmChol3 &L, const double (&b)[3], double (&z)[3]) {
  z[0] = b[0] * L.i0;
  z[1] = (b[1] - L.l10 * z[0]) * L.i1;
  z[2] = (b[2] - L.l20 * z[0] - L.l21 * z[1]) * L.i2;
}
__device__ __forceinline__ void lm_backward(const LmChol3 &L, const double (&z)[3], double (&x)[3]) {
  x[2] = z[2] * L.i2;
  x[1] = (z[1] - L.l21 * x[2]) * L.i1;
  x[0] = (z[0] - L.l10 * x[1] - L.l20 * x[2]) * L.i0;
}
__device__ __forceinline__ double lm_nrm3(const double (&a)[3]) { return fast_sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
__device__ __forceinline__ double lm_par3(const double (&A)[9], const double (&b)[3], const double (&diag)[3], double delta,
                                          double par, double (&x)[3]) {
  const double dwarf = 2.2250738585072014e-308;
  LmChol3 L;
  double z[3], wa1[3], wa2[3];
  const bool full_rank = lm_chol3(A, L);
  double dxnorm, fp, parl = 0.0, paru, gnorm, temp;
  if (full_rank) {
    lm_forward(L, b, z);
    lm_backward(L, z, x);
  } else {
    x[0] = x[1] = x[2] = 0.0;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) wa2[j] = diag[j] * x[j];
  dxnorm = lm_nrm3(wa2);
  fp = dxnorm - delta;
  if (full_rank && fp <= 0.1 * delta) return 0.0;
  const double inv_delta = fast_rcp(delta);
  if (full_rank) {
    const double inv_dx = fast_rcp(dxnorm);
#pragma unroll
    for (int j = 0; j < 3; ++j) wa1[j] = diag[j] * (wa2[j] * inv_dx);
    lm_forward(L, wa1, z);
    parl = fp * inv_delta * fast_rcp(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
  } else {
    fp = delta;
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) wa1[j] = b[j] * fast_rcp(diag[j]);
  gnorm = lm_nrm3(wa1);
  paru = gnorm * inv_delta;
  if (paru == 0.0) paru = dwarf * fast_rcp(fmin(delta, 0.1));
  par = fmax(par, parl);
  par = fmin(par, paru);
  if (par == 0.0) par = gnorm * fast_rcp(dxnorm);
  for (int iter = 1;; ++iter) {
    if (par == 0.0) par = fmax(dwarf, 0.001 * paru);
    double Ap[9];
    LmChol3 Lp;
#pragma unroll
    for (int i = 0; i < 9; ++i) Ap[i] = A[i];
#pragma unroll
    for (int j = 0; j < 3; ++j) Ap[4 * j] += par * diag[j] * diag[j];
    if (!lm_chol3(Ap, Lp)) {
      x[0] = x[1] = x[2] = 0.0;
      return par;
    }
    lm_forward(Lp, b, z);
    lm_backward(Lp, z, x);
#pragma unroll
    for (int j = 0; j < 3; ++j) wa2[j] = diag[j] * x[j];
    dxnorm = lm_nrm3(wa2);
    temp = fp;
    fp = dxnorm - delta;
    if (fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;
    const double inv_dx = fast_rcp(dxnorm);
#pragma unroll
    for (int j = 0; j < 3; ++j) wa1[j] = diag[j] * (wa2[j] * inv_dx);
    lm_forward(Lp, wa1, z);
    const double parc = fp * inv_delta * fast_rcp(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]);
    if (fp > 0.0) parl = fmax(parl, par);
    if (fp < 0.0) paru = fmin(paru, par);
    par = fmax(parl, par + parc);
  }
  return par;
}

// WK: which work counter the quad-evaluations belong to (a -DPNEC_WORK_COUNT build only; tools/count_chain_work.py)
template <int SCHEME, int WK = kWkRansacEvals>
__device__ __forceinline__ int es_minimise_queue_alt(int n_tasks, const int *tlist, const double (*Gtab)[36], double (*tv)[3],
                                                     double (*te)[3], int *tits, int *tflag) {
  static_assert(SCHEME == 1 || SCHEME == 2, "scheme 0 is es_minimise_queue / es_minimise_quad");
  const int lane = (int)threadIdx.x, quad = lane >> 2, role_of_lane = lane & 3;
  int slot = -1, trips = 0;
  int next = kHypPerRound;  // wave-uniform: the next entry of tlist to hand out
  bool done = true;
  double eb[3] = {0.0, 0.0, 1.0};  // eigenvector of the smallest eigenvalue at the current point
  // ---- scheme 1's state (per quad, the same in its four lanes)
  enum : int { kStart = 0, kLadder0, kLadder };
  [[maybe_unused]] int d_state = kStart, d_it = 0;
  [[maybe_unused]] double dv[3] = {0.0, 0.0, 0.0}, dg[3] = {0.0, 0.0, 0.0}, d_ev = 0.0, d_lam = 0.01, d_base = 0.01;
  // ---- scheme 2's state
  enum : int { kInit = 0, kTrial };
  [[maybe_unused]] int l_state = kInit, l_iter = 1, l_nfev = 0, l_info = 0;
  [[maybe_unused]] double lx[3] = {0.0, 0.0, 0.0}, lf[3] = {0.0, 0.0, 0.0}, lJ[9], ldiag[3] = {1.0, 1.0, 1.0}, lp[3] = {0.0, 0.0, 0.0};
  [[maybe_unused]] double l_par = 0.0, l_delta = 0.0, l_xnorm = 0.0, l_fnorm = 0.0, l_gnorm = 0.0, l_pnorm = 0.0;
#pragma unroll
  for (int i = 0; i < 9; ++i) lJ[i] = 0.0;
  auto arm = [&](int s) {
    slot = s;
    done = false;
    eb[0] = 0.0; eb[1] = 0.0; eb[2] = 1.0;
    if constexpr (SCHEME == 1) {
      dv[0] = tv[s][0]; dv[1] = tv[s][1]; dv[2] = tv[s][2];
      d_state = kStart; d_it = 0; d_lam = 0.01; d_base = 0.01; d_ev = 0.0;
    } else {
      lx[0] = tv[s][0]; lx[1] = tv[s][1]; lx[2] = tv[s][2];
      l_state = kInit; l_iter = 1; l_nfev = 0; l_info = 0; l_par = 0.0;
      ldiag[0] = ldiag[1] = ldiag[2] = 1.0;
    }
  };
  if (quad < n_tasks) arm(tlist[quad]);
  const double kSqrtEps = 1.4901161193847656e-08, kEps = 2.220446049250313e-16;
  [[maybe_unused]] int my_evals = 0;
  for (;;) {
    ++trips;
#ifdef PNEC_WORK_COUNT
    if (!done) ++my_evals;
#endif
    if (!done) {
      const double *G = Gtab[slot];
      int role = role_of_lane;
      asm volatile("" : "+v"(role));  // (see es_minimise_queue: keeps what depends on the role inside the loop)
      if constexpr (SCHEME == 1) {
        // ---- the point of this lane
        double p[3] = {dv[0], dv[1], dv[2]}, lam_mine = 0.0, dd[3] = {0.0, 0.0, 0.0};
        if (d_state != kStart) {
          const double nrm = sqrt(dg[0] * dg[0] + dg[1] * dg[1] + dg[2] * dg[2]);
          dd[0] = dg[0] / nrm; dd[1] = dg[1] / nrm; dd[2] = dg[2] / nrm;
          const double up = role == 0 ? 1.0 : (role == 1 ? 2.0 : (role == 2 ? 4.0 : 8.0));
          const double dn = role == 0 ? 1.0 : (role == 1 ? 0.5 : (role == 2 ? 0.25 : 0.125));
          lam_mine = d_base * (d_state == kLadder0 ? up : dn);
#pragma unroll
          for (int k = 0; k < 3; ++k) p[k] = dv[k] - lam_mine * dd[k];
        }
        double gp[3], ep[3] = {eb[0], eb[1], eb[2]};
        const double fp = es_value_grad<1, false>(G, p, gp, nullptr, ep, d_state != kStart);
        const double s0 = quad_broadcast<0>(fp), s1 = quad_broadcast<1>(fp), s2 = quad_broadcast<2>(fp), s3 = quad_broadcast<3>(fp);
        int acc = -1;          // the lane whose point the iteration moves to (-1: none in this trip)
        double lam_acc = 0.0;
        if (d_state == kStart) {
          acc = 0;
        } else if (d_state == kLadder0) {
          // first iteration: lam doubled while the value keeps falling (up to 0.08), then halved while it is worse
          if (s0 < d_ev) {
            double ev_run = d_ev, sev = s0, l = d_base;
            int kk = 0;
            for (;;) {
              if (!(sev < ev_run)) break;
              ev_run = sev;
              if (l * 2.0 > 0.08 || kk == 3) break;
              l *= 2.0;
              ++kk;
              sev = kk == 1 ? s1 : (kk == 2 ? s2 : s3);
            }
            // (a step back after a doubling lands on the previous length, whose value IS ev_run: the halving loop ends there)
            if (sev > ev_run && l > 1e-12) { l *= 0.5; --kk; }
            acc = kk;
            lam_acc = l;
          } else if (s0 > d_ev && d_base > 1e-12) {
            d_state = kLadder;   // halve on: lam0 / 2, / 4, / 8, / 16 in the next trip
            d_base *= 0.5;
          } else {
            acc = 0;
            lam_acc = d_base;
          }
        } else {
          // halving ladder: the first length whose value is not worse (or that is down at 1e-12)
          const double l0 = d_base, l1 = d_base * 0.5, l2 = d_base * 0.25, l3 = d_base * 0.125;
          if (!(s0 > d_ev && l0 > 1e-12)) { acc = 0; lam_acc = l0; }
          else if (!(s1 > d_ev && l1 > 1e-12)) { acc = 1; lam_acc = l1; }
          else if (!(s2 > d_ev && l2 > 1e-12)) { acc = 2; lam_acc = l2; }
          else if (!(s3 > d_ev && l3 > 1e-12)) { acc = 3; lam_acc = l3; }
          else d_base *= 0.0625;
        }
        if (acc >= 0) {
          // move to lane acc's point: its value, gradient and eigenvector are the new iteration's
          d_ev = quad_pick(fp, acc);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            dv[k] = quad_pick(p[k], acc);
            dg[k] = quad_pick(gp[k], acc);
            eb[k] = quad_pick(ep[k], acc);
          }
          if (d_state == kStart) {
            d_state = kLadder0;
            d_base = d_lam;
          } else {
            d_lam = lam_acc;
            ++d_it;
            if (d_lam < 1e-5 || d_it >= 50) done = true;
            d_state = kLadder;
            d_base = d_lam;
          }
          if (!done) {  // the next iteration's own exit: a vanishing gradient
            const double nrm = sqrt(dg[0] * dg[0] + dg[1] * dg[1] + dg[2] * dg[2]);
            if (!(nrm > 0.0)) done = true;
          }
        }
      } else {
        // ---- scheme 2: the point (lane 0) and its forward-difference probes (lanes 1..3)
        double xc[3] = {lx[0], lx[1], lx[2]};
        if (l_state == kTrial) {
#pragma unroll
          for (int k = 0; k < 3; ++k) xc[k] = lx[k] + lp[k];
        }
        double h[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          h[k] = kSqrtEps * fabs(xc[k]);
          if (h[k] == 0.0) h[k] = kSqrtEps;
        }
        double p[3] = {xc[0] + (role == 1 ? h[0] : 0.0), xc[1] + (role == 2 ? h[1] : 0.0), xc[2] + (role == 3 ? h[2] : 0.0)};
        double gp[3], ep[3] = {eb[0], eb[1], eb[2]};
        (void)es_value_grad<1, true>(G, p, gp, nullptr, ep, l_state != kInit);
        double f1[3], Jn[9], e0[3];
        const double ih[3] = {fast_rcp(h[0]), fast_rcp(h[1]), fast_rcp(h[2])};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          f1[r] = quad_broadcast<0>(gp[r]);
          Jn[3 * r + 0] = (quad_broadcast<1>(gp[r]) - f1[r]) * ih[0];
          Jn[3 * r + 1] = (quad_broadcast<2>(gp[r]) - f1[r]) * ih[1];
          Jn[3 * r + 2] = (quad_broadcast<3>(gp[r]) - f1[r]) * ih[2];
          e0[r] = quad_broadcast<0>(ep[r]);
        }
        bool new_outer = false;  // a Jacobian has just become the current one: run the head of an outer iteration
        bool plan = false;       // make the next trial step from the current Jacobian
        if (l_state == kInit) {
#pragma unroll
          for (int r = 0; r < 3; ++r) { lf[r] = f1[r]; eb[r] = e0[r]; }
#pragma unroll
          for (int i = 0; i < 9; ++i) lJ[i] = Jn[i];
          l_fnorm = lm_nrm3(lf);
          l_nfev = 1;
          new_outer = true;
        } else {
          ++l_nfev;
          const double fnorm1 = lm_nrm3(f1);
          const double inv_fn = fast_rcp(l_fnorm);
          double actred = -1.0;
          if (0.1 * fnorm1 < l_fnorm) actred = 1.0 - (fnorm1 * inv_fn) * (fnorm1 * inv_fn);
          const double Jp[3] = {lJ[0] * lp[0] + lJ[1] * lp[1] + lJ[2] * lp[2], lJ[3] * lp[0] + lJ[4] * lp[1] + lJ[5] * lp[2],
                                lJ[6] * lp[0] + lJ[7] * lp[1] + lJ[8] * lp[2]};
          const double t1 = lm_nrm3(Jp) * inv_fn, t2 = fast_sqrt(l_par) * l_pnorm * inv_fn;
          const double temp1 = t1 * t1, temp2 = t2 * t2;
          const double prered = temp1 + temp2 / 0.5, dirder = -(temp1 + temp2);
          const double ratio = (prered != 0.0) ? actred * fast_rcp(prered) : 0.0;
          if (ratio <= 0.25) {
            double temp = 0.5;
            if (actred < 0.0) temp = 0.5 * dirder * fast_rcp(dirder + 0.5 * actred);
            if (0.1 * fnorm1 >= l_fnorm || temp < 0.1) temp = 0.1;
            l_delta = temp * fmin(l_delta, l_pnorm * 10.0);
            l_par *= fast_rcp(temp);
          } else if (!(l_par != 0.0 && ratio < 0.75)) {
            l_delta = l_pnorm * 2.0;
            l_par = 0.5 * l_par;
          }
          const bool accepted = ratio >= 1e-4;
          if (accepted) {
#pragma unroll
            for (int r = 0; r < 3; ++r) { lx[r] = xc[r]; lf[r] = f1[r]; eb[r] = e0[r]; }
            const double dx[3] = {ldiag[0] * lx[0], ldiag[1] * lx[1], ldiag[2] * lx[2]};
            l_xnorm = lm_nrm3(dx);
            l_fnorm = fnorm1;
            ++l_iter;
          }
          const bool small_red = fabs(actred) <= 0.00005 && prered <= 0.00005 && 0.5 * ratio <= 1.0;
          const double xtol = 10.0 * kEps;
          if (small_red && l_delta <= xtol * l_xnorm) l_info = 3;
          else if (small_red) l_info = 1;
          else if (l_delta <= xtol * l_xnorm) l_info = 2;
          else if (l_nfev >= 100) l_info = 5;
          else if (fabs(actred) <= kEps && prered <= kEps && 0.5 * ratio <= 1.0) l_info = 6;
          else if (l_delta <= kEps * l_xnorm) l_info = 7;
          else if (l_gnorm <= kEps) l_info = 8;
          if (l_info != 0) {
            done = true;
          } else if (accepted) {
#pragma unroll
            for (int i = 0; i < 9; ++i) lJ[i] = Jn[i];
            new_outer = true;
          } else {
            plan = true;
          }
        }
        double wa2[3] = {0.0, 0.0, 0.0};
        if (new_outer) {
          // ---- head of an outer iteration (Eigen's minimizeOneStep up to its inner loop)
          l_nfev += 4;  // NumericalDiff<..., Forward>::df: f(x) again and the three probes
#pragma unroll
          for (int j = 0; j < 3; ++j) wa2[j] = fast_sqrt(lJ[j] * lJ[j] + lJ[3 + j] * lJ[3 + j] + lJ[6 + j] * lJ[6 + j]);
          if (l_iter == 1) {
#pragma unroll
            for (int j = 0; j < 3; ++j) ldiag[j] = (wa2[j] == 0.0) ? 1.0 : wa2[j];
            const double dx[3] = {ldiag[0] * lx[0], ldiag[1] * lx[1], ldiag[2] * lx[2]};
            l_xnorm = lm_nrm3(dx);
            l_delta = 100.0 * l_xnorm;
            if (l_delta == 0.0) l_delta = 100.0;
          }
          plan = true;
        }
        if (plan) {
          double A[9], b[3];
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            b[r] = lJ[r] * lf[0] + lJ[3 + r] * lf[1] + lJ[6 + r] * lf[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) A[3 * r + c] = lJ[r] * lJ[c] + lJ[3 + r] * lJ[3 + c] + lJ[6 + r] * lJ[6 + c];
          }
          if (new_outer) {
            double gnorm = 0.0;
            if (l_fnorm != 0.0) {
              const double inv_f = fast_rcp(l_fnorm);
#pragma unroll
              for (int j = 0; j < 3; ++j)
                if (wa2[j] != 0.0) gnorm = fmax(gnorm, fabs(b[j] * inv_f * fast_rcp(wa2[j])));
            }
            l_gnorm = gnorm;
            if (gnorm <= 0.0) { l_info = 4; done = true; }
#pragma unroll
            for (int j = 0; j < 3; ++j) ldiag[j] = fmax(ldiag[j], wa2[j]);
          }
          if (!done) {
            double pl[3];
            l_par = lm_par3(A, b, ldiag, l_delta, l_par, pl);
#pragma unroll
            for (int j = 0; j < 3; ++j) lp[j] = -pl[j];
            const double dp[3] = {ldiag[0] * lp[0], ldiag[1] * lp[1], ldiag[2] * lp[2]};
            l_pnorm = lm_nrm3(dp);
            if (l_iter == 1) l_delta = fmin(l_delta, l_pnorm);
            l_state = kTrial;
          }
        }
      }
    }
    // ---- quads that have finished: park the result, take the next problem of the queue (in quad order)
    const bool fin = done && slot >= 0;
    if (fin && role_of_lane == 0) {
      if constexpr (SCHEME == 1) {
        tv[slot][0] = dv[0]; tv[slot][1] = dv[1]; tv[slot][2] = dv[2];
        tits[slot] = d_it;
        if (tflag) tflag[slot] = 0;
      } else {
        tv[slot][0] = lx[0]; tv[slot][1] = lx[1]; tv[slot][2] = lx[2];
        tits[slot] = l_iter - 1;
        if (tflag) tflag[slot] = l_info == 5 ? 1 : 0;
      }
      te[slot][0] = eb[0]; te[slot][1] = eb[1]; te[slot][2] = eb[2];
    }
    const unsigned long long fb = __builtin_amdgcn_ballot_w64(fin && role_of_lane == 0);
    if (fin) {
      const int rank = __builtin_popcountll(fb & ((1ull << (lane & ~3)) - 1ull));
      const int idx = next + rank;
      slot = -1;
      if (idx < n_tasks) arm(tlist[idx]);
    }
    next += __builtin_popcountll(fb);
    if (__builtin_amdgcn_ballot_w64(!done) == 0ull) break;
  }
#ifdef PNEC_WORK_COUNT
  if (role_of_lane == 0) PNEC_WORK_ADD(WK, my_evals);
#endif
  return trips;
}
