/*
 * Feasibility restoration phase -- members of msd::Solver (included inside the struct; msd_kernel.hpp).
 *
 * What IPOPT does where the filter line search breaks down (MinC_1NrmRestorationPhase, Waechter & Biegler 2006 section 3.3; reached by the
 * reference through ocp.py:290,359, surfaced at ocp.py:362-370): the same filter interior-point iteration on
 *     min  rho sum(n + p) + sqrt(mu)/2 |D_R (x - x_R)|^2    s.t.  rows(x, sigma) + n - p = 0,   bounds on x and sigma,   n, p >= 0
 * with rows = the two (scaled) dynamics rows and the rows d(x) - sigma of every interval, started at the current point x_R with
 * mu = max(mu, |rows|_inf), (n, p) from the closed-form minimiser, z = mu/(n, p), zero row multipliers and the bound multipliers cut at rho.
 * It ends as soon as an iterate reduces the infeasibility of the original problem to 90 % and is acceptable to the original filter and to
 * the point it started from; if the restoration problem itself converges first, the original problem is locally infeasible there
 * (MSD_STATUS_INFEASIBLE, IPOPT's Infeasible_Problem_Detected).  Restated step for step in oracle/ms_oracle.c: restoration().
 *
 * Cold path.  It runs on the memory-resident flavour of the solver (STREAM = true: every node field lives in the workgroup's work area),
 * in a function of its own (resto_entry, noinline) between two calls of the general iteration: the caller parks the register-resident
 * iterate in the work area, this code works on it there, the general iteration picks it up again (Solver::run, `resume`).  (n, p, z_n, z_p),
 * their steps, the reference point, the original bound multipliers and the filter of the restoration problem are further fields of the work
 * area (W_R*).  The Newton system: assemble(MODE_RESTO) + riccati_resto (serial sweeps on one lane; every loss model: the couplings of the dynamic loss
 * table with b_{i+1} and of the integrated loss rows with the running time enter as cross terms).  No second-order correction inside.
 */
static constexpr double RESTO_RHO = 1000.0;            /* resto_penalty_parameter */
static constexpr double RESTO_KAPPA = 0.9;             /* required_infeasibility_reduction */
static constexpr double RESTO_THETA_MAX_FACT = 1e8;    /* resto.theta_max_fact */
static constexpr double BOUND_MULT_RESET = 1e3;        /* bound_mult_reset_threshold */
static constexpr int RESTO_MAX_ITER = 100;             /* iterations of one restoration phase (ours, IPOPT has no limit: a phase that has found no acceptable point by then --
                                                        * those that succeed take 1 to 30 -- is given up: Restoration_Failed, then the restart from the other starting point) */
static constexpr int NCR = 2 + NR;
/* scalars handed over in field W_SCAL */
enum { SC_MU = 0, SC_THETA, SC_PHI, SC_ITER, SC_NFILT, SC_THETA_MAX, SC_THETA_MIN, SC_DELTA_LAST, SC_N_REG, SC_N_SOC, SC_N_BACK, SC_N_RESTO, SC_FORCED, SC_OBJ, SC_WD_SHORT, SC_N_WD, SC_SKIP_FIRST, SC_WD_ARM };

__device__ __forceinline__ double &wf(int field, int slot) const { return work[(size_t)field*NS + slot]; }
__device__ __forceinline__ bool rs_on(int jr) const { return jr < 2 || rowOn(jr >= 2 ? jr - 2 : 0); }
__device__ __forceinline__ double rs_D(int i, int jr) const { return wf(W_RN + jr, i)/wf(W_RZN + jr, i) + wf(W_RP + jr, i)/wf(W_RZP + jr, i); }
__device__ __forceinline__ double rs_row(int j, int jr) const
{
    return jr == 0 ? n[j].sct*resc[j][0] : jr == 1 ? n[j].scb*resc[j][1] : resd[j][jr >= 2 ? jr - 2 : 0];
}
__device__ __forceinline__ double rs_y(int j, int jr) const
{
    return jr == 0 ? n[j].lam[0]/n[j].sct : jr == 1 ? n[j].lam[1]/n[j].scb : n[j].nu[jr >= 2 ? jr - 2 : 0];
}

/* static data of the nodes, as Solver::run sets it up (the row scaling of the dynamics comes from the work area) */
__device__ __forceinline__ void resto_bind(double t0, double tEnd)
{
    const int N = P.N;
#pragma unroll 1
    for (int j = 0; j < SPT; j++) {
        NodeT &nd = n[j];
        nd.i = c.tid + j*c.nt;
        nd.bind(work + nd.i);
        resc[j].bind(work + W_RESC*NS + nd.i); resd[j].bind(work + W_RESD*NS + nd.i);
        evs[j].bind(work + W_EV*NS + nd.i); lgs[j].bind(work + W_LG*NS + nd.i);
        const bool ival = nd.i < N, node = nd.i <= N;
        nd.ds = ival ? P.ds[nd.i] : 0.0;
        nd.G = ival ? track_resistance(P, P.grad[nd.i], P.curv[nd.i]) : 0.0;
        const double bm = (nd.i >= 1 && nd.i < N) ? P.bmax[nd.i] : INFINITY;
        unsigned fl = (ival ? F_IVAL : 0u) | (node ? F_NODE : 0u);
        if (nd.i >= 1 && node && t0 != tEnd) fl |= F_ON_T;
        if (nd.i >= 1 && nd.i < N && P.vminSq != bm) fl |= F_ON_B;
        if (ival && P.fmin != P.fmax) fl |= F_ON_F;
        if (ival && withPn() && P.fminPn != 0.0) fl |= F_ON_P;
        if (ival) fl |= F_ON_S;
        nd.flags = fl;
        nd.ubB = bm + K_BOUND_RELAX*fmax(1.0, fabs(bm));
        nd.sct = wf(W_SC, nd.i); nd.scb = wf(W_SC + 1, nd.i);
    }
}

/*
 * Returns 1 restored (the iterate in the work area is the new point of the original problem: bound multipliers stepped towards mu/slack,
 * row multipliers zero), 0 failed, -1 locally infeasible, -2 iteration limit; nit = iterations taken.
 */
__device__ __forceinline__ int restoration(const double *scen, double *hist, int hist_cap, int &nit)
{
    const int N = P.N;
    const double rho = RESTO_RHO;
    resto_bind(scen[MSD_SC_T0], scen[MSD_SC_TEND]);
    const double mu_orig = uni(wf(W_SCAL, SC_MU)), theta_ref = uni(wf(W_SCAL, SC_THETA)), phi_ref = uni(wf(W_SCAL, SC_PHI)),
                 theta_max_o = uni(wf(W_SCAL, SC_THETA_MAX));
    const int iter0 = (int)uni(wf(W_SCAL, SC_ITER)), nfilt_o = (int)uni(wf(W_SCAL, SC_NFILT));
    double *rfilt = work + (size_t)W_RFILT*NS;
    int nf = 0, ret = 0, k = 0;

    /* reference point, original bound multipliers; rows at the reference point */
    evaluate_current();
    double cmax = 0;
#pragma unroll 1
    for (int j = 0; j < SPT; j++) {
        const NodeT &nd = n[j];
        if (!nd.node()) continue;
#pragma unroll
        for (int kk = 0; kk < NV; kk++) { wf(W_XR + kk, nd.i) = nd.x[kk]; wf(W_OZL + kk, nd.i) = nd.zL[kk]; wf(W_OZU + kk, nd.i) = nd.zU[kk]; }
#pragma unroll
        for (int r = 0; r < NR; r++) { wf(W_SGR + r, nd.i) = nd.sg[r]; wf(W_OZLS + r, nd.i) = nd.zLs[r]; wf(W_OZUS + r, nd.i) = nd.zUs[r]; }
        if (nd.ival())
#pragma unroll
            for (int jr = 0; jr < NCR; jr++) if (rs_on(jr)) cmax = fmax(cmax, fabs(rs_row(j, jr)));
    }
    { double v[1] = {cmax}; block_reduce<1>(v, OpMax(), c); cmax = uni(v[0]); }
    double mu = fmax(mu_orig, cmax), tau = fmax(K_TAU_MIN, 1 - mu), eta = sqrt(mu);
#pragma unroll 1
    for (int j = 0; j < SPT; j++) {
        NodeT &nd = n[j];
        if (!nd.node()) continue;
        if (nd.ival())
#pragma unroll
            for (int jr = 0; jr < NCR; jr++) {
                double rn = 1, rp = 1;
                if (rs_on(jr)) {
                    const double cv = rs_row(j, jr), a = (mu - rho*cv)/(2*rho);
                    rn = a + sqrt(a*a + mu*cv/(2*rho)); rp = cv + rn;
                }
                wf(W_RN + jr, nd.i) = rn; wf(W_RP + jr, nd.i) = rp; wf(W_RZN + jr, nd.i) = mu/rn; wf(W_RZP + jr, nd.i) = mu/rp;
            }
#pragma unroll
        for (int kk = 0; kk < NV; kk++) { nd.zL[kk] = fmin(rho, nd.zL[kk]); nd.zU[kk] = fmin(rho, nd.zU[kk]); }
#pragma unroll
        for (int r = 0; r < NR; r++) { nd.zLs[r] = fmin(rho, nd.zLs[r]); nd.zUs[r] = fmin(rho, nd.zUs[r]); nd.nu[r] = 0; }
        nd.lam[0] = nd.lam[1] = 0;
    }
    __syncthreads();

    double thmax = 0, thmin = 0, delta_last = 0, alpha_pr = 0, alpha_du = 0, dnorm = 0;
    int tiny_count = 0, n_back = 0;
    const double mu_floor = fmin(P.tol, 1e-4)/(K_EPS + 1.0);
    Err E;
    RsSums RS;

    for (k = 0;; k++) {
        if (k > 0) evaluate_current();
        kkt_pass(E, true, eta, rho, &RS);
        /* the original problem's merit pair at this point */
        const double th_o = E.theta, ph_o = E.obj - mu_orig*E.L + K_D*mu_orig*E.D;
        if (k == 0) { thmax = RESTO_THETA_MAX_FACT*fmax(1.0, RS.theta); thmin = 1e-4*fmax(1.0, RS.theta); }
        if (hist && c.tid == 0 && k > 0 && iter0 + k < hist_cap) {
            double *hh = hist + HIST_COLS*(iter0 + k);
            hh[0] = iter0 + k; hh[1] = E.obj/U.sf; hh[2] = th_o; hh[3] = E.dual; hh[4] = log10(mu); hh[5] = dnorm; hh[6] = alpha_du; hh[7] = alpha_pr;
        }
        /* back to the original problem?  (not before one step has been taken) */
        if (k >= 1 && isfinite(th_o) && isfinite(ph_o) && th_o <= RESTO_KAPPA*theta_ref && th_o <= theta_max_o && filter_ok(nfilt_o, th_o, ph_o)
            && (cmp_le(th_o, (1 - G_THETA)*theta_ref, theta_ref) || cmp_le(ph_o - phi_ref, -G_PHI*theta_ref, phi_ref))) { ret = 1; break; }
        /* the restoration problem itself solved: a stationary point of the infeasibility */
        {
            const double E0 = total_err(E, 0.0);
            if (E0 <= P.tol && E.dual <= 1.0 && E.primal <= 1e-4 && compl_err(E, 0.0) <= 1e-4) { ret = (E.primal_u <= 1e-4) ? 0 : -1; break; }
            if (!isfinite(E0)) { ret = 0; break; }
        }
        if (iter0 + k >= P.maxIter) { ret = -2; break; }
        if (k >= RESTO_MAX_ITER) { ret = 0; break; }

        /* barrier parameter of the restoration problem (monotone); the proximity weight follows it */
        {
            bool changed = false;
            while (total_err(E, mu) <= K_EPS*mu && mu > mu_floor) {
                const double nm = fmax(mu_floor, fmin(K_MU_LIN*mu, mu*sqrt(mu)));
                if (nm >= mu) break;
                mu = nm; tau = fmax(K_TAU_MIN, 1 - mu); changed = true;
            }
            if (changed) { nf = 0; eta = sqrt(mu); }
        }
        const double thR = RS.theta;
        const double phR = rho*RS.np + 0.5*eta*RS.prox - mu*(E.L + RS.lognp) + K_D*mu*(E.D + RS.np);

        /* right-hand sides of the relaxed rows, D of the dynamics rows */
#pragma unroll 1
        for (int j = 0; j < SPT; j++) {
            const NodeT &nd = n[j];
            if (!nd.ival()) continue;
            double rh[NCR];
#pragma unroll
            for (int jr = 0; jr < NCR; jr++) {
                rh[jr] = 0;
                if (!rs_on(jr)) continue;
                const double rn = wf(W_RN + jr, nd.i), rp = wf(W_RP + jr, nd.i), zn = wf(W_RZN + jr, nd.i), zp = wf(W_RZP + jr, nd.i);
                rh[jr] = rs_row(j, jr) + rn - rp + (mu - rho*rn)/zn - (mu - rho*rp)/zp;
            }
            resc[j][0] = rh[0]; resc[j][1] = rh[1];
#pragma unroll
            for (int r = 0; r < NR; r++) resd[j][r] = rh[2 + r];
            wf(W_RD, nd.i) = rs_D(nd.i, 0)/(nd.sct*nd.sct); wf(W_RD + 1, nd.i) = rs_D(nd.i, 1)/(nd.scb*nd.scb);
        }

        /* Newton direction with inertia correction */
        double dw = 0;
        bool ok;
        for (bool first = true;; first = false) {
            assemble(MODE_RESTO, mu, dw, eta);
            /* static loss rows: the sweeps on the stage-parallel scan (msd_resto_scan.hpp); -1: the scan broke down, or another loss model -- one lane sweeps */
            int par = -1;
#ifdef MSD_RESTO_CHECK      /* host emulation only: both solves of the same system side by side, the largest difference of their results printed */
            double *chk_blocks = nullptr, *chk_out = nullptr;
            if constexpr (DYN == LOSS_STATIC) {
                if (c.tid == 0) { chk_blocks = (double *)malloc(sizeof(double)*S_STRIDE*(N + 1)); memcpy(chk_blocks, c.S, sizeof(double)*S_STRIDE*(N + 1)); }
                __syncthreads();
            }
#endif
            if constexpr (DYN == LOSS_STATIC && MSD_PARALLEL_RESTO && MSD_PARALLEL_RICCATI) par = ParallelResto<SPT>::solve(N, withPn(), c, work + (size_t)W_RD*NS, work + (size_t)(W_RD + 1)*NS);
#ifdef MSD_RESTO_CHECK
            if constexpr (DYN == LOSS_STATIC) {
                __syncthreads();
                if (c.tid == 0) {
                    chk_out = (double *)malloc(sizeof(double)*S_STRIDE*(N + 1)); memcpy(chk_out, c.S, sizeof(double)*S_STRIDE*(N + 1));
                    memcpy(c.S, chk_blocks, sizeof(double)*S_STRIDE*(N + 1));
                    const bool sok = riccati_resto<DYN>(N, withPn(), c.S, work + (size_t)W_RD*NS, work + (size_t)(W_RD + 1)*NS, work + (size_t)W_RX*NS, NS);
                    double worst = 0; int wi = -1, wk = -1;
                    const int slots[7] = {S_DT, S_DB, S_DF, S_DP, S_DS, S_LT, S_LB};
                    if (sok && par == 1)
                        for (int k = 0; k < 7; k++) {      /* per kind of result: largest difference over the stages against the largest entry */
                            double scale = 0;
                            for (int i = 0; i < N; i++) scale = fmax(scale, fabs(c.S[i*S_STRIDE + slots[k]]));
                            for (int i = 0; i <= (k == 0 ? N : N - 1); i++) {
                                const double a = chk_out[i*S_STRIDE + slots[k]], b = c.S[i*S_STRIDE + slots[k]];
                                const double d = fabs(a - b)/fmax(1e-300, scale);
                                if (!(d <= worst)) { worst = d; wi = i; wk = k; }
                            }
                        }
                    printf("RESTO_CHECK N %d par %d serial %d worst rel diff %.3e at stage %d slot %d\n", N, par, (int)sok, worst, wi, wk);
                    if (par == 1) memcpy(c.S, chk_out, sizeof(double)*S_STRIDE*(N + 1));      /* go on with the parallel result */
                    free(chk_blocks); free(chk_out);
                }
                __syncthreads();
            }
#endif
            if (par < 0) {
                if (c.tid == 0) c.misc[0] = riccati_resto<DYN>(N, withPn(), c.S, work + (size_t)W_RD*NS, work + (size_t)(W_RD + 1)*NS, work + (size_t)W_RX*NS, NS) ? 1.0 : 0.0;
                __syncthreads();
                ok = uni(c.misc[0]) != 0.0;
                __syncthreads();
            } else ok = par == 1;
            if (ok) break;
            if (first) dw = (delta_last == 0) ? DW_0 : fmax(DW_MIN, KW_MINUS*delta_last);
            else dw *= (delta_last == 0) ? KW_PLUS_BAR : KW_PLUS;
            if (dw > DW_MAX) break;
        }
        if (!ok) { ret = 0; break; }
        if (dw > 0) delta_last = dw;

        /* steps of sigma, the row multipliers and (n, p, z_n, z_p); directional derivative of phi_R, norms, fraction to the boundary */
        double gphid, amax;
        bool tiny_step;
        {
            double gd = 0, dn = 0, rel = -1.0, rp_ = 0, rd_ = 0;
#pragma unroll 1
            for (int j = 0; j < SPT; j++) {
                NodeT &nd = n[j];
#pragma unroll
                for (int r = 0; r < NR; r++) nd.dsg[r] = 0;
                if (!nd.node()) continue;
                Dir d; load_dir(j, d);
#pragma unroll
                for (int kk = 0; kk < NV; kk++) {
                    if (!nd.on(kk)) continue;
                    const double dx = d.dx[kk];
                    const double xr = wf(W_XR + kk, nd.i), dr = 1.0/fmax(1.0, fabs(xr));
                    double gp;
                    {
                        const double r = 1.0/(nd.x[kk] - lbv(kk)), z = nd.zL[kk];
                        gp = -mu*r; rp_ = fmax(rp_, -dx*r);
                        rd_ = fmax(rd_, -(r*(mu - z*dx) - z)/z);
                    }
                    if (hasU(kk)) {
                        const double r = 1.0/(ubv(j, kk) - nd.x[kk]), z = nd.zU[kk];
                        gp += mu*r; rp_ = fmax(rp_, dx*r);
                        rd_ = fmax(rd_, -(r*(mu + z*dx) - z)/z);
                    } else gp += K_D*mu;
                    gd += (eta*dr*dr*(nd.x[kk] - xr) + gp)*dx;
                    dn = fmax(dn, fabs(dx)); rel = fmax(rel, fabs(dx) - 10*DBL_EPSILON*(1 + fabs(nd.x[kk])));
                }
                if (!nd.ival()) continue;
                const double db1 = c.S[(nd.i + 1)*S_STRIDE + S_DB];
                const double dd_ = INTEG ? c.S[(nd.i + 1)*S_STRIDE + S_DT] - d.dx[VT] : 0.0;      /* step of the running time (integrated loss rows) */
                double gb[NR], gf[NR], gp_[NR], gs[NR], gb1[NR], gdd[NR];
                Ev ej; load_ev1(j, ej);
                row_grads(j, ej, gb, gf, gp_, gs, gb1, gdd);
                double dy[NCR];
                dy[0] = (d.lt - nd.lam[0])/nd.sct; dy[1] = (d.lb - nd.lam[1])/nd.scb;
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    dy[2 + r] = 0;
                    if (!rowOn(r)) continue;
                    const double lin = resd[j][r] + gb[r]*d.dx[VB] + gf[r]*d.dx[VF] + gp_[r]*d.dx[VP] + gs[r]*d.dx[VS] + gb1[r]*db1 + gdd[r]*dd_;
                    double Sg, gphi; row_terms(j, r, mu, Sg, gphi);
                    const double Sw = Sg + dw, St = 1.0/(rs_D(nd.i, 2 + r) + 1.0/Sw), nup = St*lin + gphi*St/Sw;
                    const double ds_ = (nup - gphi)/Sw;
                    nd.dsg[r] = ds_; dy[2 + r] = nup - nd.nu[r]; wf(W_DNU + r, nd.i) = dy[2 + r];
                    double gpr = 0;
                    if (rL(r)) {
                        const double rr = 1.0/(nd.sg[r] - U.dL[r]), z = nd.zLs[r];
                        gpr -= mu*rr; rp_ = fmax(rp_, -ds_*rr);
                        rd_ = fmax(rd_, -(rr*(mu - z*ds_) - z)/z);
                    }
                    if (rU(r)) {
                        const double rr = 1.0/(U.dU[r] - nd.sg[r]), z = nd.zUs[r];
                        gpr += mu*rr; rp_ = fmax(rp_, ds_*rr);
                        rd_ = fmax(rd_, -(rr*(mu + z*ds_) - z)/z);
                    }
                    if (rL(r) && !rU(r)) gpr += K_D*mu;
                    if (!rL(r) && rU(r)) gpr -= K_D*mu;
                    gd += gpr*ds_;
                    dn = fmax(dn, fabs(ds_)); rel = fmax(rel, fabs(ds_) - 10*DBL_EPSILON*(1 + fabs(nd.sg[r])));
                }
#pragma unroll
                for (int jr = 0; jr < NCR; jr++) {
                    double dnn = 0, dpp = 0, dzn = 0, dzp = 0;
                    if (rs_on(jr)) {
                        const double rn = wf(W_RN + jr, nd.i), rp = wf(W_RP + jr, nd.i), zn = wf(W_RZN + jr, nd.i), zp = wf(W_RZP + jr, nd.i);
                        const double y = rs_y(j, jr);
                        dnn = (mu - rn*(rho + y))/zn - rn/zn*dy[jr];
                        dpp = (mu - rp*(rho - y))/zp + rp/zp*dy[jr];
                        dzn = rho + y + dy[jr] - zn;
                        dzp = rho - y - dy[jr] - zp;
                        gd += (rho - mu/rn + K_D*mu)*dnn + (rho - mu/rp + K_D*mu)*dpp;
                        dn = fmax(dn, fmax(fabs(dnn), fabs(dpp)));
                        rel = fmax(rel, fmax(fabs(dnn) - 10*DBL_EPSILON*(1 + rn), fabs(dpp) - 10*DBL_EPSILON*(1 + rp)));
                        rp_ = fmax(rp_, fmax(-dnn/rn, -dpp/rp));
                        rd_ = fmax(rd_, fmax(-dzn/zn, -dzp/zp));
                    }
                    wf(W_RDN + jr, nd.i) = dnn; wf(W_RDP + jr, nd.i) = dpp; wf(W_RDZN + jr, nd.i) = dzn; wf(W_RDZP + jr, nd.i) = dzp;
                }
            }
            double v1[1] = {gd}; block_reduce<1>(v1, OpSum(), c);
            double v2[4] = {dn, rel, rp_, rd_}; block_reduce<4>(v2, OpMax(), c);
            gphid = uni(v1[0]); dnorm = uni(v2[0]); tiny_step = uni(v2[1]) < 0;
            const double rpm = uni(v2[2]), rdm = uni(v2[3]);
            amax = (rpm > tau) ? tau/rpm : 1.0;
            alpha_du = (rdm > tau) ? tau/rdm : 1.0;
        }
        __syncthreads();

        double alpha = amax;
        bool accepted = false, ftype_armijo = false;
        if (tiny_step) {
            accepted = true;
            if (++tiny_count >= 2 && mu <= mu_floor*(1 + 1e-12)) { ret = 0; break; }
        } else tiny_count = 0;
        double amin = G_THETA;
        if (gphid < 0) {
            amin = fmin(amin, G_PHI*thR/(-gphid));
            if (thR <= thmin) amin = fmin(amin, K_DELTA*pow(thR, S_THETA)/pow(-gphid, S_PHI));
        }
        amin *= ALPHA_MIN_FRAC;
        while (!accepted) {
            double th_t, ph_t; bool okt;
            merit(alpha, mu, th_t, ph_t, okt, true, eta, rho);
            const bool ftype = (gphid < 0) && (alpha*pow(-gphid, S_PHI) > K_DELTA*pow(thR, S_THETA));
            bool acc = false;
            if (okt && th_t <= thmax) {
                if (ftype && thR <= thmin) acc = cmp_le(ph_t - phR, ETA_PHI*alpha*gphid, phR);
                else acc = cmp_le(th_t, (1 - G_THETA)*thR, thR) || cmp_le(ph_t - phR, -G_PHI*thR, phR);
                if (acc) for (int m = 0; m < nf; m++) if (th_t >= rfilt[2*m] && ph_t >= rfilt[2*m + 1]) { acc = false; break; }
            }
            if (acc) { accepted = true; ftype_armijo = ftype && cmp_le(ph_t - phR, ETA_PHI*alpha*gphid, phR); break; }
            alpha *= 0.5; n_back++;
            if (alpha < amin) break;
        }
        if (!accepted) { ret = 0; break; }
        alpha_pr = alpha;
        if (!tiny_step && !ftype_armijo && nf < FILT_CAP) {
            __syncthreads();
            if (c.tid == 0) { rfilt[2*nf] = (1 - G_THETA)*thR; rfilt[2*nf + 1] = phR - G_PHI*thR; }
            nf++;
            __syncthreads();
        }

        /* accept */
#pragma unroll 1
        for (int j = 0; j < SPT; j++) {
            NodeT &nd = n[j];
            if (!nd.node()) continue;
            Dir dd; load_dir(j, dd);
#pragma unroll
            for (int kk = 0; kk < NV; kk++) {
                if (!nd.on(kk)) continue;
                const double dzl = dzL_var(j, kk, mu, dd.dx[kk]), dzu = hasU(kk) ? dzU_var(j, kk, mu, dd.dx[kk]) : 0.0;
                nd.x[kk] += alpha_pr*dd.dx[kk];
                nd.zL[kk] += alpha_du*dzl;
                if (hasU(kk)) nd.zU[kk] += alpha_du*dzu;
                { const double s = nd.x[kk] - lbv(kk); nd.zL[kk] = sigma_clamp(nd.zL[kk], mu, s); }
                if (hasU(kk)) { const double s = ubv(j, kk) - nd.x[kk]; nd.zU[kk] = sigma_clamp(nd.zU[kk], mu, s); }
            }
            if (!nd.ival()) continue;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (!rowOn(r)) continue;
                const double dzl = rL(r) ? dzL_row(j, r, mu) : 0.0, dzu = rU(r) ? dzU_row(j, r, mu) : 0.0;
                nd.sg[r] += alpha_pr*nd.dsg[r];
                nd.nu[r] += alpha_pr*wf(W_DNU + r, nd.i);
                if (rL(r)) { nd.zLs[r] += alpha_du*dzl; const double s = nd.sg[r] - U.dL[r]; nd.zLs[r] = sigma_clamp(nd.zLs[r], mu, s); }
                if (rU(r)) { nd.zUs[r] += alpha_du*dzu; const double s = U.dU[r] - nd.sg[r]; nd.zUs[r] = sigma_clamp(nd.zUs[r], mu, s); }
            }
            nd.lam[0] += alpha_pr*(dd.lt - nd.lam[0]); nd.lam[1] += alpha_pr*(dd.lb - nd.lam[1]);
#pragma unroll
            for (int jr = 0; jr < NCR; jr++) {
                if (!rs_on(jr)) continue;
                const double rn = wf(W_RN + jr, nd.i) + alpha_pr*wf(W_RDN + jr, nd.i), rp = wf(W_RP + jr, nd.i) + alpha_pr*wf(W_RDP + jr, nd.i);
                wf(W_RN + jr, nd.i) = rn; wf(W_RP + jr, nd.i) = rp;
                wf(W_RZN + jr, nd.i) = sigma_clamp(wf(W_RZN + jr, nd.i) + alpha_du*wf(W_RDZN + jr, nd.i), mu, rn);
                wf(W_RZP + jr, nd.i) = sigma_clamp(wf(W_RZP + jr, nd.i) + alpha_du*wf(W_RDZP + jr, nd.i), mu, rp);
            }
        }
        __syncthreads();
    }

    if (ret == 1) {
        /* bound multipliers of the original problem: z + alpha dz, dz = (mu - z slack_new)/slack_old, alpha by the fraction-to-the-boundary rule;
         * all of them 1 when one ends above 1000.  Row multipliers zero (constr_mult_reset_threshold = 0) */
        const double tau_o = fmax(K_TAU_MIN, 1 - mu_orig);
        double rdm = 0;
#pragma unroll 1
        for (int j = 0; j < SPT; j++) {
            const NodeT &nd = n[j];
            if (!nd.node()) continue;
#pragma unroll
            for (int kk = 0; kk < NV; kk++) {
                if (!nd.on(kk)) continue;
                const double xr = wf(W_XR + kk, nd.i);
                { const double z = wf(W_OZL + kk, nd.i), dz = (mu_orig - z*(nd.x[kk] - lbv(kk)))/(xr - lbv(kk)); rdm = fmax(rdm, -dz/z); }
                if (hasU(kk)) { const double z = wf(W_OZU + kk, nd.i), dz = (mu_orig - z*(ubv(j, kk) - nd.x[kk]))/(ubv(j, kk) - xr); rdm = fmax(rdm, -dz/z); }
            }
            if (!nd.ival()) continue;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (!rowOn(r)) continue;
                const double sr = wf(W_SGR + r, nd.i);
                if (rL(r)) { const double z = wf(W_OZLS + r, nd.i), dz = (mu_orig - z*(nd.sg[r] - U.dL[r]))/(sr - U.dL[r]); rdm = fmax(rdm, -dz/z); }
                if (rU(r)) { const double z = wf(W_OZUS + r, nd.i), dz = (mu_orig - z*(U.dU[r] - nd.sg[r]))/(U.dU[r] - sr); rdm = fmax(rdm, -dz/z); }
            }
        }
        { double v[1] = {rdm}; block_reduce<1>(v, OpMax(), c); rdm = uni(v[0]); }
        const double a = (rdm > tau_o) ? tau_o/rdm : 1.0;
        double zmax = 0;
#pragma unroll 1
        for (int j = 0; j < SPT; j++) {
            NodeT &nd = n[j];
            if (!nd.node()) continue;
#pragma unroll
            for (int kk = 0; kk < NV; kk++) {
                if (!nd.on(kk)) { nd.zL[kk] = wf(W_OZL + kk, nd.i); nd.zU[kk] = wf(W_OZU + kk, nd.i); continue; }
                const double xr = wf(W_XR + kk, nd.i);
                { const double z = wf(W_OZL + kk, nd.i), dz = (mu_orig - z*(nd.x[kk] - lbv(kk)))/(xr - lbv(kk)); nd.zL[kk] = z + a*dz; zmax = fmax(zmax, nd.zL[kk]); }
                if (hasU(kk)) { const double z = wf(W_OZU + kk, nd.i), dz = (mu_orig - z*(ubv(j, kk) - nd.x[kk]))/(ubv(j, kk) - xr); nd.zU[kk] = z + a*dz; zmax = fmax(zmax, nd.zU[kk]); }
                else nd.zU[kk] = 0;
            }
#pragma unroll
            for (int r = 0; r < NR; r++) {
                nd.nu[r] = 0;
                if (!nd.ival() || !rowOn(r)) { nd.zLs[r] = wf(W_OZLS + r, nd.i); nd.zUs[r] = wf(W_OZUS + r, nd.i); continue; }
                const double sr = wf(W_SGR + r, nd.i);
                if (rL(r)) { const double z = wf(W_OZLS + r, nd.i), dz = (mu_orig - z*(nd.sg[r] - U.dL[r]))/(sr - U.dL[r]); nd.zLs[r] = z + a*dz; zmax = fmax(zmax, nd.zLs[r]); }
                if (rU(r)) { const double z = wf(W_OZUS + r, nd.i), dz = (mu_orig - z*(U.dU[r] - nd.sg[r]))/(U.dU[r] - sr); nd.zUs[r] = z + a*dz; zmax = fmax(zmax, nd.zUs[r]); }
            }
            nd.lam[0] = nd.lam[1] = 0;
        }
        { double v[1] = {zmax}; block_reduce<1>(v, OpMax(), c); zmax = uni(v[0]); }
        if (zmax > BOUND_MULT_RESET) {
#pragma unroll 1
            for (int j = 0; j < SPT; j++) {
                NodeT &nd = n[j];
                if (!nd.node()) continue;
#pragma unroll
                for (int kk = 0; kk < NV; kk++) if (nd.on(kk)) { nd.zL[kk] = 1; if (hasU(kk)) nd.zU[kk] = 1; }
                if (nd.ival())
#pragma unroll
                    for (int r = 0; r < NR; r++) if (rowOn(r)) { if (rL(r)) nd.zLs[r] = 1; if (rU(r)) nd.zUs[r] = 1; }
            }
        }
    } else {
        /* the original iterate keeps its multipliers; (x, sigma) stay where the restoration phase ended */
#pragma unroll 1
        for (int j = 0; j < SPT; j++) {
            NodeT &nd = n[j];
            if (!nd.node()) continue;
#pragma unroll
            for (int kk = 0; kk < NV; kk++) { nd.zL[kk] = wf(W_OZL + kk, nd.i); nd.zU[kk] = wf(W_OZU + kk, nd.i); }
#pragma unroll
            for (int r = 0; r < NR; r++) { nd.zLs[r] = wf(W_OZLS + r, nd.i); nd.zUs[r] = wf(W_OZUS + r, nd.i); nd.nu[r] = 0; }
            nd.lam[0] = nd.lam[1] = 0;
        }
    }
    __syncthreads();
    if (c.tid == 0) wf(W_SCAL, SC_N_BACK) += n_back;
    __syncthreads();
    nit = k;
    return ret;
}
