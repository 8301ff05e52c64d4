/*
 * msd_scan.hpp -- algebra of the stage-parallel Riccati recursion (device code, included by msd_kernel.hpp).
 *
 * The serial backward sweep maps the value function of stage i+1 to that of stage i,
 *     P_i = J_i + A_i^T P_{i+1} (I + C_i P_{i+1})^-1 A_i,
 * with the control-eliminated stage data  A = Fx - Fu R^-1 S,  C = Fu R^-1 Fu^T,  J = Q - S^T R^-1 S
 * (R = control block of the condensed stage Hessian).  These maps compose associatively on the triple (A, C, J):
 * for a run of stages e1 followed by a run e2
 *     M   = I + C1 J2
 *     A12 = A2 M^-1 A1,   C12 = C2 + A2 (M^-1 C1) A2^T,   J12 = J1 + A1^T (J2 M^-1) A1
 * (the conditional value function of the run in its dual form; Saerkkae & Garcia-Fernandez, "Temporal parallelization of
 * dynamic programming and linear quadratic control", IEEE TAC 2023, restated for a backward suffix product).  So the value
 * functions at the chunk boundaries of a horizon follow from a log-depth suffix scan over the lanes of a wave, and every lane
 * then runs the ordinary recursion over its own few stages.  Only the boundary matrices come from the scan: pivots (inertia),
 * feedback and the affine terms are those of the ordinary recursion.
 *
 * The affine parts of both sweeps (value-function gradient backward, state forward) are scans over 3x3 affine maps.
 */
#pragma once

namespace msd {

/* packed symmetric 3x3: tt tb tq bb bq qq */
__device__ __forceinline__ constexpr int sy(int i, int j) { return i <= j ? (i == 0 ? j : i == 1 ? 2 + j : 5) : (j == 0 ? i : j == 1 ? 2 + i : 5); }

struct Elem { double A[3][3], C[6], J[6]; };

__device__ __forceinline__ void elem_identity(Elem &e)
{
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) e.A[i][j] = (i == j) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) { e.C[k] = 0; e.J[k] = 0; }
}

/* inverse of a 3x3 matrix by cofactors: one reciprocal, no pivot search; det is returned for the caller's sanity check */
__device__ __forceinline__ double inv3(const double (&m)[3][3], double (&r)[3][3])
{
    const double c00 = m[1][1]*m[2][2] - m[1][2]*m[2][1], c10 = m[1][2]*m[2][0] - m[1][0]*m[2][2], c20 = m[1][0]*m[2][1] - m[1][1]*m[2][0];
    const double det = m[0][0]*c00 + m[0][1]*c10 + m[0][2]*c20;
    const double id = frcp(det);
    r[0][0] = c00*id; r[1][0] = c10*id; r[2][0] = c20*id;
    r[0][1] = (m[0][2]*m[2][1] - m[0][1]*m[2][2])*id; r[1][1] = (m[0][0]*m[2][2] - m[0][2]*m[2][0])*id; r[2][1] = (m[0][1]*m[2][0] - m[0][0]*m[2][1])*id;
    r[0][2] = (m[0][1]*m[1][2] - m[0][2]*m[1][1])*id; r[1][2] = (m[0][2]*m[1][0] - m[0][0]*m[1][2])*id; r[2][2] = (m[0][0]*m[1][1] - m[0][1]*m[1][0])*id;
    return det;
}

/* shared front of both combines: Mi = (I + C1 J2)^-1 and N = J2 Mi (symmetric, packed) */
__device__ __forceinline__ double scan_kernel(const double (&C1)[6], const double (&J2)[6], double (&Mi)[3][3], double (&Nn)[6])
{
    double M[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) M[i][j] = ((i == j) ? 1.0 : 0.0) + C1[sy(i, 0)]*J2[sy(0, j)] + C1[sy(i, 1)]*J2[sy(1, j)] + C1[sy(i, 2)]*J2[sy(2, j)];
    const double det = inv3(M, Mi);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i; j < 3; j++) Nn[sy(i, j)] = J2[sy(i, 0)]*Mi[0][j] + J2[sy(i, 1)]*Mi[1][j] + J2[sy(i, 2)]*Mi[2][j];
    return det;
}

/* J of (e1 followed by a value function P2): J1 + A1^T (P2 (I + C1 P2)^-1) A1 */
__device__ __forceinline__ double combine_value(const Elem &e1, const double (&P2)[6], double (&P)[6])
{
    double Mi[3][3], Nn[6], T[3][3];
    const double det = scan_kernel(e1.C, P2, Mi, Nn);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) T[i][j] = Nn[sy(i, 0)]*e1.A[0][j] + Nn[sy(i, 1)]*e1.A[1][j] + Nn[sy(i, 2)]*e1.A[2][j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i; j < 3; j++) P[sy(i, j)] = e1.J[sy(i, j)] + e1.A[0][i]*T[0][j] + e1.A[1][i]*T[1][j] + e1.A[2][i]*T[2][j];
    return det;
}

/* e1 followed by e2 (e1 covers the earlier stages) */
__device__ __forceinline__ double combine(const Elem &e1, const Elem &e2, Elem &o)
{
    double Mi[3][3], Nn[6], MC[6], T[3][3], U[3][3];
    const double det = scan_kernel(e1.C, e2.J, Mi, Nn);
    /* M^-1 C1 is symmetric */
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i; j < 3; j++) MC[sy(i, j)] = Mi[i][0]*e1.C[sy(0, j)] + Mi[i][1]*e1.C[sy(1, j)] + Mi[i][2]*e1.C[sy(2, j)];
    /* J = J1 + A1^T N A1 */
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) T[i][j] = Nn[sy(i, 0)]*e1.A[0][j] + Nn[sy(i, 1)]*e1.A[1][j] + Nn[sy(i, 2)]*e1.A[2][j];
    Elem r;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i; j < 3; j++) r.J[sy(i, j)] = e1.J[sy(i, j)] + e1.A[0][i]*T[0][j] + e1.A[1][i]*T[1][j] + e1.A[2][i]*T[2][j];
    /* A = A2 (M^-1 A1) */
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) U[i][j] = Mi[i][0]*e1.A[0][j] + Mi[i][1]*e1.A[1][j] + Mi[i][2]*e1.A[2][j];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) r.A[i][j] = e2.A[i][0]*U[0][j] + e2.A[i][1]*U[1][j] + e2.A[i][2]*U[2][j];
    /* C = C2 + A2 (M^-1 C1) A2^T */
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) T[i][j] = MC[sy(i, 0)]*e2.A[j][0] + MC[sy(i, 1)]*e2.A[j][1] + MC[sy(i, 2)]*e2.A[j][2];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = i; j < 3; j++) r.C[sy(i, j)] = e2.C[sy(i, j)] + e2.A[i][0]*T[0][j] + e2.A[i][1]*T[1][j] + e2.A[i][2]*T[2][j];
    o = r;
    return det;
}

/* affine map x -> M x + v */
struct Aff { double M[3][3], v[3]; };

__device__ __forceinline__ void aff_identity(Aff &a)
{
#pragma unroll
    for (int i = 0; i < 3; i++) {
        a.v[i] = 0;
#pragma unroll
        for (int j = 0; j < 3; j++) a.M[i][j] = (i == j) ? 1.0 : 0.0;
    }
}

/* o = outer after inner: x -> outer(inner(x)) */
__device__ __forceinline__ void aff_compose(const Aff &outer, const Aff &inner, Aff &o)
{
    Aff r;
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) r.M[i][j] = outer.M[i][0]*inner.M[0][j] + outer.M[i][1]*inner.M[1][j] + outer.M[i][2]*inner.M[2][j];
        r.v[i] = outer.M[i][0]*inner.v[0] + outer.M[i][1]*inner.v[1] + outer.M[i][2]*inner.v[2] + outer.v[i];
    }
    o = r;
}

__device__ __forceinline__ void aff_apply(const Aff &a, const double (&x)[3], double (&y)[3])
{
    double r[3];
#pragma unroll
    for (int i = 0; i < 3; i++) r[i] = a.M[i][0]*x[0] + a.M[i][1]*x[1] + a.M[i][2]*x[2] + a.v[i];
#pragma unroll
    for (int i = 0; i < 3; i++) y[i] = r[i];
}

}  // namespace msd
