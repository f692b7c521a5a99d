// dense.hip — k_nr_dense: the Newton-Raphson power flow for ANY connected topology (meshed nets, closed tie switches),
// gfx950 (MI355X).  One env per workgroup; the full Jacobian lives in LDS as a dense matrix and is factorised by a
// blocked right-looking LU whose trailing updates are v_mfma_f64_16x16x4_f64 tiles — the only true contraction on the
// hot path (SURVEY.md 8(d) "dense K4 variant").  pandapower solves the same linear system with SuperLU
// (pypower/newtonpf.py: dx = -spsolve(J, F); reference call site voltage_control_env.py:557); the radial fast path
// (kernels.hip, k_nr_tree) eliminates it without fill and is used whenever the feeder is a tree.
//
// Same Newton iteration as k_nr_tree: flat start, polar form in the scaled unknowns z = [dtheta ; d|V|/|V|] interleaved
// per bus (rows P_k, Q_k; columns theta_k, ln|V_k|), full Jacobian every iteration, ||F||inf < tol, <= max_it
// iterations, V <- V (1 - z1) e^{-j z0}.  Pivoting is by 2x2 diagonal BLOCKS (one bus at a time, like the tree
// elimination): the bus blocks [[-Q - B|V|^2, P + G|V|^2], [P - G|V|^2, Q - B|V|^2]] have determinant ~ |Y_kk|^2 |V|^4
// whatever the R/X ratio, while a scalar pivot -Q - B|V|^2 vanishes on a purely resistive line.
//
// LDS: A[N][LDA] (N = 2n rounded up to 16, identity on the padding) | rhs[N] | dinv[N/2][4] | V pairs [n+2] | epilogue
// partials.  N <= 128 (nets of at most 65 buses) keeps the Jacobian LDS-resident.  Round 4: beyond that — the 141- and 322-bus
// MAPDN shapes, N = 288 / 656 — the same blocked LU runs with the Jacobian of every env in its own slab of GLOBAL memory
// (`GA` instantiations, one thread per row: 64 W >= N, up to 1024 threads): the panel, U12 and trailing tiles stream through
// L1 / L2 / HBM instead of LDS (`double*` is a flat pointer: dense_lu_solve is the same code), right-hand side, pivot inverses and
// voltages stay in LDS.  It is an exhibit of the one true contraction on this path at the sizes north_star names, not a fast
// path: a factorisation moves ~2/3 N^3 / 16 x 24 bytes through the memory system.
// After the solve the workgroup runs the same fused epilogue as the radial kernel (nr_common.hpp).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.hpp"
#include "nr_common.hpp"

namespace mapdn {

typedef double d4 __attribute__((ext_vector_type(4)));

// -------------------------------------------------------------------------------------------------------------------
// Blocked LU with 2x2 block pivots + forward / backward substitution of one right-hand side, all in LDS.
//   in : A (N x N, row stride LDA), rhs (N)          out: rhs = A^-1 rhs    (A is overwritten by its factors)
// Called by every thread of a 64*W-thread workgroup (N <= 64*W: thread r owns row r in the panel and substitution
// phases; wave w owns every W-th 16x16 tile of the trailing update).
// Panel kb (columns c0 = 16 kb ..): (1) eight 2x2 block steps on the 16 panel columns, each row held in registers,
// the two pivot rows published through LDS; (2) U12 = L11^-1 A12 and the same for the right-hand side, one thread
// per column, the column held in registers; (3) A22 -= L21 U12 as 16x16x16 tiles = 4 x v_mfma_f64_16x16x4_f64 each
// (A operand: lane l holds -L21[l & 15][4 kk + (l >> 4)]; B operand: U12[4 kk + (l >> 4)][l & 15]; C/D: 4 doubles per
// lane, row (l >> 4) + 4 reg, column l & 15), and rhs2 -= L21 y1 on the vector unit.
// -------------------------------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void dense_lu_solve(double* A, double* rhs, double* dinv, int N, int LDA) {
  const int r = (int)threadIdx.x;
  const int lane = r & 63, wave = r >> 6;
  const int NP = N >> 4;
  for (int kb = 0; kb < NP; ++kb) {
    const int c0 = kb << 4;
    // ---- (1) panel
    const bool own = r >= c0 && r < N;
    double p[16];
    {
      const double* row = A + (size_t)(own ? r : c0) * LDA + c0;
#pragma unroll
      for (int c = 0; c < 16; ++c) p[c] = row[c];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int pr = c0 + 2 * j;
      if (r == pr || r == pr + 1) {              // the two pivot rows publish their current panel row
        double* row = A + (size_t)r * LDA + c0;
#pragma unroll
        for (int c = 0; c < 16; ++c) row[c] = p[c];
      }
      __syncthreads();
      const double* r0 = A + (size_t)pr * LDA + c0;
      const double* r1 = r0 + LDA;
      const double D0 = r0[2 * j], D1 = r0[2 * j + 1], D2 = r1[2 * j], D3 = r1[2 * j + 1];
      const double idet = rcp_nr(D0 * D3 - D1 * D2);
      const double I0 = D3 * idet, I1 = -D1 * idet, I2 = -D2 * idet, I3 = D0 * idet;
      if (r == pr) { double* q = dinv + 4 * (pr >> 1); q[0] = I0; q[1] = I1; q[2] = I2; q[3] = I3; }
      if (own && r > pr + 1) {                   // L_r = A[r, pr:pr+2] D^-1, then the rest of the panel row
        const double l0 = p[2 * j] * I0 + p[2 * j + 1] * I2, l1 = p[2 * j] * I1 + p[2 * j + 1] * I3;
        p[2 * j] = l0; p[2 * j + 1] = l1;
#pragma unroll
        for (int c = 2 * j + 2; c < 16; ++c) p[c] -= l0 * r0[c] + l1 * r1[c];
      }
    }
    if (own && r >= c0 + 16) {                   // rows below the panel: their L block
      double* row = A + (size_t)r * LDA + c0;
#pragma unroll
      for (int c = 0; c < 16; ++c) row[c] = p[c];
    }
    __syncthreads();
    // ---- (2) U12 = L11^-1 A12 (thread per column right of the panel) and y1 = L11^-1 b1 (one more thread)
    const int ncol = N - c0 - 16;
    if (r <= ncol) {
      const bool is_rhs = r == ncol;
      double* col = is_rhs ? rhs + c0 : A + (size_t)c0 * LDA + (c0 + 16 + r);
      const int cs = is_rhs ? 1 : LDA;
      double a[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) a[i] = col[(size_t)i * cs];
      const double* L11 = A + (size_t)c0 * LDA + c0;
#pragma unroll
      for (int jb = 0; jb < 7; ++jb) {
#pragma unroll
        for (int i = 2 * jb + 2; i < 16; ++i)
          a[i] -= L11[(size_t)i * LDA + 2 * jb] * a[2 * jb] + L11[(size_t)i * LDA + 2 * jb + 1] * a[2 * jb + 1];
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) col[(size_t)i * cs] = a[i];
    }
    __syncthreads();
    // ---- (3) trailing update
    const int nt = NP - kb - 1;
    for (int tix = wave; tix < nt * nt; tix += W) {
      const int I = kb + 1 + tix / nt, J = kb + 1 + tix % nt;
      double* C = A + (size_t)(16 * I + (lane >> 4)) * LDA + 16 * J + (lane & 15);
      d4 c = {C[0], C[(size_t)4 * LDA], C[(size_t)8 * LDA], C[(size_t)12 * LDA]};
      const double* La = A + (size_t)(16 * I + (lane & 15)) * LDA + c0 + (lane >> 4);
      const double* Ub = A + (size_t)(c0 + (lane >> 4)) * LDA + 16 * J + (lane & 15);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(-La[4 * kk], Ub[(size_t)(4 * kk) * LDA], c, 0, 0, 0);
      C[0] = c[0]; C[(size_t)4 * LDA] = c[1]; C[(size_t)8 * LDA] = c[2]; C[(size_t)12 * LDA] = c[3];
    }
    if (r >= c0 + 16 && r < N) {
      const double* row = A + (size_t)r * LDA + c0;
      double s = rhs[r];
#pragma unroll
      for (int i = 0; i < 16; ++i) s -= row[i] * rhs[c0 + i];
      // (rhs[c0 .. c0+15] is read-only in this phase, rhs[r] is written by its owner only)
      rhs[r] = s;
    }
    __syncthreads();
  }
  // ---- back substitution U z = y, block by block from the bottom; thread r keeps y_r in a register
  double yr = r < N ? rhs[r] : 0.0;
  for (int jb = (N >> 1) - 1; jb >= 0; --jb) {
    const int c = 2 * jb;
    if (r == c || r == c + 1) rhs[r] = yr;
    __syncthreads();
    const double* q = dinv + 4 * jb;
    const double b0 = rhs[c], b1 = rhs[c + 1];
    const double z0 = q[0] * b0 + q[1] * b1, z1 = q[2] * b0 + q[3] * b1;
    if (r < c) { const double* row = A + (size_t)r * LDA + c; yr -= row[0] * z0 + row[1] * z1; }
    else if (r == c) yr = z0;
    else if (r == c + 1) yr = z1;
  }
  __syncthreads();                                // (the last readers of rhs[0], rhs[1] are done)
  if (r < N) rhs[r] = yr;
  __syncthreads();
}

static inline __host__ __device__ size_t dense_lds_doubles(int N, int LDA, int n, int tpb, bool global_a = false) {
  return (global_a ? 0 : (size_t)N * LDA) + (size_t)N + (size_t)2 * N + (size_t)2 * (n + 2) + (size_t)10 * tpb;
}

template <int W, bool GA>
__global__ void __launch_bounds__(64 * W)
k_nr_dense(Dev d, int mode, double* __restrict__ reward, uint8_t* __restrict__ terminated, double* __restrict__ info) {
  extern __shared__ double lds[];
  constexpr unsigned TPB = 64u * W;
  const int tid = (int)threadIdx.x;
  const unsigned e = blockIdx.x;
  const int n = d.n, N = d.dn_N, LDA = d.dn_lda;
  double* A = GA ? d.dn_A + (size_t)e * N * LDA : lds;      // GA: this env's slab of the global Jacobian scratch
  double* rhs = GA ? lds : A + (size_t)N * LDA;
  double* dinv = rhs + N;
  d2* sV = (d2*)(dinv + 2 * N);                   // (N * LDA + 3 N and 3 N are even: 16-byte aligned)
  double* s_epi = (double*)(sV + (n + 2));
  const double vroot = d.vroot, tol = d.tol;
  for (int k = tid; k < n + 2; k += TPB) sV[k] = d2{vroot, 0.0};     // runpp init="auto": flat start at the slack set-point
  const bool act = d.active[e] != 0;
  const int bk_steps = d.steps[e];
  const uint32_t bk_draw = d.draw[e];
  const double bk_sum = d.sum_rewards[e];
  typedef const __attribute__((address_space(4))) int32_t* c_i32;
  typedef const __attribute__((address_space(4))) double* c_f64;
  const c_i32 yptr = (c_i32)(unsigned long long)d.gy_ptr, ycol = (c_i32)(unsigned long long)d.gy_col;
  const c_f64 yval = (c_f64)(unsigned long long)d.gy_val;
  const bool bus = tid < n;                       // thread k < n serves bus position k
  double sbr = 0.0, sbi = 0.0;
  int p0 = 0, p1 = 0;
  if (bus) {
    const double2 sb = ((const double2*)((const char*)d.nrbuf + d.sb_off))[(size_t)d.sb_index[tid] * d.Bp + e];
    sbr = sb.x; sbi = sb.y;
    p0 = yptr[tid]; p1 = yptr[tid + 1];
  }
  __syncthreads();
  bool conv = false;
  int it = 0;
  if (act) {                                      // (uniform: one env per workgroup)
    for (;;) {
      // ---- S_k = V_k conj(sum_j Y_kj V_j), mismatch F_k = S_k - Sbus_k            (newtonpf: _evaluate_Fx)
      double sr = 0.0, si = 0.0, akk_r = 0.0, akk_i = 0.0, ek = 0.0, fk = 0.0;
      if (bus) {
        const d2 vk = sV[tid];
        ek = vk.x; fk = vk.y;
        for (int q = p0; q < p1; ++q) {
          const int j = ycol[q];
          const double g = yval[2 * q], b = yval[2 * q + 1];
          const d2 vj = sV[j];
          const double tr = g * vj.x - b * vj.y, ti = g * vj.y + b * vj.x;
          const double ar = ek * tr + fk * ti, ai = fk * tr - ek * ti;         // A_kj = V_k conj(Y_kj V_j)
          sr += ar; si += ai;
          if (j == tid) { akk_r = ar; akk_i = ai; }
        }
      }
      const double Fp = sr - sbr, Fq = si - sbi;
      const bool ok = !bus || (fabs(Fp) < tol && fabs(Fq) < tol);
      conv = __syncthreads_and(ok ? 1 : 0) != 0;  // wave-level AND (ballot) combined across the waves
      if (conv || it == d.max_it) break;
      // ---- Jacobian in the scaled unknowns: dS_k/dtheta_j = -j A_kj, dS_k/dln|V_j| = A_kj  (j != k);
      //      dS_k/dtheta_k = j (S_k - A_kk), dS_k/dln|V_k| = S_k + A_kk                       (create_jacobian_matrix)
      for (int i = tid; i < N * LDA / 2; i += TPB) ((d2*)A)[i] = d2{0.0, 0.0};
      __syncthreads();
      if (bus) {
        double* r0 = A + (size_t)(2 * tid) * LDA;
        double* r1 = r0 + LDA;
        for (int q = p0; q < p1; ++q) {
          const int j = ycol[q];
          if (j >= n || j == tid) continue;       // the slack column is not an unknown
          const double g = yval[2 * q], b = yval[2 * q + 1];
          const d2 vj = sV[j];
          const double tr = g * vj.x - b * vj.y, ti = g * vj.y + b * vj.x;
          const double ar = ek * tr + fk * ti, ai = fk * tr - ek * ti;
          r0[2 * j] = ai; r0[2 * j + 1] = ar; r1[2 * j] = -ar; r1[2 * j + 1] = ai;
        }
        r0[2 * tid] = -(si - akk_i); r0[2 * tid + 1] = sr + akk_r;
        r1[2 * tid] = sr - akk_r;    r1[2 * tid + 1] = si + akk_i;
        rhs[2 * tid] = Fp; rhs[2 * tid + 1] = Fq;
      }
      if (tid >= 2 * n && tid < N) { A[(size_t)tid * LDA + tid] = 1.0; rhs[tid] = 0.0; }   // padding: identity
      __syncthreads();
      dense_lu_solve<W>(A, rhs, dinv, N, LDA);
      // ---- newtonpf update: Va -= z0, Vm -= |V| z1, V = Vm e^{jVa}   =>   V <- V (1 - z1) e^{-j z0}
      if (bus) {
        const double y0 = rhs[2 * tid], y1 = rhs[2 * tid + 1];
        double s, c;
        sincos(-y0, &s, &c);
        sV[tid] = nr_rotate(d2{ek, fk}, s, c, y1);
      }
      ++it;
      __syncthreads();
    }
  }
  if (tid == 0) { d.iters[e] = it; d.conv[e] = conv ? 1 : 0; }
  nr_epilogue<1u, TPB>(d, mode, reward, terminated, info, sV, (unsigned)tid, e, act, conv, bk_steps, bk_draw, bk_sum, s_epi,
                       (const double*)nullptr, [](int) {});
}

// debug / pin entry: solve `batch` dense systems (row-major n x n, n even) with the kernel's own LU — tests compare with
// numpy.linalg.solve (tests/test_gpu_parity.py::test_dense_lu_matches_numpy)
template <int W, bool GA>
__global__ void __launch_bounds__(64 * W) k_dense_solve(const double* __restrict__ Ain, const double* __restrict__ bin,
                                                       double* __restrict__ xout, int n, int N, int LDA, double* __restrict__ scratch) {
  extern __shared__ double lds[];
  constexpr int TPB = 64 * W;
  const size_t s = blockIdx.x;
  double* A = GA ? scratch + s * (size_t)N * LDA : lds;
  double* rhs = GA ? lds : A + (size_t)N * LDA;
  double* dinv = rhs + N;
  const int tid = (int)threadIdx.x;
  for (int i = tid; i < N * LDA; i += TPB) A[i] = 0.0;
  __syncthreads();
  for (int i = tid; i < n * n; i += TPB) A[(size_t)(i / n) * LDA + i % n] = Ain[s * n * n + i];
  for (int i = tid; i < N; i += TPB) {
    rhs[i] = i < n ? bin[s * n + i] : 0.0;
    if (i >= n) A[(size_t)i * LDA + i] = 1.0;
  }
  __syncthreads();
  dense_lu_solve<W>(A, rhs, dinv, N, LDA);
  for (int i = tid; i < n; i += TPB) xout[s * n + i] = rhs[i];
}

// waves per workgroup: one thread per matrix row.  LDS-resident Jacobian: N <= 64 -> 1, <= 128 -> 2; global Jacobian: the
// instantiated W >= N / 64
static const int DENSE_GA_W[] = {3, 4, 5, 6, 8, 11, 13, 16};
static int dense_waves(const Dev& d) {
  if (!d.dn_A) return d.dn_N <= 64 ? 1 : 2;
  for (int w : DENSE_GA_W) if (64 * w >= d.dn_N) return w;
  return 0;
}
size_t nr_dense_lds_bytes(const Dev& d) {
  const int W = dense_waves(d);
  return dense_lds_doubles(d.dn_N, d.dn_lda, d.n, 64 * W, d.dn_A != nullptr) * sizeof(double);
}

// dynamic LDS limit of k_nr_dense: the CU's 160 KB minus the 256 bytes of static LDS that __syncthreads_and's
// cross-wave reduction allocates (asking for all 160 KB makes hipFuncSetAttribute fail with invalid value)
static constexpr size_t DENSE_LDS_MAX = 160 * 1024 - 256;

#define DENSE_GA_FOR_EACH(X) X(3) X(4) X(5) X(6) X(8) X(11) X(13) X(16)
static const void* dense_fn(const Dev& d) {
  const int W = dense_waves(d);
  if (!d.dn_A) return W == 1 ? (const void*)k_nr_dense<1, false> : (const void*)k_nr_dense<2, false>;
#define X(w) if (W == w) return (const void*)k_nr_dense<w, true>;
  DENSE_GA_FOR_EACH(X)
#undef X
  return nullptr;
}

int nr_dense_prepare(const Dev& d) {
  const size_t lds = nr_dense_lds_bytes(d);
  const void* f = dense_fn(d);
  if (!f || (!d.dn_A && d.dn_N > 128) || lds > DENSE_LDS_MAX) return -2;
  return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DENSE_LDS_MAX) == hipSuccess ? 0 : -1;
}

void launch_nr_dense(const Dev& d, int mode, double* reward, uint8_t* term, double* info, hipStream_t st) {
  const size_t lds = nr_dense_lds_bytes(d);
  const void* f = dense_fn(d);
  if (!f) return;
  Dev dd = d;
  void* args[] = {(void*)&dd, (void*)&mode, (void*)&reward, (void*)&term, (void*)&info};
  (void)hipLaunchKernel(f, dim3(d.Bp), dim3(64 * dense_waves(d)), args, lds, st);
}

int dense_solve_debug(const double* A, const double* b, double* x, int n, int batch, hipStream_t st) {
  if (n < 2 || (n & 1) || n > 1024 || batch < 1) return -1;
  const int N = (n + 15) / 16 * 16, LDA = N + 2;
  if (N <= 128) {
    const size_t lds = ((size_t)N * LDA + 3 * (size_t)N) * sizeof(double);
    const void* f = N <= 64 ? (const void*)k_dense_solve<1, false> : (const void*)k_dense_solve<2, false>;
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2;
    double* none = nullptr;
    void* args[] = {(void*)&A, (void*)&b, (void*)&x, (void*)&n, (void*)&N, (void*)&LDA, (void*)&none};
    if (hipLaunchKernel(f, dim3(batch), dim3(N <= 64 ? 64 : 128), args, lds, st) != hipSuccess) return -2;
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  // the matrix in global memory (the form k_nr_dense uses beyond 65 buses); synchronous: the scratch is freed on return
  int W = 0;
  for (int w : DENSE_GA_W) if (64 * w >= N) { W = w; break; }
  if (!W) return -1;
  double* scratch = nullptr;
  if (hipMalloc((void**)&scratch, (size_t)batch * N * LDA * sizeof(double)) != hipSuccess) return -2;
  const void* f = nullptr;
#define X(w) if (W == w) f = (const void*)k_dense_solve<w, true>;
  DENSE_GA_FOR_EACH(X)
#undef X
  const size_t lds = (3 * (size_t)N) * sizeof(double);
  void* args[] = {(void*)&A, (void*)&b, (void*)&x, (void*)&n, (void*)&N, (void*)&LDA, (void*)&scratch};
  int rc = hipLaunchKernel(f, dim3(batch), dim3(64 * W), args, lds, st) == hipSuccess ? 0 : -2;
  if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) rc = -2;
  (void)hipFree(scratch);
  return rc;
}

}  // namespace mapdn
