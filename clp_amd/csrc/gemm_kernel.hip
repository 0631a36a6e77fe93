// gemm_kernel.hip -- the engine's own f64 GEMM on the matrix cores (gfx950, v_mfma_f64_16x16x4_f64):
//   C = beta * C + alpha * A * B      row-major, arbitrary M x N x K, leading dimensions lda / ldb / ldc.
// Used by the Newton-Schulz steps on an explicit inverse (polish of the LU mode's dense tail, verified refresh of
// the explicit-inverse mode): R = I - S X and X += X R are two k^3 products each.  The reference's counterpart is
// CoinAbcDgemm (src/CoinAbcHelperFunctions.cpp:1658), the blocked update inside CoinAbcDgetrf
// (src/AbcSimplexParallel.cpp:2491-2534).
//
// Tiling: a workgroup of 4 waves owns a 128 x 128 tile of C; wave w owns rows [32 w, 32 w + 32) x 128 columns =
// 2 x 8 MFMA sub-tiles of 16 x 16 (16 accumulators of 4 f64 per lane).  K advances in steps of 16: the 128 x 16
// slab of A and the 16 x 128 slab of B are staged in LDS ([k][row] / [k][column], so that the 16 lanes of an
// operand read consecutive words), the next slabs are fetched into registers while the current ones feed the
// MFMAs.  64 MFMAs (131 k flop) per wave and stage against 40 LDS operand loads per lane; 2 x 16 KB of global
// loads per workgroup and stage -> 16 flop per byte, above the HBM/L2 ridge of the f64 matrix pipe.
// Lane layout of the instruction (as in k_gj2_trail_mfma): lane l supplies A[l & 15][l >> 4], B[l >> 4][l & 15]
// and receives D[(l >> 4) + 4 v][l & 15], v = 0..3.
#pragma once

namespace clpgpu {

#define DG_T 128
#define DG_K 16
#define DG_LD (DG_T + 4)

__global__ void __launch_bounds__(256) k_dgemm(int M, int N, int K, double alpha, const double *A, int lda, const double *B, int ldb, double beta,
                                               double *C, int ldc)
{
  __shared__ double sA[DG_K][DG_LD];
  __shared__ double sB[DG_K][DG_LD];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int row0 = blockIdx.y * DG_T, col0 = blockIdx.x * DG_T;
  // global -> register staging: A slab 128 rows x 16: thread t takes row t >> 1, 8 consecutive k from (t & 1) * 8;
  // B slab 16 rows x 128: thread t takes k row t >> 4, 8 consecutive columns from (t & 15) * 8
  const int aRow = tid >> 1, aK = (tid & 1) * 8;
  const int bK = tid >> 4, bCol = (tid & 15) * 8;
  double ra[8], rb[8];
  auto fetch = [&](int k0) {
    const int gr = row0 + aRow;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int gk = k0 + aK + u;
      ra[u] = (gr < M && gk < K) ? A[(size_t)gr * lda + gk] : 0.0;
    }
    const int gkb = k0 + bK;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int gc = col0 + bCol + u;
      rb[u] = (gkb < K && gc < N) ? B[(size_t)gkb * ldb + gc] : 0.0;
    }
  };
  gj_v4d acc[2][8];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 8; j++)
      acc[i][j] = gj_v4d{ 0.0, 0.0, 0.0, 0.0 };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += DG_K) {
    __syncthreads();  // the previous stage's operand reads are done
#pragma unroll
    for (int u = 0; u < 8; u++) {
      sA[aK + u][aRow] = ra[u];
      sB[bK][bCol + u] = rb[u];
    }
    __syncthreads();
    if (k0 + DG_K < K)
      fetch(k0 + DG_K);
#pragma unroll
    for (int ks = 0; ks < DG_K; ks += 4) {
      const int kk = ks + (lane >> 4);
      double av[2], bv[8];
#pragma unroll
      for (int i = 0; i < 2; i++)
        av[i] = sA[kk][wv * 32 + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 8; j++)
        bv[j] = sB[kk][j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 8; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
      for (int v = 0; v < 4; v++) {
        const int r = row0 + wv * 32 + i * 16 + (lane >> 4) + 4 * v;
        const int cc = col0 + j * 16 + (lane & 15);
        if (r < M && cc < N) {
          double *p = C + (size_t)r * ldc + cc;
          const double old = beta != 0.0 ? *p : 0.0;
          *p = beta * old + alpha * acc[i][j][v];
        }
      }
}

}  // namespace clpgpu
