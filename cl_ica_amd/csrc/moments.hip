// Second-moment matrix of [A | B | 1] in fp64  --  everything the disentanglement scores of
// /root/reference/disentanglement_utils.py:17-221 need of the data, in one pass over it:
//     G = [A | B | 1]^T [A | B | 1]      (d x d, d = da + db + 1, fp64 accumulation of exact fp32 x fp32 products)
// holds X^T X and X^T Z of the linear regression behind `linear_disentanglement` (sklearn LinearRegression, :97-100), the sums
// of squares of r2_score (:23), and the covariances of np.corrcoef (:40) / of the ranks for spearmanr (:38).  The reference does
// this on the host (D2H copy of 4096 x n embeddings + sklearn / scipy) every n_log_steps; here the d x d matrix (d <= 129) is all
// that leaves the device.  SURVEY.md 8(f) N2.
//
// HBM-bound in principle (reads M (da + db) floats once); at the evaluation sizes (M = 4096, n <= 64) it is launch-latency bound:
// two launches (row-chunk partial matrices, deterministic reduction over the chunks -- no atomics).
#include "common.h"

namespace clica {
namespace moments {
constexpr int THREADS = 256, ROWS = 64, MAXD = 129;

__global__ __launch_bounds__(THREADS) void partial_k(const float* __restrict__ A, int64_t lda, int da, const float* __restrict__ B, int64_t ldb,
                                                     int db, int64_t M, double* __restrict__ part) {
  __shared__ float tile[ROWS][MAXD + 1];
  const int d = da + db + 1;
  const int64_t r0 = (int64_t)blockIdx.x * ROWS;
  const int rows = (int)min((int64_t)ROWS, M - r0);
  for (int idx = threadIdx.x; idx < ROWS * d; idx += THREADS) {
    const int r = idx / d, c = idx - r * d;
    float v = 0.f;
    if (r < rows) v = c < da ? A[(r0 + r) * lda + c] : (c < da + db ? B[(r0 + r) * ldb + (c - da)] : 1.f);
    tile[r][c] = v;
  }
  __syncthreads();
  double* out = part + (int64_t)blockIdx.x * d * d;
  for (int idx = threadIdx.x; idx < d * d; idx += THREADS) {
    const int i = idx / d, j = idx - i * d;
    double s = 0.0;
#pragma unroll 8
    for (int r = 0; r < ROWS; ++r) s += (double)tile[r][i] * (double)tile[r][j];      // rows beyond the batch are zero
    out[idx] = s;
  }
}

__global__ __launch_bounds__(THREADS) void reduce_k(const double* __restrict__ part, int64_t chunks, int dd, double* __restrict__ out) {
  const int idx = blockIdx.x * THREADS + threadIdx.x;
  if (idx >= dd) return;
  double s = 0.0;
  for (int64_t c = 0; c < chunks; ++c) s += part[c * dd + idx];        // fixed order: deterministic
  out[idx] = s;
}
}  // namespace moments
}  // namespace clica

using namespace clica;

extern "C" int clica_moments_workspace_bytes(int64_t M, int32_t d, size_t* bytes) {
  CLICA_CHECK_ARG(bytes && M > 0 && d >= 2 && d <= moments::MAXD, "clica_moments_workspace_bytes: M > 0 and 2 <= d <= %d required (got %lld, %d)",
                  moments::MAXD, (long long)M, d);
  *bytes = (size_t)ceil_div(M, moments::ROWS) * d * d * sizeof(double);
  return CLICA_OK;
}

extern "C" int clica_moments(const float* A, int64_t lda, int32_t da, const float* B, int64_t ldb, int32_t db, int64_t M, double* out,
                             void* workspace, size_t workspace_bytes, clica_stream_t stream) {
  CLICA_CHECK_ARG(A && out && workspace && M > 0 && da >= 1 && db >= 0 && lda >= da && (db == 0 || (B && ldb >= db)), "clica_moments: bad argument");
  const int d = da + db + 1;
  CLICA_CHECK_ARG(d <= moments::MAXD, "clica_moments: %d + %d + 1 columns exceed %d", da, db, moments::MAXD);
  const int64_t chunks = ceil_div(M, moments::ROWS);
  if (workspace_bytes < (size_t)chunks * d * d * sizeof(double)) {
    set_error("clica_moments: workspace of %zu bytes, need %zu", workspace_bytes, (size_t)chunks * d * d * sizeof(double));
    return CLICA_E_WORKSPACE;
  }
  double* part = reinterpret_cast<double*>(workspace);
  hipLaunchKernelGGL(moments::partial_k, dim3((unsigned)chunks), dim3(moments::THREADS), 0, as_stream(stream), A, lda, (int)da, B, ldb, (int)db, M, part);
  hipLaunchKernelGGL(moments::reduce_k, dim3((unsigned)ceil_div(d * d, moments::THREADS)), dim3(moments::THREADS), 0, as_stream(stream), part, chunks,
                     d * d, out);
  return launch_status("clica_moments");
}
