// Internal (not part of the C ABI): the fp16-input tile core of wide16.hip as a plain "similarity block" service for other translation units.
#pragma once
#include <hip/hip_runtime.h>

// out[m][n] = sum_k A[m][k] B[n][k]   (A [M][lda], B [N][ldb] fp16 with k contiguous, K % 8 == 0; out fp32 [M][ldo], plainly stored)
struct SgaW16Store { const void* A; long lda; int M; const void* B; long ldb; int N; int K; float* out; long ldo; };
// up to 8 such products in one launch
int sga_wide16_store_batch(const SgaW16Store* e, int n, hipStream_t stream);
