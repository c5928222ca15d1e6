/* The drop-in boundary is a C ABI: this translation unit is plain C99 (gcc -std=c99 -pedantic -Werror), includes
 * include/lanczos_hip.h and links liblanczos_hip.so without any C++ or HIP on the caller's side.  Without a device the
 * library must refuse loudly (no CPU fallback); with one it runs the reference's README sample through a host callback
 * (src/samples/sample1_simple.cpp:22-28: 3x3 matrix, largest eigenvalue 4). */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "lanczos_hip.h"

static int mv_mul(const double* in, double* out, int64_t n, void* user) {
  static const double m[3][3] = {{2, 1, 1}, {1, 2, 1}, {1, 1, 2}};
  int64_t i, j;
  (void)user;
  for (i = 0; i < n; ++i)
    for (j = 0; j < n; ++j) out[i] += m[i][j] * in[j];
  return 0;
}

static void init(void* vec, int64_t n_local, int64_t row_begin, void* user) {
  double* v = (double*)vec;
  int64_t i;
  (void)user;
  for (i = 0; i < n_local; ++i) v[i] = 1.0 + 0.5 * (double)(row_begin + i);
}

int main(void) {
  ll_context* ctx = NULL;
  ll_operator* op = NULL;
  ll_lanczos_params p;
  ll_expo_params e;
  ll_stencil_desc d;
  ll_run_stats st;
  double value = 0.0, vec[3];
  int64_t found = 0, counts[8];
  int rc;
  memset(&d, 0, sizeof(d));
  if (ll_lanczos_params_default(&p, 3, 1, 1) != LL_OK || ll_expo_params_default(&e, 3) != LL_OK) return 10;
  rc = ll_ctx_create(0, &ctx);
  if (rc != LL_OK) {
    printf("no device: rc=%d: %s\n", rc, ll_last_error());
    return strstr(ll_last_error(), "no CPU fallback") ? 2 : 11;
  }
  p.init_vector = init;
  if (ll_op_create_host_d(ctx, 3, mv_mul, NULL, &op) != LL_OK) return 12;
  if (ll_lanczos_run_d(ctx, op, &p, &value, vec, &found, counts, 8, NULL, NULL, &st) != LL_OK) {
    printf("run failed: %s\n", ll_last_error());
    return 13;
  }
  printf("lambda_max = %.15f after %lld iterations\n", value, (long long)counts[0]);
  if (found != 1 || fabs(value - 4.0) > 1e-12) return 14;
  ll_op_destroy(op);
  ll_ctx_destroy(ctx);
  return 0;
}
