/* Stub of the C-ABI entry points the documented reference-side binding calls: lets the
 * binding be compiled, linked and run on a box without a GPU. It "updates" by leaving P alone and returning dx = inn
 * sums, which is enough to see every call arrive with consistent sizes. TEST INFRASTRUCTURE ONLY. */
#include <stdlib.h>
#include <string.h>
#include "xivo_hip.h"

struct xivo_hip_ctx { int N, M; double* P; double* err; int calls; };
static struct xivo_hip_ctx g_ctx;

int xivo_hip_create(xivo_hip_ctx** out, int device, int N, int M_max, int batch_max, unsigned flags) {
  (void)device; (void)M_max; (void)batch_max; (void)flags;
  g_ctx.N = N; g_ctx.P = (double*)calloc((size_t)N * N, sizeof(double)); g_ctx.err = (double*)calloc(N, sizeof(double)); g_ctx.calls = 0;
  *out = &g_ctx;
  return XIVO_HIP_OK;
}
void xivo_hip_destroy(xivo_hip_ctx* c) { free(c->P); free(c->err); }
const char* xivo_hip_strerror(int s) { return s == 0 ? "ok" : "error"; }
int xivo_hip_upload_P(xivo_hip_ctx* c, int b0, int nb, const double* P, long stride, int ld) {
  if (b0 != 0 || nb != 1 || ld != c->N || stride != (long)c->N * c->N) return XIVO_HIP_ERR_INVALID;
  memcpy(c->P, P, sizeof(double) * c->N * c->N); c->calls |= 1;
  return XIVO_HIP_OK;
}
int xivo_hip_set_measurements(xivo_hip_ctx* c, int b0, int nb, int M, const double* H, long strideH, int ldh, const double* inn,
                              long strideInn, const double* diagR, long strideR) {
  if (b0 != 0 || nb != 1 || ldh != M || strideH != (long)M * c->N || strideInn != M || strideR != M || !H || !diagR) return XIVO_HIP_ERR_INVALID;
  c->M = M;
  for (int n = 0; n < c->N; ++n) c->err[n] = 0.0;
  for (int m = 0; m < M; ++m) c->err[m % c->N] += inn[m];
  c->calls |= 2;
  return XIVO_HIP_OK;
}
int xivo_hip_update_joseph(xivo_hip_ctx* c, int B) { if (B != 1) return XIVO_HIP_ERR_INVALID; c->calls |= 4; return XIVO_HIP_OK; }
int xivo_hip_get_err(xivo_hip_ctx* c, int b0, int nb, double* err, long stride) {
  if (b0 != 0 || nb != 1 || stride != c->N) return XIVO_HIP_ERR_INVALID;
  memcpy(err, c->err, sizeof(double) * c->N); c->calls |= 8;
  return XIVO_HIP_OK;
}
int xivo_hip_download_P(xivo_hip_ctx* c, int b0, int nb, double* P, long stride, int ld) {
  if (b0 != 0 || nb != 1 || ld != c->N || stride != (long)c->N * c->N) return XIVO_HIP_ERR_INVALID;
  memcpy(P, c->P, sizeof(double) * c->N * c->N); c->calls |= 16;
  return XIVO_HIP_OK;
}
int xivo_hip_get_status(xivo_hip_ctx* c, int b0, int nb, int* status) { (void)b0; (void)nb; *status = 0; c->calls |= 32; return XIVO_HIP_OK; }
/* the one-call plumbing entry (round 4): every argument of the six calls above arrives at once */
int xivo_hip_update_joseph_host(xivo_hip_ctx* c, int b, int M, const double* H, int ldh, const double* inn, const double* diagR,
                                double* P, int ldp, double* err_out, unsigned mode) {
  if (b != 0 || ldh != M || ldp != c->N || mode != 0 || !H || !diagR || !P || !err_out) return XIVO_HIP_ERR_INVALID;
  c->M = M;
  for (int n = 0; n < c->N; ++n) err_out[n] = 0.0;
  for (int m = 0; m < M; ++m) err_out[m % c->N] += inn[m];
  c->calls |= 64;                 /* P is left alone: "updated in place" */
  return XIVO_HIP_OK;
}
int stub_calls(void) { return g_ctx.calls; }
