/* c_rollout.c -- the C ABI from plain C, no Python anywhere: MJCF file -> sg_model_compile -> a batch of envs -> the reference's
 * squeeze schedule (create_dataset.py:41-56: 40 idle env steps, close at -0.2, toggle to +0.2 after 80 more) -> sensor rows.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_rollout.c -o c_rollout \
 *       -L soft-grip_amd -lsoftgrip -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/soft-grip_amd -Wl,-rpath,/opt/rocm/lib
 *   ./c_rollout tests/data/mini_gripper.xml 8 [flags]
 *
 * Stiffness: env e of n gets k = 300 + 1100 e / (n - 1) on every composite slider (the model's last nelem dofs) and on tendon 0 --
 * the per-env randomisation of manenv.py:103-109 with a deterministic draw.  Prints, per env step, the 12 sensor values of the
 * first and the last env (%.17g) and at the end the number of env steps that raised a flag.  tests/test_gpu_parity.py builds and
 * runs it and compares the rows with the Python host's on the same inputs. */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "softgrip.h"

#define SG(call)                                                                  \
  do {                                                                            \
    int rc_ = (call);                                                             \
    if (rc_ != SG_OK) { fprintf(stderr, "%s: %d: %s\n", #call, rc_, sg_last_error()); return 1; } \
  } while (0)
#define HIP(call)                                                                 \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s scene.xml n_envs [compile flags]\n", argv[0]); return 2; }
  const int n = atoi(argv[2]), flags = argc > 3 ? atoi(argv[3]) : 0;
  sg_model* m = NULL;
  sg_batch* b = NULL;
  SG(sg_model_compile(argv[1], flags, &m));
  const int nq = sg_model_nq(m), nelem = sg_model_nelem(m), ns = sg_model_nsensordata(m), nu = sg_model_nu(m);
  SG(sg_batch_create(m, n, 0, &b));
  double* k = (double*)malloc(sizeof(double) * n);
  int* ids = (int*)malloc(sizeof(int) * nelem);
  int tid = 0;
  for (int e = 0; e < n; e++) k[e] = n > 1 ? 300.0 + 1100.0 * e / (n - 1) : 700.0;
  for (int i = 0; i < nelem; i++) ids[i] = nq - nelem + i;
  SG(sg_set_stiffness(b, k, 1, ids, nelem, &tid, 1, NULL));
  double* sens = NULL;
  int32_t *fl = NULL, *touch = NULL;
  HIP(hipMalloc((void**)&sens, sizeof(double) * n * ns));
  HIP(hipMalloc((void**)&fl, sizeof(int32_t) * n));
  HIP(hipMalloc((void**)&touch, sizeof(int32_t) * n));
  double* hs = (double*)malloc(sizeof(double) * n * ns);
  int32_t* hf = (int32_t*)malloc(sizeof(int32_t) * n);
  SG(sg_reset(b, NULL, 1, sens, fl, touch, NULL));
  double ctrl[8] = {0};
  long flagged = 0;
  for (int t = 0; t < 200; t++) {
    if (t == 40 || t == 120) {
      for (int u = 0; u < nu && u < 8; u++) ctrl[u] = t == 40 ? -0.2 : 0.2;
      SG(sg_set_ctrl(b, ctrl, 1, NULL));
    }
    SG(sg_step(b, 7, sens, 0, fl, touch, NULL));
    HIP(hipMemcpy(hs, sens, sizeof(double) * n * ns, hipMemcpyDeviceToHost));
    HIP(hipMemcpy(hf, fl, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    for (int e = 0; e < n; e++) flagged += hf[e] != 0;
    printf("%d", t);
    for (int i = 0; i < ns; i++) printf(" %.17g", hs[i]);
    for (int i = 0; i < ns; i++) printf(" %.17g", hs[(size_t)(n - 1) * ns + i]);
    printf("\n");
  }
  printf("flagged %ld\n", flagged);
  sg_batch_destroy(b);
  sg_model_destroy(m);
  (void)hipFree(sens); (void)hipFree(fl); (void)hipFree(touch);
  free(k); free(ids); free(hs); free(hf);
  return 0;
}
