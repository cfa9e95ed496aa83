// TEST-ONLY diagnostic: evaluate the device's exp/log10/pow/sqrt/div on arrays so that the test
// suite can count how often they differ from the host libm the reference links (glibc).
#include <hip/hip_runtime.h>
#include <math.h>
#ifdef C2R_PROBE_USE_PRODUCT_MATH
#include "../c2-ray3dm1d_helium_amd/csrc/c2ray_shell.hpp"
#else
#define C2R_MATH_EXP(x) exp(x)
#define C2R_MATH_LOG10(x) log10(x)
#define C2R_MATH_POW(x, y) pow(x, y)
#endif
__global__ void k_probe(int op, int n, const double *x, const double *y, double *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r;
  switch (op) {
    case 0: r = C2R_MATH_EXP(x[i]); break;
    case 1: r = C2R_MATH_LOG10(x[i]); break;
    case 2: r = C2R_MATH_POW(x[i], y[i]); break;
    case 3: r = sqrt(x[i]); break;
#ifdef C2R_PROBE_USE_PRODUCT_MATH
    case 13: r = c2r::weightf_fast(x[i], y[i]); break;             // must equal 1 / max(0.6, x * y)
    case 5: r = c2r::recip_nr(x[i]); break;                       // must equal 1.0 / x
    case 6: case 7: case 8: {                                      // div_by_vol: must equal a / b, all three
      const c2r::Recip R = c2r::make_recip(y[i]);
      const double a[3] = {x[i], x[i] * 0x1p-30, 0.0};
      double d[3];
      c2r::div_by_vol<3>(R, a, d);
      r = d[op - 6];
      break;
    }
    case 9: case 10: {                                             // the two-argument table position
      c2r::TauPos pa, pb;
      c2r::tau_table_positions(x[i], y[i], c2r::gm::log_table(), pa, pb);
      const c2r::TauPos p = op == 9 ? pa : pb;
      r = (double)p.ipos + p.residual * 0.5;                       // residual in [0,1): both recoverable
      break;
    }
    case 11: {
      const c2r::TauPos p = c2r::tau_table_position(x[i]);
      r = (double)p.ipos + p.residual * 0.5;
      break;
    }
#endif
    default: r = x[i] / y[i]; break;
  }
  out[i] = r;
}
extern "C" int probe_math(int op, int n, const double *x, const double *y, double *out) {
  double *dx, *dy, *dout;
  size_t b = sizeof(double) * n;
  if (hipMalloc(&dx, b) || hipMalloc(&dy, b) || hipMalloc(&dout, b)) return 1;
  hipMemcpy(dx, x, b, hipMemcpyHostToDevice);
  hipMemcpy(dy, y, b, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_probe, dim3((n + 255) / 256), dim3(256), 0, 0, op, n, dx, dy, dout);
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  hipMemcpy(out, dout, b, hipMemcpyDeviceToHost);
  hipFree(dx); hipFree(dy); hipFree(dout);
  return 0;
}
