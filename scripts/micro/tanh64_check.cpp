// Host replica of mlp64_tanh (csrc/mi_ode_mlp64.h): max relative error against libm tanh.  g++ -O2 -ffp-contract=off tanh64_check.cpp && ./a.out
#include <cmath>
#include <cstdio>
#include <cstdlib>
static inline double rcp_approx(double s) { return (double)(float)(1.0 / s) * (1.0 + 3e-8); }   // a hardware reciprocal of ~2^-24 accuracy
static inline double my_exp2x(double x) {      // e^(2x)
  const double a = 2.0 * x;
  double n = std::nearbyint(a * 1.4426950408889634);
  if (n > 1023) n = 1023; if (n < -1074) n = -1074;
  double r = std::fma(-n, 0.6931471805599453, a);          // hi
  r = std::fma(-n, 2.3190468138462996e-17, r);             // lo
  double p = 2.08767569878681e-09;                          // 1/12!
  p = std::fma(p, r, 2.505210838544172e-08);                // 1/11!
  p = std::fma(p, r, 2.755731922398589e-07);
  p = std::fma(p, r, 2.7557319223985893e-06);
  p = std::fma(p, r, 2.48015873015873e-05);
  p = std::fma(p, r, 0.0001984126984126984);
  p = std::fma(p, r, 0.001388888888888889);
  p = std::fma(p, r, 0.008333333333333333);
  p = std::fma(p, r, 0.041666666666666664);
  p = std::fma(p, r, 0.16666666666666666);
  p = std::fma(p, r, 0.5);
  p = std::fma(p, r, 1.0);
  p = std::fma(p, r, 1.0);
  return std::ldexp(p, (int)n);
}
static inline double my_tanh(double x) {
  const double xc = x > 20.0 ? 20.0 : (x < -20.0 ? -20.0 : x);
  const double e = my_exp2x(xc);
  const double s = e + 1.0;
  double r = rcp_approx(s);
  r = std::fma(std::fma(-s, r, 1.0), r, r);
  r = std::fma(std::fma(-s, r, 1.0), r, r);
  const double big = std::fma(-2.0, r, 1.0);
  const double x2 = x * x;
  const double small = x * (1.0 + x2 * (-1.0 / 3.0 + x2 * (2.0 / 15.0 + x2 * (-17.0 / 315.0 + x2 * (62.0 / 2835.0 + x2 * (-1382.0 / 155925.0))))));
  double out = std::fabs(x) < 0.0625 ? small : big;
  return x != x ? x : out;
}
int main() {
  double worst = 0, wx = 0; srand(1);
  for (int i = 0; i < 20000000; ++i) {
    double u = (double)rand() / RAND_MAX;
    double x = (i % 3 == 0) ? (u - 0.5) * 0.4 : (i % 3 == 1 ? (u - 0.5) * 6 : (u - 0.5) * 50);
    double a = my_tanh(x), b = std::tanh(x);
    double err = std::fabs(a - b) / (std::fabs(b) > 1e-300 ? std::fabs(b) : 1e-300);
    if (err > worst) { worst = err; wx = x; }
  }
  printf("max rel err %.3e at x = %.6f  (eps = 2.2e-16)\n", worst, wx);
  printf("tanh(1000) %g tanh(-1000) %g tanh(0) %g\n", my_tanh(1000), my_tanh(-1000), my_tanh(0.0));
}
