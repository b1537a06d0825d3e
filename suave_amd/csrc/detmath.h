// Deterministic binary32 math for discrete sampling decisions.
//
// exp / log / softplus / sigmoid written with individually rounded + - * /
// only (no fused multiply-add, no hardware transcendental), so that the values
// feeding an argmax / threshold compare are bit-identical to the CPU oracle's
// numpy float32 evaluation of the same operation sequence.  The polynomial
// kernels are the classic fdlibm expf / logf ones.
#pragma once
#include <hip/hip_runtime.h>

namespace detm {

// NOTE: HIP's __fmul_rn / __fadd_rn are plain operators and are contracted into FMAs under the
// default -ffp-contract=fast-honor-pragmas, and __fsqrt_rn is the 1-ulp v_sqrt_f32.  Contraction is
// therefore switched off for everything in this header, translation units that use it are built
// with -ffp-contract=off (build.sh), and sqrt / divide go through the correctly rounded default
// lowering of sqrtf() and operator/.
#pragma clang fp contract(off)
__device__ __forceinline__ float mul(float a, float b) { return a * b; }
__device__ __forceinline__ float add(float a, float b) { return a + b; }
__device__ __forceinline__ float sub(float a, float b) { return a - b; }
__device__ __forceinline__ float dvd(float a, float b) { return a / b; }
__device__ __forceinline__ float sqrt_rn(float a) { return sqrtf(a); }

__device__ __forceinline__ float exp(float x) {
  const float LN2_HI = 6.9313812256e-01f, LN2_LO = 9.0580006145e-06f, INV_LN2 = 1.4426950216e+00f;
  const float P1 = 1.6666625440e-01f, P2 = -2.7667332906e-03f;
  float xc = fminf(fmaxf(x, -87.0f), 88.0f);
  float n = __builtin_rintf(mul(xc, INV_LN2));
  float hi = sub(xc, mul(n, LN2_HI));
  float lo = mul(n, LN2_LO);
  float r = sub(hi, lo);
  float t = mul(r, r);
  float c = sub(r, mul(t, add(P1, mul(t, P2))));
  float y = sub(1.0f, sub(sub(lo, dvd(mul(r, c), sub(2.0f, c))), hi));
  int bits = __float_as_int(y) + static_cast<int>(static_cast<unsigned>(static_cast<int>(n)) << 23);
  float out = __int_as_float(bits);
  return x < -87.0f ? 0.0f : out;
}

__device__ __forceinline__ float log(float x) {
  const float LN2_HI = 6.9313812256e-01f, LN2_LO = 9.0580006145e-06f;
  const float LG1 = 0.66666662693f, LG2 = 0.40000972152f, LG3 = 0.28498786688f, LG4 = 0.24279078841f;
  const float SQRT2 = 1.41421353817f;
  x = fmaxf(x, 1.17549435e-38f);
  int ix = __float_as_int(x);
  int e = (ix >> 23) - 127;
  float m = __int_as_float((ix & 0x007FFFFF) | 0x3F800000);
  bool big = m > SQRT2;
  m = big ? mul(m, 0.5f) : m;
  float ef = static_cast<float>(e + (big ? 1 : 0));
  float f = sub(m, 1.0f);
  float s = dvd(f, add(2.0f, f));
  float z = mul(s, s);
  float w = mul(z, z);
  float t1 = mul(w, add(LG2, mul(w, LG4)));
  float t2 = mul(z, add(LG1, mul(w, LG3)));
  float R = add(t2, t1);
  float hfsq = mul(mul(0.5f, f), f);
  float lm = sub(f, sub(hfsq, mul(s, add(hfsq, R))));
  return add(mul(ef, LN2_HI), add(lm, mul(ef, LN2_LO)));
}

__device__ __forceinline__ float softplus(float x) {
  float sp = log(add(1.0f, exp(fminf(x, 20.0f))));
  return x > 20.0f ? x : sp;
}

__device__ __forceinline__ float sigmoid(float x) { return dvd(1.0f, add(1.0f, exp(-x))); }

}  // namespace detm
