// Which GPU instruction sequence reproduces the CPU reference's fp32 dot-product chain  acc = acc + e_k * q_k  (fp16 inputs,
// ascending k, one rounding per step)?  Three device forms against a host loop, on random unit-norm vectors:
//   A  v_fma_mix_f32 with both fp16 sources (inline asm, one wait state after each)
//   B  v_cvt_f32_f16 of both operands, then fp32 fma (what the compiler emits for  fmaf((float)e, (float)q, acc))
//   C  like B but with a compiler barrier on the accumulator after every step
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/chain_probe.hip -o chain_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef _Float16 half_t;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
constexpr int D = 128;

__global__ void k_chain(const uint32_t* e, const uint32_t* q, int n, float* outA, float* outB, float* outC) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* ev = e + (size_t)i * (D / 2);
  const uint32_t* qv = q + (size_t)i * (D / 2);
  float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
  for (int k = 0; k < D / 2; ++k) {
    const uint32_t ew = ev[k], qw = qv[k];
    asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]\n\ts_nop 0\n\t"
                 "v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]\n\ts_nop 0"
                 : "+v"(a) : "v"(ew), "v"(qw));
    const h2 eh = __builtin_bit_cast(h2, ew), qh = __builtin_bit_cast(h2, qw);
    b = __builtin_fmaf((float)eh.x, (float)qh.x, b);
    b = __builtin_fmaf((float)eh.y, (float)qh.y, b);
    c = __builtin_fmaf((float)eh.x, (float)qh.x, c);
    asm volatile("" : "+v"(c));
    c = __builtin_fmaf((float)eh.y, (float)qh.y, c);
    asm volatile("" : "+v"(c));
  }
  outA[i] = a; outB[i] = b; outC[i] = c;
}

int main() {
  const int n = 1 << 18;
  std::vector<half_t> e((size_t)n * D), q((size_t)n * D);
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
  for (int i = 0; i < n; ++i) {
    double ne = 0, nq = 0;
    std::vector<double> te(D), tq(D);
    for (int k = 0; k < D; ++k) {
      te[k] = rnd() + rnd() + rnd() + rnd() - 2.0;
      tq[k] = 0.7 * te[k] + (rnd() + rnd() + rnd() + rnd() - 2.0);
      if (k % 37 == 5) te[k] *= 1e-3;   // some tiny (fp16-denormal) components
      ne += te[k] * te[k]; nq += tq[k] * tq[k];
    }
    for (int k = 0; k < D; ++k) { e[(size_t)i * D + k] = (half_t)(te[k] / std::sqrt(ne)); q[(size_t)i * D + k] = (half_t)(tq[k] / std::sqrt(nq)); }
  }
  std::vector<float> ref(n);
  for (int i = 0; i < n; ++i) {
    volatile float acc = 0.f;
    for (int k = 0; k < D; ++k) { const float p = (float)e[(size_t)i * D + k] * (float)q[(size_t)i * D + k]; acc = acc + p; }   // product exact (22 bits)
    ref[i] = acc;
  }
  uint32_t *de, *dq; float *dA, *dB, *dC;
  hipMalloc(&de, (size_t)n * D * 2); hipMalloc(&dq, (size_t)n * D * 2);
  hipMalloc(&dA, n * 4); hipMalloc(&dB, n * 4); hipMalloc(&dC, n * 4);
  hipMemcpy(de, e.data(), (size_t)n * D * 2, hipMemcpyHostToDevice);
  hipMemcpy(dq, q.data(), (size_t)n * D * 2, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_chain, dim3((n + 255) / 256), dim3(256), 0, 0, de, dq, n, dA, dB, dC);
  std::vector<float> A(n), B(n), C(n);
  hipMemcpy(A.data(), dA, n * 4, hipMemcpyDeviceToHost); hipMemcpy(B.data(), dB, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(C.data(), dC, n * 4, hipMemcpyDeviceToHost);
  long ma = 0, mb = 0, mc = 0, mab = 0, h16a = 0, h16b = 0;
  for (int i = 0; i < n; ++i) {
    ma += A[i] != ref[i]; mb += B[i] != ref[i]; mc += C[i] != ref[i]; mab += A[i] != B[i];
    h16a += (half_t)A[i] != (half_t)ref[i]; h16b += (half_t)B[i] != (half_t)ref[i];
  }
  std::printf("vectors %d: fp32 mismatches vs host chain: mix %ld, cvt+fma %ld, cvt+fma(barrier) %ld; mix vs cvt+fma %ld; after fp16 rounding: mix %ld, cvt+fma %ld\n",
              n, ma, mb, mc, mab, h16a, h16b);
  int shown = 0;
  for (int i = 0; i < n && shown < 4; ++i)
    if (B[i] != ref[i] || A[i] != ref[i]) { std::printf("  i=%d ref %.9g mix %.9g cvt+fma %.9g\n", i, ref[i], A[i], B[i]); ++shown; }
  return 0;
}
