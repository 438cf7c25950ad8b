// Device-side helpers shared by the HIP translation units (fp_kernels.hip, fp_maxsim.hip).  gfx950 only.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define NEG_MASK_F (-10000.0f)  // masked_fill(-9999.0) stored in fp16 (search.rs:395)

__device__ __forceinline__ uint32_t h2_as_u32(h2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ h2 u32_as_h2(uint32_t v) { return __builtin_bit_cast(h2, v); }
__device__ __forceinline__ h2 pk_max(h2 a, h2 b) { return __builtin_elementwise_max(a, b); }
// The bare instructions: __builtin_elementwise_max / min first canonicalise both operands (v_pk_max_f16 x, x, x) because the
// kernels run in IEEE mode, tripling the count.  Only for values that cannot be signalling NaNs (everything the kernels produce).
__device__ __forceinline__ uint32_t pk_max_raw(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ uint32_t pk_min_raw(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_pk_min_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// floor of both halves in two instructions: the second one writes the upper half only (the compiler's form is floor, floor
// into another register, pack).  The trailing s_nop covers the wait state a sub-dword write needs before a VALU reads it.
__device__ __forceinline__ uint32_t pk_floor_raw(uint32_t y) {
  uint32_t r;
  asm("v_floor_f16_e32 %0, %1\n\tv_floor_f16_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n\ts_nop 0"
      : "=&v"(r)
      : "v"(y));
  return r;
}
// fp16 rounding mode of the wave (MODE.fp_round[3:2], shared with fp64): 0 nearest-even, 2 toward -inf.  Volatile, like the
// instructions that are meant to run under it (pk_fma_rd_raw): the compiler keeps volatile asm statements in program order.
__device__ __forceinline__ void f16_round_down() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 2\n\ts_nop 1"); }
__device__ __forceinline__ void f16_round_nearest() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0\n\ts_nop 1"); }
// a * b + c on both halves, ONE rounding, in the wave's current fp16 rounding mode
__device__ __forceinline__ uint32_t pk_fma_mode_raw(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t r;
  asm volatile("v_pk_fma_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_add_u16(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ uint32_t shfl_xor_u32(uint32_t v, int m) { return (uint32_t)__shfl_xor((int)v, m, 64); }

// order-preserving maps (larger float <-> larger unsigned)
__device__ __forceinline__ uint32_t mono16(uint16_t h) {
  if ((h & 0x7FFF) == 0) h = 0;  // -0 == +0
  return (h & 0x8000) ? (uint32_t)(uint16_t)~h : (uint32_t)(h | 0x8000);
}
__device__ __forceinline__ uint32_t mono32(float f) {
  uint32_t b = __float_as_uint(f + 0.0f);  // -0 -> +0
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float unmono32(uint32_t k) {
  uint32_t b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(b);
}

// two-term reciprocal: r_hi + r_lo = 1/n to ~2^-47; specials (n = 0, inf, nan) keep IEEE semantics
__device__ __forceinline__ void recip2(float n, float& r_hi, float& r_lo) {
  r_hi = 1.0f / n;
  const float t = __builtin_fmaf(-n, r_hi, 1.0f) * r_hi;
  const bool ok = (r_hi != 0.0f) && (__builtin_fabsf(r_hi) < __builtin_inff());  // false for nan too
  r_lo = ok ? t : 0.0f;
}
__device__ __forceinline__ float quot2(float e, float r_hi, float r_lo) { return __builtin_fmaf(e, r_hi, e * r_lo); }
// h(fl32(fma(e, r_hi, e*r_lo))) for two packed fp16 pairs in 10 VALU instructions (2.5 per
// element, no separate fp16->fp32 conversions: v_fma_mix_f32 takes fp16 sources in place).
// The result is rounded to fp32 FIRST and then to fp16 by v_cvt_pk_f16_f32, exactly like the
// reference's h(fl32(e/n)); v_fma_mixlo/hi_f16 would round once and disagree on exact
// subnormal ties (measured: 5,626 of 2^32 pairs).  Every consumer sits >= 2 instructions after
// its producer (mix -> dependent op needs one wait state; hipcc does not pad inside asm).
// fp_selftest_arith runs THIS function over all 2^32 (e, n) pairs.
__device__ __forceinline__ void norm_pair2(uint32_t& a, uint32_t& b, float r_hi, float r_lo) {
  uint32_t da, db;
  float t0, t1, t2, t3;
  asm("v_fma_mix_f32 %2, %6, %9, 0 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %3, %6, %9, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %4, %7, %9, 0 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %5, %7, %9, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %2, %6, %8, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %3, %6, %8, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %4, %7, %8, %4 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %5, %7, %8, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
      "v_cvt_pk_f16_f32 %1, %4, %5"
      : "=&v"(da), "=&v"(db), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(a), "v"(b), "v"(r_hi), "v"(r_lo));
  a = da;
  b = db;
}

// h(fl32(e * r)) for two packed fp16 pairs: the one-multiply normalisation of k_maxsim6.  r is the token's stored reciprocal
// (k_token_rinv picked it so that the result equals norm_pair2's for EVERY dim of the token, i.e. the reference's e^ bits).
__device__ __forceinline__ void norm_mul2(uint32_t& a, uint32_t& b, float r) {
  uint32_t da, db;
  float t0, t1, t2, t3;
  asm("v_fma_mix_f32 %2, %6, %8, 0 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %3, %6, %8, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %4, %7, %8, 0 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mix_f32 %5, %7, %8, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
      "v_cvt_pk_f16_f32 %1, %4, %5"
      : "=&v"(da), "=&v"(db), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(a), "v"(b), "v"(r));
  a = da;
  b = db;
}
// max(a, b, c) without the canonicalisation __builtin_fmaxf adds in IEEE mode (a NaN operand is ignored, as by fmaxf)
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// More than 64 KiB of dynamic LDS needs an opt-in per kernel AND per device (gfx950: 160 KiB per workgroup); one process
// may drive several GPUs (FastPlaid(device=[...])), so the "done" flag is a bit per device.
static inline void fp_allow_big_lds(const void* fn, std::atomic<uint64_t>& done, int bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_relaxed) & bit)) {
    // the limit covers static + dynamic LDS: a kernel with static __shared__ variables can only be granted the remainder
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, fn) == hipSuccess) {
      const int room = 160 * 1024 - (int)fa.sharedSizeBytes;
      if (bytes > room) bytes = room;
    }
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
      (void)hipGetLastError();   // the launch that needs the room reports the failure itself (stage-boundary check)
    done.fetch_or(bit, std::memory_order_relaxed);
  }
}


#if defined(__HIPCC__)
// Inclusive prefix sum over a whole workgroup (blockDim.x = 64 * nwaves <= 1024, every thread calls it): inside a wave by lane
// shuffles, across the waves through their totals in LDS (`wt`, [16]).  Two barriers -- the shared-memory Hillis-Steele loops it
// replaces were two per STEP (twenty for 1024 threads), and the kernels that scan are one-shot kernels whose time is such chains.
// *total (optional) receives the sum over the workgroup.  `wt` may be reused as soon as the call returns.
template <typename T>
__device__ __forceinline__ T fp_block_scan_incl(T v, T* wt, T* total = nullptr) {
  const int lane = (int)threadIdx.x & 63, wave = (int)threadIdx.x >> 6, nw = ((int)blockDim.x + 63) >> 6;
  T incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const T y = __shfl_up(incl, d, 64);
    if (lane >= d) incl += y;
  }
  if (lane == 63) wt[wave] = incl;
  __syncthreads();
  T before = 0, tot = 0;
  for (int w = 0; w < nw; ++w) {
    const T t = wt[w];
    before += w < wave ? t : (T)0;
    tot += t;
  }
  __syncthreads();
  if (total) *total = tot;
  return before + incl;
}

// Bitonic sort of 1024 64-bit keys, one per thread of a 1024-thread workgroup, descending: returns the key of rank threadIdx.x.
// Exchanges inside a wave go through lane shuffles (45 of the 55 stages: no barrier), the ten stages with partners in other
// waves through `lds` ([1024] keys).  Same network as the all-LDS loops it replaces (k_final_topk / k_final_mark /
// k_sel_finish took ~20 us for 1024 keys, most of it barriers).
__device__ __forceinline__ unsigned long long fp_sort1024_desc(unsigned long long key, unsigned long long* lds) {
  const int i = (int)threadIdx.x;
  for (int k = 2; k <= 1024; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      unsigned long long other;
      if (j >= 64) {
        lds[i] = key;
        __syncthreads();
        other = lds[i ^ j];
        __syncthreads();
      } else {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)key, j, 64);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(key >> 32), j, 64);
        other = ((unsigned long long)hi << 32) | lo;
      }
      const bool desc = ((i & k) == 0);
      const bool lower = ((i & j) == 0);
      const bool take_max = (lower == desc);
      key = take_max ? (key > other ? key : other) : (key < other ? key : other);
    }
  }
  return key;
}
#endif

static inline int fp_next_pow2(int x) {
  int p = 1;
  while (p < x) p <<= 1;
  return p;
}
