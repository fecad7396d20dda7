// Microbenchmark: what wave64 VALU issue rate does gfx950 actually sustain for the instruction
// mix of k_render's inner loops (v_mul_f32 / v_add_f32 / v_sub_f32, no FMA, SGPR operands)?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MODE 0: independent mul/add on VGPRs only (8 accumulators)
// MODE 1: the primary sphere test body on SGPR operands, records from a uniform pointer (s_load), no branch
// MODE 2: MODE 1 + the max/cmp/branch filter (never taken)
// MODE 3: same as 2 but table in LDS (ds_read_b128 broadcast)
template <int MODE>
__global__ void __launch_bounds__(256) k(const float4 *__restrict__ tab, int n, int iters, float *out,
                                         float dx, float dy, float dz) {
  __shared__ float4 lds[2048];
  float acc = 0.f;
  float x = dx + threadIdx.x * 1e-6f, y = dy, z = dz;
  if (MODE == 3) {
    for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = tab[i];
    __syncthreads();
  }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      float a0 = x, a1 = y, a2 = z, a3 = x + 1.f, a4 = y + 1.f, a5 = z + 1.f, a6 = x + 2.f, a7 = y + 2.f;
      for (int k = 0; k < n; k += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          a0 = a0 * x; a1 = a1 + y; a2 = a2 * z; a3 = a3 + x;
          a4 = a4 * y; a5 = a5 + z; a6 = a6 * x; a7 = a7 + y;
        }
      }
      acc += a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    } else {
      for (int k = 0; k < n; k += 4) {
        float4 s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] = (MODE == 3) ? lds[(k + u) & 2047] : tab[k + u];
        float q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float b = (s[u].x * x + s[u].y * y) + s[u].z * z;
          q[u] = b * b - s[u].w;
        }
        if (MODE >= 2) {
          float m = fmaxf(fmaxf(q[0], q[1]), fmaxf(q[2], q[3]));
          if (__builtin_amdgcn_ballot_w64(!(m < 0.f))) acc += sqrtf(q[0]) + sqrtf(q[1]) + sqrtf(q[2]) + sqrtf(q[3]);
        } else {
          acc += q[0]; acc += q[1]; acc += q[2]; acc += q[3];
        }
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

typedef float v2f __attribute__((ext_vector_type(2)));

// MODE 4: packed fp32 (v_pk_mul_f32 / v_pk_add_f32), 8 independent float2 accumulators
__global__ void __launch_bounds__(256) kpk(int n, int iters, float *out, float dx, float dy) {
  float acc = 0.f;
  v2f x = {dx + threadIdx.x * 1e-6f, dx * 0.5f}, y = {dy, dy * 2.f};
  for (int it = 0; it < iters; ++it) {
    v2f a0 = x, a1 = y, a2 = x + 1.f, a3 = y + 1.f, a4 = x + 2.f, a5 = y + 2.f, a6 = x + 3.f, a7 = y + 3.f;
    for (int k = 0; k < n; k += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = a0 * x; a1 = a1 + y; a2 = a2 * y; a3 = a3 + x;
        a4 = a4 * y; a5 = a5 + x; a6 = a6 * x; a7 = a7 + y;
      }
    }
    v2f t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    acc += t.x + t.y;
  }
  if (acc == 12345.678f) out[0] = acc;
}

int run_pk(float *d_out, int n, int iters, int blocks) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kpk, dim3(blocks), dim3(256), 0, 0, n, 1, d_out, 0.3f, 0.5f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kpk, dim3(blocks), dim3(256), 0, 0, n, iters, d_out, 0.3f, 0.5f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double wave_instr = (double)blocks * 4 * iters * (n / 4.0) * 32;
  double per = wave_instr / 1024.0 / (ms * 1e3);
  printf("%-28s blocks %6d  %8.3f ms  %6.1f PK wave-instr/us/SIMD  -> %.2f cycles/instr @2.4GHz (%.1f Tlane-op/s, 2 ops per pk lane)\n",
         "mode4 v_pk mul/add x32", blocks, ms, per, 2400.0 / per, wave_instr * 128 / (ms * 1e-3) / 1e12);
  return 0;
}

// MODE 5: the primary sphere test for 2 pixels per lane, 4 spheres per block, hand-written:
// packed fp32 with the sphere constants read straight from SGPR pairs through op_sel (no moves),
// the four independent chains interleaved so no v_pk result is consumed by the next instruction.
__global__ void __launch_bounds__(256) kasm(const float4 *__restrict__ tab, int n, int iters, float *out,
                                            float dx, float dy, float dz) {
  float acc = 0.f;
  v2f x = {dx + threadIdx.x * 1e-6f, dx * 0.5f}, y = {dy, dy * 2.f}, z = {dz, dz * 0.25f};
  for (int it = 0; it < iters; ++it) {
    for (int k = 0; k < n; k += 4) {
      float4 s0 = tab[k], s1 = tab[k + 1], s2 = tab[k + 2], s3 = tab[k + 3];
      v2f q0, q1, q2, q3, t0, t1, t2, t3;
      float m;
      asm volatile(
          "v_pk_mul_f32 %0, %[s0xy], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
          "v_pk_mul_f32 %1, %[s1xy], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
          "v_pk_mul_f32 %2, %[s2xy], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
          "v_pk_mul_f32 %3, %[s3xy], %[x] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
          "v_pk_mul_f32 %4, %[s0xy], %[y] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
          "v_pk_mul_f32 %5, %[s1xy], %[y] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
          "v_pk_mul_f32 %6, %[s2xy], %[y] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
          "v_pk_mul_f32 %7, %[s3xy], %[y] op_sel:[1,0] op_sel_hi:[1,1]\n\t"
          "v_pk_add_f32 %0, %0, %4\n\t"
          "v_pk_add_f32 %1, %1, %5\n\t"
          "v_pk_add_f32 %2, %2, %6\n\t"
          "v_pk_add_f32 %3, %3, %7\n\t"
          "v_pk_mul_f32 %4, %[s0zw], %[z] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
          "v_pk_mul_f32 %5, %[s1zw], %[z] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
          "v_pk_mul_f32 %6, %[s2zw], %[z] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
          "v_pk_mul_f32 %7, %[s3zw], %[z] op_sel:[0,0] op_sel_hi:[0,1]\n\t"
          "v_pk_add_f32 %0, %0, %4\n\t"
          "v_pk_add_f32 %1, %1, %5\n\t"
          "v_pk_add_f32 %2, %2, %6\n\t"
          "v_pk_add_f32 %3, %3, %7\n\t"
          "v_pk_mul_f32 %4, %0, %0\n\t"
          "v_pk_mul_f32 %5, %1, %1\n\t"
          "v_pk_mul_f32 %6, %2, %2\n\t"
          "v_pk_mul_f32 %7, %3, %3\n\t"
          "v_pk_add_f32 %4, %4, %[s0zw] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 %5, %5, %[s1zw] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 %6, %6, %[s2zw] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          "v_pk_add_f32 %7, %7, %[s3zw] op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
          : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
          : [x] "v"(x), [y] "v"(y), [z] "v"(z),
            [s0xy] "s"(*(const v2f *)&s0.x), [s0zw] "s"(*(const v2f *)&s0.z),
            [s1xy] "s"(*(const v2f *)&s1.x), [s1zw] "s"(*(const v2f *)&s1.z),
            [s2xy] "s"(*(const v2f *)&s2.x), [s2zw] "s"(*(const v2f *)&s2.z),
            [s3xy] "s"(*(const v2f *)&s3.x), [s3zw] "s"(*(const v2f *)&s3.z));
      m = fmaxf(fmaxf(fmaxf(q0.x, q0.y), fmaxf(q1.x, q1.y)), fmaxf(fmaxf(q2.x, q2.y), fmaxf(q3.x, q3.y)));
      if (__builtin_amdgcn_ballot_w64(!(m < 0.f)))
        acc += sqrtf(q0.x) + sqrtf(q1.y) + sqrtf(q2.x) + sqrtf(q3.y) + t0.x + t1.y + t2.x + t3.y;
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

int run_asm(const float4 *d_tab, float *d_out, int n, int iters, int blocks) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kasm, dim3(blocks), dim3(256), 0, 0, d_tab, n, 1, d_out, 0.3f, 0.5f, -0.8f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kasm, dim3(blocks), dim3(256), 0, 0, d_tab, n, iters, d_out, 0.3f, 0.5f, -0.8f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double tests = (double)blocks * 4 * iters * n * 128.0;  // pixel-tests
  printf("%-28s blocks %6d  %8.3f ms  %.1f Gtests/s (pixel x sphere)\n", "mode5 asm pk smem px2", blocks, ms, tests / (ms * 1e-3) / 1e9);
  return 0;
}

// MODE 6/7: does a v_pk op with two VGPR-pair sources pay for VGPR bank conflicts?  16 independent
// v_pk_mul_f32 per block with explicit registers: sources in the SAME banks (v[8:9] x v[12:13],
// both pairs start at bank 0) versus in DIFFERENT banks (v[8:9] x v[14:15]).
template <int CONFLICT>
__global__ void __launch_bounds__(256) kbank(int n, float *out) {
  float acc = 0.f;
  for (int k = 0; k < n; ++k) {
    if (CONFLICT) {
      asm volatile(
          "v_pk_mul_f32 v[16:17], v[8:9], v[12:13]\n\tv_pk_mul_f32 v[20:21], v[8:9], v[12:13]\n\t"
          "v_pk_mul_f32 v[24:25], v[8:9], v[12:13]\n\tv_pk_mul_f32 v[28:29], v[8:9], v[12:13]\n\t"
          "v_pk_mul_f32 v[32:33], v[8:9], v[12:13]\n\tv_pk_mul_f32 v[36:37], v[8:9], v[12:13]\n\t"
          "v_pk_mul_f32 v[40:41], v[8:9], v[12:13]\n\tv_pk_mul_f32 v[44:45], v[8:9], v[12:13]\n\t"
          "v_pk_mul_f32 v[16:17], v[8:9], v[12:13]\n\tv_pk_mul_f32 v[20:21], v[8:9], v[12:13]\n\t"
          "v_pk_mul_f32 v[24:25], v[8:9], v[12:13]\n\tv_pk_mul_f32 v[28:29], v[8:9], v[12:13]\n\t"
          "v_pk_mul_f32 v[32:33], v[8:9], v[12:13]\n\tv_pk_mul_f32 v[36:37], v[8:9], v[12:13]\n\t"
          "v_pk_mul_f32 v[40:41], v[8:9], v[12:13]\n\tv_pk_mul_f32 v[44:45], v[8:9], v[12:13]\n\t"
          ::: "v8","v9","v12","v13","v14","v15","v16","v17","v20","v21","v24","v25","v28","v29","v32","v33","v36","v37","v40","v41","v44","v45");
    } else {
      asm volatile(
          "v_pk_mul_f32 v[16:17], v[8:9], v[14:15]\n\tv_pk_mul_f32 v[20:21], v[8:9], v[14:15]\n\t"
          "v_pk_mul_f32 v[24:25], v[8:9], v[14:15]\n\tv_pk_mul_f32 v[28:29], v[8:9], v[14:15]\n\t"
          "v_pk_mul_f32 v[32:33], v[8:9], v[14:15]\n\tv_pk_mul_f32 v[36:37], v[8:9], v[14:15]\n\t"
          "v_pk_mul_f32 v[40:41], v[8:9], v[14:15]\n\tv_pk_mul_f32 v[44:45], v[8:9], v[14:15]\n\t"
          "v_pk_mul_f32 v[16:17], v[8:9], v[14:15]\n\tv_pk_mul_f32 v[20:21], v[8:9], v[14:15]\n\t"
          "v_pk_mul_f32 v[24:25], v[8:9], v[14:15]\n\tv_pk_mul_f32 v[28:29], v[8:9], v[14:15]\n\t"
          "v_pk_mul_f32 v[32:33], v[8:9], v[14:15]\n\tv_pk_mul_f32 v[36:37], v[8:9], v[14:15]\n\t"
          "v_pk_mul_f32 v[40:41], v[8:9], v[14:15]\n\tv_pk_mul_f32 v[44:45], v[8:9], v[14:15]\n\t"
          ::: "v8","v9","v12","v13","v14","v15","v16","v17","v20","v21","v24","v25","v28","v29","v32","v33","v36","v37","v40","v41","v44","v45");
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int CONFLICT> int run_bank(float *d_out, int blocks) {
  const int n = 20000;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kbank<CONFLICT>, dim3(blocks), dim3(256), 0, 0, 16, d_out);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(kbank<CONFLICT>, dim3(blocks), dim3(256), 0, 0, n, d_out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double wave_instr = (double)blocks * 4 * n * 16;
  double per = wave_instr / 1024.0 / (ms * 1e3);
  printf("mode%d v_pk src banks %-9s blocks %6d  %8.3f ms  -> %.2f cycles/pk-instr @2.4GHz\n", 6 + CONFLICT,
         CONFLICT ? "same" : "different", blocks, ms, 2400.0 / per);
  return 0;
}

template <int MODE> int run(const char *name, int valu_per_4, const float4 *d_tab, float *d_out, int n, int iters, int blocks) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_tab, n, 1, d_out, 0.3f, 0.5f, -0.8f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_tab, n, iters, d_out, 0.3f, 0.5f, -0.8f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double waves = (double)blocks * 4;
  double wave_instr = waves * iters * (n / 4.0) * valu_per_4;
  double per_simd_per_us = wave_instr / 1024.0 / (ms * 1e3);
  printf("%-28s blocks %6d  %8.3f ms  %6.1f VALU wave-instr/us/SIMD  -> %.2f cycles/instr @2.4GHz (%.1f Tlane-op/s) %.1f Gtests/s\n",
         name, blocks, ms, per_simd_per_us, 2400.0 / per_simd_per_us, wave_instr * 64 / (ms * 1e-3) / 1e12,
         waves * iters * (double)n * 64.0 / (ms * 1e-3) / 1e9);
  return 0;
}

int main() {
  const int n = 10000;
  std::vector<float4> h(n);
  for (int i = 0; i < n; i++) h[i] = make_float4(0.01f * i, 1.f, 2.f, 1e9f);
  float4 *d_tab; float *d_out;
  CHECK(hipMalloc((void **)&d_tab, n * sizeof(float4)));
  CHECK(hipMalloc((void **)&d_out, 4));
  CHECK(hipMemcpy(d_tab, h.data(), n * sizeof(float4), hipMemcpyHostToDevice));
  for (int blocks : {2048, 8192}) {
    int iters = 16384 * 8 / blocks; if (iters < 1) iters = 1; if (iters > 64) iters = 64;
    if (run<0>("mode0 vgpr mul/add x32", 32, d_tab, d_out, n, iters, blocks)) return 1;
    if (run<1>("mode1 sphere body smem", 32, d_tab, d_out, n, iters, blocks)) return 1;
    if (run<2>("mode2 +filter/branch smem", 31, d_tab, d_out, n, iters, blocks)) return 1;
    if (run<3>("mode3 +filter/branch lds", 31, d_tab, d_out, n, iters, blocks)) return 1;
    if (run_pk(d_out, n, iters, blocks)) return 1;
    if (run_asm(d_tab, d_out, n, iters, blocks)) return 1;
    if (run_bank<0>(d_out, blocks)) return 1;
    if (run_bank<1>(d_out, blocks)) return 1;
  }
  return 0;
}
